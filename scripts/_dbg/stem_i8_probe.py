"""GPU box: fq_conv2d_i8_stem alone at the bench's shape (256 x 3 x 224 x 224 -> 64, 7x7/2), device time per launch + a checksum."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
w = torch.randint(-127, 128, (64, 3, 7, 7), device="cuda", generator=g).float()
qb = torch.randint(-60, 61, (64,), device="cuda", generator=g).float()
ws = nat.pack_weight_stem(w)
run = lambda: nat.conv2d_i8_stem(x, ws, qb, 64, 7, (2, 2), (3, 3), 5, 9, 4, True)
q = run(); q = run()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    torch.cuda.synchronize(); a.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e) / 20 * 1e3)
print("%s  stem %d images: %.1f us per launch (best of 5 x 20; all: %s)  checksum %d" % (
    os.path.basename(nat.LIB_PATH), B, min(ts), " ".join("%.1f" % t for t in ts), int(q.to(torch.int64).sum())))
