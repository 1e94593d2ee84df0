"""GPU box: does the tile count's remainder over the 256 CUs show in the fp32 convolution's time?  Sweeps the batch size around 256
for the layers whose 256-image launch has 784 / 1568 tiles and prints time per image."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pytorch-quantity_amd", "quantity")]
from common.quantity import _native as nat

def timed(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for (cin, cout, hw, k) in ((256, 256, 14, 3), (512, 512, 7, 3), (2048, 512, 7, 1), (1024, 256, 14, 1)):
    w = torch.randn(cout, cin, k, k, device="cuda") * (cin * k * k) ** -0.5
    wt = nat.pack_kxk_weight(w) if k > 1 else w.view(cout, cin).t().contiguous()
    b = torch.randn(cout, device="cuda")
    m = torch.zeros(1, device="cuda")
    line = []
    for n in (192, 224, 240, 248, 250, 252, 256, 260, 272, 288, 320):
        x = torch.randn(n, cin, hw, hw, device="cuda")
        if k > 1:
            t = timed(lambda: nat.conv_kxk_f32(x, wt, b, (k, k), 1, 1, max_dev=m, row=0))
        else:
            t = timed(lambda: nat.conv1x1_f32(x, wt, b, 1, max_dev=m, row=0))
        cols = n * hw * hw
        t22 = -(-cols // 128) * -(-cout // 128)
        tiles = -(-cols // 128) * -(-cout // (64 if t22 <= 1024 else 128))
        line.append("%d:%.0fus/%.2f(%.2f)" % (n, t, t / n, tiles / 256.0))
    print("%d->%d @%d k%d  " % (cin, cout, hw, k) + "  ".join(line))
