"""GPU box: int8 convolution time against the batch size (the tile count's remainder over the resident workgroup slots)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pytorch-quantity_amd", "quantity")]
from common.quantity import _native as nat

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for (C, H, K, R, st, pd) in ((256, 14, 256, 3, 1, 1), (512, 7, 512, 3, 1, 1), (128, 28, 128, 3, 1, 1), (1024, 14, 256, 1, 1, 0), (2048, 7, 512, 1, 1, 0), (512, 28, 128, 1, 1, 0)):
    w = torch.randint(-127, 128, (K, C, R, R), device="cuda").float()
    qb = torch.randint(-100, 100, (K,), device="cuda").float()
    wq = nat.pack_weight_krsc(w)
    line = []
    for B in (64, 96, 128, 160, 168, 176, 192, 224, 256, 288, 320, 336, 352, 384, 512):
        x = torch.randn(B, C, H, H, device="cuda") * 2
        xq = nat.quantize_i8_nhwc(x, 4, wq.shape[-1])
        t = timed(lambda: nat.conv2d_i8_resident(xq, wq, qb, (st, st), (pd, pd), (1, 1), 8, 4, False, True, True))
        tiles = -(-(B * H * H) // 128) * (K // (128 if K >= 128 else 64))
        line.append("%d:%.1f(%.2f)" % (B, t, tiles / 256.0))
    print("%dx%d %d->%d k%d   " % (H, H, C, K, R) + "  ".join(line))
