#!/usr/bin/env python3
"""GPU probe: one conv2d_i8 layer shape in a loop (for rocprofv3 --pmc passes).
usage: conv_one.py C H K R stride pad [i8|f32] [iters]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
C, H, K, R, st, pd = [int(v) for v in sys.argv[1:7]]
mode = sys.argv[7] if len(sys.argv) > 7 else "i8"
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
B = int(os.environ.get("FQ_CONV_ONE_BATCH", "128"))
x = torch.randn(B, C, H, H, device="cuda") * 2
w = torch.randint(-127, 128, (K, C, R, R), device="cuda").float()
qb = torch.randint(-100, 100, (K,), device="cuda").float()
wq = nat.pack_weight_krsc(w)
xq = nat.quantize_i8_nhwc(x, 4, wq.shape[-1])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    if mode == "i8":
        nat.conv2d_i8_resident(xq, wq, qb, (st, st), (pd, pd), (1, 1), 8, 4, False, True, True)
    else:
        nat.conv2d_i8(xq, wq, qb, (st, st), (pd, pd), (1, 1), 8, 4)
b.record(); torch.cuda.synchronize()
print("%s: %.1f us" % (" ".join(sys.argv[1:8]), a.elapsed_time(b) / iters * 1e3))
