#!/usr/bin/env python3
"""GPU probe: fused QuanDequan rate at the cared-tensor sizes of ResNet-50 (batch 128 and 64)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]
out = []
for n in (6422528, 12845056, 25690112, 51380224, 102760448, 205520896):
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    ms = timeit(lambda: nat.quandequan(x, 4, out=y))
    ms2 = timeit(lambda: nat.quandequan(x, 4, out=x))
    out.append("%5.0f/%5.0f" % (n * 8 / ms / 1e6, n * 8 / ms2 / 1e6))
print("var=%s cap=%s GB/s (out-of-place/in-place) for 6.4M..205M elems: %s" % (os.environ.get("FQ_OPS_VAR"), os.environ.get("FQ_OPS_CAP"), "  ".join(out)))
