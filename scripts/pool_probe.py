#!/usr/bin/env python3
"""GPU probe: fq_maxpool2d_f32 / fq_avgpool_global_f32 on ResNet-50's two pooling layers beside torch's kernels.
usage: pool_probe.py [batch=256]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
def timed(fn, n=20):
    fn(); fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
x = torch.randn(B, 64, 112, 112, device="cuda").relu_()
for name, fn, by in (("own max pool 3x3/2", lambda: nat.maxpool2d_f32(x, (3, 3), (2, 2), (1, 1)), x.numel() * 4 * 1.25),
                     ("torch max pool", lambda: torch.nn.functional.max_pool2d(x, 3, 2, 1), x.numel() * 4 * 1.25)):
    ms = timed(fn); print("%-22s %.3f ms  %.2f TB/s (read once + write)" % (name, ms, by / ms / 1e9))
z = torch.randn(B, 2048, 7, 7, device="cuda")
for name, fn in (("own global avg 7x7", lambda: nat.avgpool_global_f32(z)), ("torch avg pool", lambda: torch.nn.functional.avg_pool2d(z, 7))):
    ms = timed(fn); print("%-22s %.3f ms  %.2f TB/s" % (name, ms, z.numel() * 4 / ms / 1e9))
