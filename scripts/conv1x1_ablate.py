#!/usr/bin/env python3
"""GPU probe (debug build libfq_hip_ablate.so, -DFQ_C1_ABLATE): time of fq_conv1x1_f32 on one layer with parts of the kernel
left out (FQ_C1_ABLATE bits: 1 no stores, 2 global loads of the first K step only, 4 no barriers in the K loop (results are wrong, timing only)) -- what the pieces cost.
usage: conv1x1_ablate.py Cin Cout H stride batch     (run once per FQ_C1_ABLATE value; the library reads it at first launch)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
nat.LIB_PATH = nat.LIB_PATH.replace("libfq_hip.so", "libfq_hip_ablate.so")
cin, cout, h, s, B = (int(v) for v in sys.argv[1:6])
x = torch.randn(B, cin, h, h, device="cuda")
wt = (torch.randn(cin, cout, device="cuda") * cin ** -0.5).contiguous()
bias = torch.randn(cout, device="cuda")
ho = (h - 1) // s + 1
y = torch.empty(B, cout, ho, ho, device="cuda")
mx = torch.zeros(1, device="cuda")
run = lambda: nat.conv1x1_f32(x, wt, bias, s, max_dev=mx, row=0, out=y)
run(); run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(20):
    run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print("ablate %s  %d->%d %dx%d s%d b%d: %.3f ms  (%.1f TFLOP/s if it were the whole kernel)" % (
    os.environ.get("FQ_C1_ABLATE", "0"), cin, cout, h, h, s, B, ms, 2.0 * B * cout * ho * ho * cin / ms / 1e9))
