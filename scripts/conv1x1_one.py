#!/usr/bin/env python3
"""GPU: fq_conv1x1_f32 on one layer shape, for rocprofv3.  usage: conv1x1_one.py Cin Cout H stride batch [max|hist|none] [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native
cin, cout, h, s, B = (int(v) for v in sys.argv[1:6])
mode = sys.argv[6] if len(sys.argv) > 6 else "max"
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
x = torch.randn(B, cin, h, h, device="cuda")
wt = (torch.randn(cin, cout, device="cuda") * cin ** -0.5).contiguous()
bias = torch.randn(cout, device="cuda")
ho = (h - 1) // s + 1
y = torch.empty(B, cout, ho, ho, device="cuda")
mx = torch.zeros(1, device="cuda")
iv = torch.full((1,), 8.0 / 2048, device="cuda")
hist = torch.zeros(1, 2048, dtype=torch.int64, device="cuda")
for _ in range(reps):
    if mode == "max":
        _native.conv1x1_f32(x, wt, bias, s, max_dev=mx, row=0, out=y)
    elif mode == "hist":
        _native.conv1x1_f32(x, wt, bias, s, interval_dev=iv, hist_dev=hist, row=0, out=y)
    else:
        _native.conv1x1_f32(x, wt, bias, s, out=y)
torch.cuda.synchronize()
print("done", float(mx[0]))
