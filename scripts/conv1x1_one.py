#!/usr/bin/env python3
"""GPU: fq_conv1x1_f32 (FQ_ONE_KIND=kxk: fq_conv_kxk_f32 3x3 padding 1; =stem: fq_conv_stem_f32 7x7 stride 2 on 3 channels; =wino:
fq_conv3x3_wino_f32; =sb: fq_conv1x1_sb_f32) on
one layer shape, for rocprofv3.  usage: conv1x1_one.py Cin Cout H stride batch [max|hist|none|add|addhist|addkeep] [reps]
(add / addhist: fq_conv1x1_add_f32 / fq_conv1x1_add_hist_f32, the residual tail in one kernel, nothing kept; addkeep: both tensors kept)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native
cin, cout, h, s, B = (int(v) for v in sys.argv[1:6])
mode = sys.argv[6] if len(sys.argv) > 6 else "max"
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
kind = os.environ.get("FQ_ONE_KIND", "c1")
x = torch.randn(B, cin, h, h, device="cuda")
wt = (torch.randn(cin, cout, device="cuda") * cin ** -0.5).contiguous()
if kind == "kxk":
    wt = _native.pack_kxk_weight(torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5)
elif kind == "stem":
    wt = _native.pack_stem_weight(torch.randn(cout, 3, 7, 7, device="cuda") * 147 ** -0.5)
elif kind == "wino":                                      # fq_conv3x3_wino_f32 (3x3, stride 1, padding 1)
    wt = _native.pack_wino_weight(torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5)
elif kind == "sb":                                        # fq_conv1x1_sb_f32: the split-bf16 form of the 1x1 kernel
    wt = _native.pack_sb_weight((torch.randn(cout, cin, 1, 1, device="cuda") * cin ** -0.5))
bias = torch.randn(cout, device="cuda")
ho = (h - 1) // s + 1 if kind in ("c1", "sb") else (h if kind == "wino" else ((h + 2 - 3) // s + 1 if kind == "kxk" else (h + 6 - 7) // 2 + 1))


def conv(**kw):
    if kind == "kxk":
        return _native.conv_kxk_f32(x, wt, bias, (3, 3), s, 1, **kw)
    if kind == "stem":
        return _native.conv_stem_f32(x, wt, bias, cout, (7, 7), 2, 3, **kw)
    if kind == "wino":
        return _native.conv_wino_f32(x, wt, bias, cout, **kw)
    return _native.conv1x1_f32(x, wt, bias, s, **kw)


y = torch.empty(B, cout, ho, ho, device="cuda")
mx = torch.zeros(1, device="cuda")
iv = torch.full((1,), 8.0 / 2048, device="cuda")
hist = torch.zeros(1, 2048, dtype=torch.int64, device="cuda")
res = torch.randn(B, cout, ho, ho, device="cuda") if mode.startswith("add") else None
relu = torch.empty(B, cout, ho, ho, device="cuda") if mode.startswith("add") else None
mx2 = torch.zeros(2, device="cuda")
iv2 = torch.full((2,), 8.0 / 2048, device="cuda")
hist2 = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
ysum = torch.empty(B, cout, ho, ho, device="cuda") if mode == "addkeep" else None
for _ in range(reps):
    if mode == "add":
        _native.conv1x1_add_f32(x, wt, bias, s, res, mx2, 0, 1, relu)
    elif mode == "addkeep":
        _native.conv1x1_add_f32(x, wt, bias, s, res, mx2, 0, 1, relu, out=y, sum_out=ysum)
    elif mode == "addhist":
        _native.conv1x1_add_hist_f32(x, wt, bias, s, res, iv2, hist2, 0, 1, relu)
    elif mode == "max":
        conv(max_dev=mx, row=0, out=y)
    elif mode == "hist":
        conv(interval_dev=iv, hist_dev=hist, row=0, out=y)
    else:
        conv(out=y)
torch.cuda.synchronize()
print("done", float(mx[0]))
