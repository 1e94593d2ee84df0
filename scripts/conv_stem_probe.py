#!/usr/bin/env python3
"""GPU probe: fq_conv_stem_f32 (7x7/2, 3 -> 64) against torch's convolution + the bias-add producer it replaces.
usage: conv_stem_probe.py [batch=256] [size=224]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 224
x = torch.randn(B, 3, HW, HW, device="cuda")
w = torch.randn(64, 3, 7, 7, device="cuda") * 147 ** -0.5
b = torch.randn(64, device="cuda")
wp = nat.pack_stem_weight(w)
ho = (HW + 6 - 7) // 2 + 1
y = torch.empty(B, 64, ho, ho, device="cuda"); r = torch.empty_like(y)
mx = torch.zeros(2, device="cuda"); iv = torch.ones(2, device="cuda"); hist = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
ref = torch.nn.functional.conv2d(x, w, b, stride=2, padding=3)
bound = torch.nn.functional.conv2d(x.abs(), w.abs(), b.abs(), stride=2, padding=3)
nat.conv_stem_f32(x, wp, b, 64, (7, 7), 2, 3, max_dev=mx, row=1, relu_out=r, out=y)
print("err %.1e (relative to sum|w||x|)  max ok %s  relu ok %s" % (float(((y - ref).abs() / bound).max()), float(mx[1]) == float(y.abs().max()), torch.equal(r, torch.relu(y))))
def timed(fn, n=10):
    fn(); fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
flop = 2.0 * B * 64 * ho * ho * 147
for name, fn in (("own, abs-max + ReLU copy", lambda: nat.conv_stem_f32(x, wp, b, 64, (7, 7), 2, 3, max_dev=mx, row=1, relu_out=r, out=y)),
                 ("own, histogram + ReLU copy", lambda: nat.conv_stem_f32(x, wp, b, 64, (7, 7), 2, 3, interval_dev=iv, hist_dev=hist, row=1, relu_out=r, out=y)),
                 ("own, plain", lambda: nat.conv_stem_f32(x, wp, b, 64, (7, 7), 2, 3, out=y)),
                 ("library conv + bias_add_absmax (ReLU copy)", lambda: nat.bias_add_absmax(torch.nn.functional.conv2d(x, w, None, stride=2, padding=3), b, mx, 0, relu_out=r)),
                 ("library conv alone", lambda: torch.nn.functional.conv2d(x, w, None, stride=2, padding=3))):
    ms = timed(fn)
    print("%-46s %.3f ms  %.1f TFLOP/s (147-tap flops)" % (name, ms, flop / ms / 1e9))
