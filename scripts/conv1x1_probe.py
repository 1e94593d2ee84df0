#!/usr/bin/env python3
"""GPU probe: fq_conv1x1_f32 on every 1x1 convolution shape of ResNet-50 against torch's convolution -- agreement (error
relative to sum |w||x|), the folded abs-max / histogram against the streaming kernels on the same output, and time per
layer (HIP events, 10 launches) beside the library convolution followed by the bias-add producer it replaces.
usage: conv1x1_probe.py [batch=256] [mode=max|hist|none]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MODE = sys.argv[2] if len(sys.argv) > 2 else "max"
SHAPES = [  # Cin, Cout, H(in), stride, relu behind it, count in ResNet-50
    (64, 64, 56, 1, True, 1), (64, 256, 56, 1, False, 4), (256, 64, 56, 1, True, 2),
    (256, 128, 56, 1, True, 1), (128, 512, 28, 1, False, 4), (512, 128, 28, 1, True, 3), (256, 512, 56, 2, False, 1),
    (512, 256, 28, 1, True, 1), (256, 1024, 14, 1, False, 6), (1024, 256, 14, 1, True, 5), (512, 1024, 28, 2, False, 1),
    (1024, 512, 14, 1, True, 1), (512, 2048, 7, 1, False, 3), (2048, 512, 7, 1, True, 2), (1024, 2048, 14, 2, False, 1),
]


def timed(fn, n=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot_own = tot_lib = tot_flop = 0.0
g = torch.Generator(device="cuda").manual_seed(1)
for cin, cout, h, s, relu, count in SHAPES:
    x = torch.randn(B, cin, h, h, device="cuda", generator=g)
    w = torch.randn(cout, cin, 1, 1, device="cuda", generator=g) * (cin ** -0.5)
    bias = torch.randn(cout, device="cuda", generator=g)
    wt = w.view(cout, cin).t().contiguous()
    ho = (h - 1) // s + 1
    mx = torch.zeros(2, device="cuda")
    iv = torch.full((2,), 1.0, device="cuda")
    hist = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
    r = torch.empty(B, cout, ho, ho, device="cuda") if relu else None
    y = torch.empty(B, cout, ho, ho, device="cuda")
    ref = torch.nn.functional.conv2d(x, w, bias, stride=s)
    bound = torch.nn.functional.conv2d(x.abs(), w.abs(), bias.abs(), stride=s)
    _native.conv1x1_f32(x, wt, bias, s, max_dev=mx, row=1, relu_out=r, out=y)
    err = float(((y - ref).abs() / bound).max())
    ok_max = float(mx[1]) == float(y.abs().max())
    iv[1] = float(mx[1]) / 2048 + 1e-12
    _native.conv1x1_f32(x, wt, bias, s, interval_dev=iv, hist_dev=hist, row=1, relu_out=r, out=y)
    want = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
    _native.hist2048_seg([y], [1], iv, want)
    ok_hist = torch.equal(hist, want)
    ok_relu = (not relu) or torch.equal(r, torch.relu(y))
    y2 = _native.conv1x1_f32(x, wt, bias, s)
    same = torch.equal(y, y2)
    if MODE == "max":
        own = timed(lambda: _native.conv1x1_f32(x, wt, bias, s, max_dev=mx, row=1, relu_out=r, out=y))
    elif MODE == "hist":
        own = timed(lambda: _native.conv1x1_f32(x, wt, bias, s, interval_dev=iv, hist_dev=hist, row=1, relu_out=r, out=y))
    else:
        own = timed(lambda: _native.conv1x1_f32(x, wt, bias, s, out=y))

    def lib():
        raw = torch.nn.functional.conv2d(x, w, None, stride=s)
        if MODE == "max":
            _native.bias_add_absmax(raw, bias, mx, 0, relu_out=r)
        elif MODE == "hist":
            _native.bias_add_hist(raw, bias, iv, hist, 0, relu_out=r)
        else:
            raw.add_(bias.view(1, -1, 1, 1))
    libt = timed(lib)
    flop = 2.0 * B * cout * ho * ho * cin
    tot_own += own * count; tot_lib += libt * count; tot_flop += flop * count
    print("%4d->%-4d %2dx%-2d s%d x%d  err %.1e  max %s hist %s relu %s repeat %s | own %.3f ms %6.1f TFLOP/s | library+producer %.3f ms | x%.2f"
          % (cin, cout, h, h, s, count, err, ok_max, ok_hist, ok_relu, same, own, flop / own / 1e9, libt, libt / own), flush=True)
print("all 36 layers, batch %d, mode %s: own %.2f ms (%.1f TFLOP/s), library + producer %.2f ms" % (B, MODE, tot_own, tot_flop / tot_own / 1e9, tot_lib))
