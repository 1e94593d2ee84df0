#!/usr/bin/env python3
"""GPU probe: fq_conv_kxk_f32 on the 3x3 layer shapes of ResNet-50 against torch's convolution -- agreement, folded
statistics, time beside the library convolution followed by the bias-add producer.  usage: conv_kxk_probe.py [batch=256]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(64, 56, 1, 3), (128, 56, 2, 1), (128, 28, 1, 3), (256, 28, 2, 1), (256, 14, 1, 5), (512, 14, 2, 1), (512, 7, 1, 2)]   # C, H, stride, count
def timed(fn, n=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
g = torch.Generator(device="cuda").manual_seed(3)
tot_own = tot_lib = 0.0
for c, h, s, count in SHAPES:
    x = torch.randn(B, c, h, h, device="cuda", generator=g)
    w = torch.randn(c, c, 3, 3, device="cuda", generator=g) * (9 * c) ** -0.5
    bias = torch.randn(c, device="cuda", generator=g)
    wt = nat.pack_kxk_weight(w)
    ho = (h + 2 - 3) // s + 1
    y = torch.empty(B, c, ho, ho, device="cuda"); r = torch.empty_like(y)
    mx = torch.zeros(2, device="cuda")
    ref = torch.nn.functional.conv2d(x, w, bias, stride=s, padding=1)
    bound = torch.nn.functional.conv2d(x.abs(), w.abs(), bias.abs(), stride=s, padding=1)
    nat.conv_kxk_f32(x, wt, bias, (3, 3), s, 1, max_dev=mx, row=1, relu_out=r, out=y)
    err = float(((y - ref).abs() / bound).max())
    ok = float(mx[1]) == float(y.abs().max()) and torch.equal(r, torch.relu(y)) and torch.equal(y, nat.conv_kxk_f32(x, wt, bias, (3, 3), s, 1))
    own = timed(lambda: nat.conv_kxk_f32(x, wt, bias, (3, 3), s, 1, max_dev=mx, row=1, relu_out=r, out=y))
    lib = timed(lambda: nat.bias_add_absmax(torch.nn.functional.conv2d(x, w, None, stride=s, padding=1), bias, mx, 0, relu_out=r))
    flop = 2.0 * B * c * ho * ho * c * 9
    tot_own += own * count; tot_lib += lib * count
    print("%4d->%-4d %2dx%-2d s%d x%d  err %.1e  stats/relu/repeat %s | own %.3f ms %6.1f TFLOP/s | library+producer %.3f ms (%5.1f) | x%.2f"
          % (c, c, h, h, s, count, err, ok, own, flop / own / 1e9, lib, flop / lib / 1e9, lib / own), flush=True)
print("all 16 3x3 layers, batch %d: own %.2f ms, library + producer %.2f ms" % (B, tot_own, tot_lib))
