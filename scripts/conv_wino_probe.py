#!/usr/bin/env python3
"""GPU probe: fq_conv3x3_wino_f32 on the stride-1 3x3 layer shapes of ResNet-50 against an fp64 convolution and against the
direct kernel (fq_conv_kxk_f32) -- error of both relative to sum |w||x|, folded statistics, ReLU copy, time.
usage: conv_wino_probe.py [batch=256] [form=max|hist|plain]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
if os.environ.get("FQ_WINO_LIB"):                              # a debug build (make wino_ablate): timing only, results are wrong
    nat.LIB_PATH = nat.LIB_PATH.replace("libfq_hip.so", "libfq_hip_%s.so" % os.environ["FQ_WINO_LIB"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
FORM = sys.argv[2] if len(sys.argv) > 2 else "max"
SHAPES = [(64, 56, 3), (128, 28, 3), (256, 14, 5), (512, 7, 2)]   # C, H, count
if len(sys.argv) > 3:
    SHAPES = [tuple(int(v) for v in s.split("x")) + (1,) for s in sys.argv[3:]]
def timed(fn, n=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
g = torch.Generator(device="cuda").manual_seed(3)
tot_w = tot_d = 0.0
for c, h, count in SHAPES:
    x = torch.randn(B, c, h, h, device="cuda", generator=g)
    w = torch.randn(c, c, 3, 3, device="cuda", generator=g) * (9 * c) ** -0.5
    bias = torch.randn(c, device="cuda", generator=g)
    wt = nat.pack_kxk_weight(w)
    u = nat.pack_wino_weight(w)
    y = torch.empty(B, c, h, h, device="cuda"); r = torch.empty_like(y)
    yd = torch.empty_like(y)
    mx = torch.zeros(2, device="cuda")
    iv = torch.full((2,), 16.0 / 2048, device="cuda"); hist = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
    nb = min(B, 16)                                             # the fp64 reference on a slice: it is slow
    ref = torch.nn.functional.conv2d(x[:nb].double(), w.double(), bias.double(), padding=1)
    bound = torch.nn.functional.conv2d(x[:nb].abs().double(), w.abs().double(), bias.abs().double(), padding=1)
    kw = dict(max_dev=mx, row=1) if FORM == "max" else (dict(interval_dev=iv, hist_dev=hist, row=1) if FORM == "hist" else {})
    nat.conv_wino_f32(x, u, bias, c, relu_out=r, out=y, **kw)
    nat.conv_kxk_f32(x, wt, bias, (3, 3), 1, 1, out=yd)
    torch.cuda.synchronize()
    err_w = float(((y[:nb].double() - ref).abs() / bound).max())
    err_d = float(((yd[:nb].double() - ref).abs() / bound).max())
    ok = torch.equal(r, torch.relu(y)) and torch.equal(y, nat.conv_wino_f32(x, u, bias, c))
    if FORM == "max":
        ok = ok and float(mx[1]) == float(y.abs().max())
    elif FORM == "hist":
        want = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
        nat.hist2048_seg([y], [1], iv, want)
        ok = ok and torch.equal(hist, want)
    tw = timed(lambda: nat.conv_wino_f32(x, u, bias, c, relu_out=r, out=y, **kw))
    kd = dict(max_dev=mx, row=0) if FORM == "max" else (dict(interval_dev=iv, hist_dev=hist, row=0) if FORM == "hist" else {})
    td = timed(lambda: nat.conv_kxk_f32(x, wt, bias, (3, 3), 1, 1, relu_out=r, out=yd, **kd))
    flop = 2.0 * B * c * h * h * c * 9
    tot_w += tw * count; tot_d += td * count
    print("%4d->%-4d %2dx%-2d x%d  err wino %.1e direct %.1e (of sum|w||x|)  stats/relu/repeat %s | wino %.3f ms %6.1f eff. TFLOP/s | direct %.3f ms %6.1f | x%.2f"
          % (c, c, h, h, count, err_w, err_d, ok, tw, flop / tw / 1e9, td, flop / td / 1e9, td / tw), flush=True)
print("the 13 stride-1 3x3 layers, batch %d, %s form: wino %.2f ms, direct %.2f ms" % (B, FORM, tot_w, tot_d))
