#!/usr/bin/env python3
"""GPU probe: the split-bf16 1x1 convolution (fq_conv1x1_sb_f32) beside the fp32-MFMA one (fq_conv1x1_f32) on every 1x1 layer
shape of ResNet-50 -- error of both against an fp64 GEMM (relative to sum |w||x|), statistics / ReLU copy / repeatability,
time per layer.  usage: conv1x1_sb_probe.py [batch=256] [mode=max|hist|none]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
if os.environ.get("FQ_SB_LIB"):                                # a debug build (timing only, wrong results)
    nat.LIB_PATH = nat.LIB_PATH.replace("libfq_hip.so", "libfq_hip_%s.so" % os.environ["FQ_SB_LIB"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MODE = sys.argv[2] if len(sys.argv) > 2 else "max"
# cin, cout, hw, stride, count
LAYERS = [(64, 64, 56, 1, 1), (64, 256, 56, 1, 4), (256, 64, 56, 1, 2), (256, 128, 56, 1, 1), (128, 512, 28, 1, 4), (512, 128, 28, 1, 3),
          (256, 512, 56, 2, 1), (512, 256, 28, 1, 1), (256, 1024, 14, 1, 6), (1024, 256, 14, 1, 5), (512, 1024, 28, 2, 1),
          (1024, 512, 14, 1, 1), (512, 2048, 7, 1, 3), (2048, 512, 7, 1, 2), (1024, 2048, 14, 2, 1)]
def timed(fn, n=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
g = torch.Generator(device="cuda").manual_seed(5)
tot_s = tot_f = 0.0
for cin, cout, hw, st, count in LAYERS:
    x = torch.randn(B, cin, hw, hw, device="cuda", generator=g)
    w = torch.randn(cout, cin, device="cuda", generator=g) * cin ** -0.5
    bias = torch.randn(cout, device="cuda", generator=g)
    wt, wsb = w.t().contiguous(), nat.pack_sb_weight(w)
    ho = (hw - 1) // st + 1
    ys = torch.empty(B, cout, ho, ho, device="cuda"); yf = torch.empty_like(ys); r = torch.empty_like(ys)
    mx = torch.zeros(2, device="cuda")
    iv = torch.full((2,), 16.0 / 2048, device="cuda"); hist = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
    kw = dict(max_dev=mx, row=1) if MODE == "max" else (dict(interval_dev=iv, hist_dev=hist, row=1) if MODE == "hist" else {})
    nat.conv1x1_f32(x, wsb, bias, st, relu_out=r, out=ys, **kw)
    nat.conv1x1_f32(x, wt, bias, st, out=yf)
    nb = min(B, 8)
    xs = x[:nb, :, ::st, ::st].reshape(nb, cin, -1).double()
    ref = (torch.matmul(w.double(), xs) + bias.double().view(1, -1, 1)).view(nb, cout, ho, ho)
    bound = (torch.matmul(w.double().abs(), xs.abs()) + bias.double().abs().view(1, -1, 1)).view(nb, cout, ho, ho)
    es = float(((ys[:nb].double() - ref).abs() / bound).max()); ef = float(((yf[:nb].double() - ref).abs() / bound).max())
    ok = torch.equal(r, torch.relu(ys)) and torch.equal(ys, nat.conv1x1_f32(x, wsb, bias, st))
    if MODE == "max":
        ok = ok and float(mx[1]) == float(ys.abs().max())
    elif MODE == "hist":
        want = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
        nat.hist2048_seg([ys], [1], iv, want)
        ok = ok and torch.equal(hist, want)
    ts = timed(lambda: nat.conv1x1_f32(x, wsb, bias, st, relu_out=r, out=ys, **kw))
    kf = dict(max_dev=mx, row=0) if MODE == "max" else (dict(interval_dev=iv, hist_dev=hist, row=0) if MODE == "hist" else {})
    tf = timed(lambda: nat.conv1x1_f32(x, wt, bias, st, relu_out=r, out=yf, **kf))
    flop = 2.0 * B * cout * ho * ho * cin
    tot_s += ts * count; tot_f += tf * count
    print("%4d->%-4d %2dx%-2d s%d x%d  err split-bf16 %.1e fp32-mfma %.1e  stats/relu/repeat %s | split-bf16 %.3f ms %6.1f TFLOP/s | fp32 mfma %.3f ms %6.1f | x%.2f"
          % (cin, cout, hw, hw, st, count, es, ef, ok, ts, flop / ts / 1e9, tf, flop / tf / 1e9, tf / ts), flush=True)
print("all 36 1x1 layers, batch %d, mode %s: split-bf16 %.2f ms, fp32 mfma %.2f ms" % (B, MODE, tot_s, tot_f))
