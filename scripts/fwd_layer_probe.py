#!/usr/bin/env python3
"""GPU probe: per-module time of the float ResNet-50 forward the calibration runs (HIP events in forward pre/post hooks),
to see which MIOpen / torch kernels the 16.9 ms per batch of 128 go to."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, torch.device("cuda"))
sys.stdout = out
torch.backends.cudnn.benchmark = False
x = torch.randn(B, 3, 224, 224, device="cuda")
recs = {}
shapes = {}
def pre(name):
    def f(m, i):
        e = torch.cuda.Event(enable_timing=True); e.record(); recs.setdefault(name, []).append([e, None])
    return f
def post(name):
    def f(m, i, o):
        e = torch.cuda.Event(enable_timing=True); e.record(); recs[name][-1][1] = e
        shapes[name] = tuple(o.shape)
    return f
for name, m in model.named_modules():
    if len(list(m.children())) == 0:
        m.register_forward_pre_hook(pre(name)); m.register_forward_hook(post(name))
with torch.no_grad():
    for _ in range(3): model(x)
    recs.clear()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): model(x)
    b.record(); torch.cuda.synchronize()
print("forward B=%d: %.2f ms" % (B, a.elapsed_time(b) / 5))
rows = []
for name, lst in recs.items():
    ms = sum(s.elapsed_time(e) for s, e in lst) / 5
    m = dict(model.named_modules())[name]
    rows.append((ms, name, type(m).__name__, str(getattr(m, "kernel_size", "")), str(getattr(m, "stride", "")), getattr(m, "in_channels", ""), getattr(m, "out_channels", "")))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
by = {}
for r in rows: by[r[2]] = by.get(r[2], 0) + r[0]
print("by module type (ms):", {k: round(v, 2) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}, "sum %.2f" % tot)
for r in rows[:14]: print("%7.3f ms  %-28s %-10s k%s s%s %s->%s" % r)
print("convolutions, slowest first (fp32 TFLOP/s = 2 * MAC / time):")
conv_ms = conv_flop = 0.0
for ms, name, kind, k, st, ci, co in rows:
    if kind != "Conv2d":
        continue
    m = dict(model.named_modules())[name]
    n, c, h, w = shapes[name]
    flop = 2.0 * n * c * h * w * (ci // m.groups) * m.kernel_size[0] * m.kernel_size[1]
    conv_ms += ms; conv_flop += flop
    print("%7.3f ms %6.1f TFLOP/s  %-24s k%s s%s %4s->%-4s out %dx%d" % (ms, flop / ms / 1e9, name, k, st, ci, co, h, w))
print("all convolutions: %.2f ms, %.1f TFLOP/s" % (conv_ms, conv_flop / conv_ms / 1e9))
