#!/usr/bin/env python3
"""GPU probe: fq_hist2048_chan (and fq_absmax_chan) per tensor shape of ResNet-50 at 256 images: microseconds and algorithmic
GB/s (4 bytes per element), with the owner flush (rows one workgroup owns are flushed with plain 16-byte read-add-write) and
with atomics everywhere (FQ_CHAN_OWN_FLUSH=0).   usage: chan_hist_probe.py [batch=256] [reps=10]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
SHAPES = [(64, 112, 112), (256, 56, 56), (64, 56, 56), (512, 28, 28), (128, 28, 28), (1024, 14, 14), (256, 14, 14), (2048, 7, 7), (512, 7, 7)]
g = torch.Generator(device="cuda").manual_seed(3)


def timed(fn):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(REPS)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3


print("%-16s %10s | %9s %8s | %9s %8s | %9s %8s" % ("C x H x W", "MB", "own us", "GB/s", "atomic us", "GB/s", "absmax us", "GB/s"))
tot = [0.0, 0.0, 0.0, 0.0]
for (C, H, W) in SHAPES:
    # as many tensors of the shape in one call as make ~1.6 GB: the steady-state rate of the shape, not one tensor's launch
    copies = max(1, int(1.6e9 // (B * C * H * W * 4)))
    xs = [torch.randn(B, C, H, W, device="cuda", generator=g) for _ in range(copies)]
    if C % 3:
        xs = [torch.relu(x) for x in xs]                  # some post-ReLU tensors (half zeros)
    row0s = [C * i for i in range(copies)]
    mx = torch.zeros(C * copies, device="cuda")
    nat.absmax_chan(xs, row0s, mx)
    iv = mx / 2048 + 1e-12
    hist = torch.zeros(C * copies, 2048, dtype=torch.int64, device="cuda")
    nbytes = xs[0].numel() * 4 * copies
    res = []
    for env in ("1", "0"):
        os.environ["FQ_CHAN_OWN_FLUSH"] = env
        res.append(timed(lambda: nat.hist2048_chan(xs, row0s, iv, hist)))
    os.environ.pop("FQ_CHAN_OWN_FLUSH")
    tm = timed(lambda: nat.absmax_chan(xs, row0s, mx))
    print("%-16s %10.1f | %9.1f %8.0f | %9.1f %8.0f | %9.1f %8.0f" % ("%dx%dx%d x%d" % (C, H, W, copies), nbytes / 1e6, res[0], nbytes / res[0] / 1e3,
                                                                  res[1], nbytes / res[1] / 1e3, tm, nbytes / tm / 1e3))
    tot[0] += nbytes; tot[1] += res[0]; tot[2] += res[1]; tot[3] += tm
    del xs
print("%-16s %10.1f | %9.1f %8.0f | %9.1f %8.0f | %9.1f %8.0f" % ("all", tot[0] / 1e6, tot[1], tot[0] / tot[1] / 1e3, tot[2], tot[0] / tot[2] / 1e3,
                                                              tot[3], tot[0] / tot[3] / 1e3))
