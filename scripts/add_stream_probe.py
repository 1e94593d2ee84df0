#!/usr/bin/env python3
"""GPU probe: what a PURE streaming kernel reaches on the byte mix of a residual block's tail -- fq_add_resident on an int8 and an
int16 operand, writing the int16 sum and its int8 re-quantisation (1 + 2 bytes read, 2 + 1 written per element, no matrix work,
no LDS) -- at the element counts of ResNet-50's stages at 256 images; the ceiling the fused conv3 + add kernels are measured
against.   usage: add_stream_probe.py [batch=256]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(1)
for (C, H) in ((256, 56), (512, 28), (1024, 14), (2048, 7)):
    sets = 3 if C == 256 else 6                       # rotate operand sets: nothing stays in the Infinity Cache for the big ones
    xs = [torch.randint(-128, 128, (B, H, H, C), dtype=torch.int8, device="cuda", generator=g) for _ in range(sets)]
    ys = [torch.randint(-3000, 3000, (B, H, H, C), dtype=torch.int16, device="cuda", generator=g) for _ in range(sets)]
    n = xs[0].numel()
    for want_wide, want_narrow in ((True, True), (True, False), (False, True)):
        for i in range(sets):
            nat.add_resident(xs[i], 4, ys[i], 5, want_wide, 5, want_narrow, 4, True)
        torch.cuda.synchronize()
        reps = 12
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for i, (a, b) in enumerate(ev):
            a.record(); nat.add_resident(xs[i % sets], 4, ys[i % sets], 5, want_wide, 5, want_narrow, 4, True); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)[reps // 2] * 1e3
        nbytes = n * (3 + (2 if want_wide else 0) + (1 if want_narrow else 0))
        print("%4d ch @%2dx%-2d  wide %d narrow %d: %7.1f us  %6.0f GB/s  (%.0f MB)" % (C, H, H, want_wide, want_narrow, t, nbytes / t / 1e3, nbytes / 1e6))
