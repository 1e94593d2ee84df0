#!/usr/bin/env python3
"""GPU probe: fq_hist2048_chain_seg on ResNet-50's four residual stages at 256 images (L conv3 outputs + one shortcut each), HIP events,
two rotating operand sets per stage; run it under FQ_CHAIN_WG_PER_CU / FQ_CHAIN_MIN_CHUNKS to tune the launch shape.
usage: chain_hist_probe.py [batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(3)
stages = [("stage 1", 3, B * 256 * 56 * 56), ("stage 2", 4, B * 512 * 28 * 28), ("stage 3", 6, B * 1024 * 14 * 14), ("stage 4", 3, B * 2048 * 7 * 7)]
iv = torch.full((16,), 0.004, device="cuda")
hist = torch.zeros(16, 2048, dtype=torch.int64, device="cuda")
total_ms, total_b = 0.0, 0.0
print("FQ_CHAIN_WG_PER_CU=%s FQ_CHAIN_MIN_CHUNKS=%s" % (os.environ.get("FQ_CHAIN_WG_PER_CU"), os.environ.get("FQ_CHAIN_MIN_CHUNKS")))
for name, L, n in stages:
    sets = [(torch.randn(n, generator=g, device="cuda"), [torch.randn(n, generator=g, device="cuda") for _ in range(L)]) for _ in range(2)]
    rows_y, rows_s = [2 * k for k in range(L)], [2 * k + 1 for k in range(L)]
    for head, ys in sets:
        nat.hist2048_chain_seg([(head, ys, rows_y, rows_s)], iv, hist)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for i, (a, b) in enumerate(evs):
        head, ys = sets[i % 2]
        a.record()
        nat.hist2048_chain_seg([(head, ys, rows_y, rows_s)], iv, hist)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    by = 4.0 * n * (L + 1)
    total_ms += ms
    total_b += by
    print("%s  L=%d  %.3f GB  %.1f us  %.2f TB/s" % (name, L, by / 1e9, ms * 1e3, by / ms / 1e9))
    del sets
print("all four: %.3f ms, %.2f TB/s" % (total_ms, total_b / total_ms / 1e9))
