#!/usr/bin/env python3
"""GPU probe: the 1x1 conv + fused residual add (fq_conv2d_i8_add_resident) at equal byte counts but different row
contiguity (K = 256 / 128 / 64 channels per pixel row): 4.8-5.3 TB/s in every case, i.e. the kernel's streaming rate does
not depend on how the output rows are split between workgroups (a pure streaming add reaches 6.4 TB/s)."""
import os
import sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
def run(N, C, H, K):
    x = torch.randint(-128, 128, (N, H, H, C), dtype=torch.int8, device="cuda")
    w = nat.pack_weight_krsc(torch.randint(-127, 128, (K, C, 1, 1), device="cuda").float())
    qb = torch.randint(-100, 100, (K,), device="cuda").float()
    res = torch.randint(-2000, 2000, (N, H, H, K), dtype=torch.int16, device="cuda")
    f = lambda: nat.conv2d_i8_add_resident(x, w, qb, (1, 1), (0, 0), (1, 1), 9, 5, res, 6, True, 6, True, 6, True)
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    byts = N * H * H * (K * 5 + C)
    print("N=%d C=%d H=%d K=%d: %.1f us, %.2f TB/s" % (N, C, H, K, us, byts / us / 1e6))
run(128, 64, 56, 256)
run(256, 64, 56, 128)
run(512, 64, 56, 64)
run(128, 128, 28, 512)
run(512, 128, 28, 128)
