#!/usr/bin/env python3
"""GPU probe: the 1x1 layers of ResNet-50 that fq_conv1x1_i8.hip takes, at 256 images, with and without the fused NewAdd
(int16 residual in, int16 sum + int8 out): us, algorithmic GB/s, TOP/s.  FQ_CONV_STREAM=0 in the environment gives the
general kernel on the same shapes (one process per setting: the switches are read once).
usage: stream_bench.py [batch] [iters]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# (C, H, K, stride, add)
LAYERS = [(64, 56, 256, 1, True), (128, 28, 512, 1, True), (256, 14, 1024, 1, True), (512, 7, 2048, 1, True),
          (64, 56, 64, 1, False), (64, 56, 256, 1, False), (256, 56, 64, 1, False), (256, 56, 128, 1, False), (512, 28, 128, 1, False),
          (512, 28, 256, 1, False), (256, 56, 512, 2, False), (512, 28, 1024, 2, False)]


def timeit(fn):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(IT):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / IT * 1e3


print("stream=%s wg_per_cu=%s" % (os.environ.get("FQ_CONV_STREAM", "1"), os.environ.get("FQ_STREAM_WG_PER_CU", "-")))
tot = 0.0
for (C, H, K, st, add) in LAYERS:
    x = torch.randint(-128, 128, (B, H, H, C), device="cuda", dtype=torch.int8)
    w = torch.randint(-127, 128, (K, C, 1, 1), device="cuda").float()
    qb = torch.randint(-100, 100, (K,), device="cuda").float()
    wq = nat.pack_weight_krsc(w)
    P = (H - 1) // st + 1
    if add:
        res = torch.randint(-2000, 2000, (B, P, P, K), device="cuda", dtype=torch.int16)
        us = timeit(lambda: nat.conv2d_i8_add_resident(x, wq, qb, (st, st), (0, 0), (1, 1), 8, 4, res, 4, True, 4, True, 4, True))
        nb = x.numel() + wq.numel() + res.numel() * 2 + B * P * P * K * 3
    else:
        us = timeit(lambda: nat.conv2d_i8_resident(x, wq, qb, (st, st), (0, 0), (1, 1), 8, 4, False, True, True))
        nb = x.numel() + wq.numel() + B * P * P * K
    tot += us
    print("%4d,%2d -> %4d s%d %-4s %8.1f us %7.0f GB/s %7.1f TOP/s" % (C, H, K, st, "add" if add else "", us, nb / us / 1e3,
                                                                    2.0 * B * P * P * K * C / us / 1e6))
print("sum %.1f us" % tot)
