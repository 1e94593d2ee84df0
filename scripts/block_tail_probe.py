#!/usr/bin/env python3
"""GPU probe: the four conv3 + NewAdd launch shapes of ResNet-50's bottlenecks (fq_conv2d_i8_add_resident: 1x1 expand, int16
residual in, int16 sum + int8 re-quantisation out) at `batch` images, each timed over rotating operand sets larger than the
Infinity Cache where the layer's own tensors are not.  One line per shape: us, algorithmic TB/s, the kernel that ran.
usage: block_tail_probe.py [batch]      A/B: FQ_RES_EARLY=0|1, FQ_BLOCK_TAIL=0|1 in the environment."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
print("FQ_RES_EARLY=%s FQ_BLOCK_TAIL=%s batch %d" % (os.environ.get("FQ_RES_EARLY"), os.environ.get("FQ_BLOCK_TAIL"), B))


def run(C, H, K, res_dtype=torch.int16, wide=True):
    sets = max(2, int(600e6 // (B * H * H * K * 5)) + 1)            # operands + results of all sets exceed the 256 MB cache
    xs = [torch.randint(-128, 128, (B, H, H, C), dtype=torch.int8, device="cuda") for _ in range(sets)]
    w = nat.pack_weight_krsc(torch.randint(-127, 128, (K, C, 1, 1), device="cuda").float())
    qb = torch.randint(-100, 100, (K,), device="cuda").float()
    # (grids as in the calibrated ResNet-50: conv output 2^-4, residual and sum 2^-5, next Quantity 2^-4 -> shx 1, shy 0, k 1: the
    #  packed-int16 form of the add; rs 9)
    lim = 2000 if res_dtype == torch.int16 else 128
    rs = [torch.randint(-lim, lim, (B, H, H, K), dtype=res_dtype, device="cuda") for _ in range(sets)]
    nat.conv_variant_log = log = {}
    f = lambda i: nat.conv2d_i8_add_resident(xs[i % sets], w, qb, (1, 1), (0, 0), (1, 1), 9, 4, rs[i % sets], 5, wide, 5, True, 4, True)
    for i in range(3):
        f(i)
    nat.conv_variant_log = None
    reps = 5 * sets
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        f(i)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    byts = B * H * H * (K * ((2 if res_dtype == torch.int16 else 1) + (2 if wide else 0) + 1) + C)
    print("%4d -> %4d @%2dx%-2d res %-5s wide %d: %7.1f us  %5.2f TB/s  (%6.1f us at 8 TB/s)  %s" %
          (C, K, H, H, str(res_dtype).replace("torch.", ""), wide, us, byts / us / 1e6, byts / 8e6, ",".join(log)))


for shape in ((64, 56, 256), (128, 28, 512), (256, 14, 1024), (512, 7, 2048)):
    run(*shape)
run(64, 56, 256, torch.int8, True)        # a stage's first block: the residual is the projection's int8 output
run(64, 56, 256, torch.int16, False)      # a stage's last block: nobody reads the wide sum


def run_bt(C, H, K3, C2, res_dtype=torch.int16, wide=True):
    """The same block tail + the next conv1: as two launches and as fq_block_tail_i8."""
    sets = max(2, int(600e6 // (B * H * H * K3 * 5)) + 1)
    xs = [torch.randint(-128, 128, (B, H, H, C), dtype=torch.int8, device="cuda") for _ in range(sets)]
    w3 = nat.pack_weight_krsc(torch.randint(-127, 128, (K3, C, 1, 1), device="cuda").float())
    b3 = torch.randint(-100, 100, (K3,), device="cuda").float()
    lim = 2000 if res_dtype == torch.int16 else 128
    rs = [torch.randint(-lim, lim, (B, H, H, K3), dtype=res_dtype, device="cuda") for _ in range(sets)]
    w1 = nat.pack_weight_krsc(torch.randint(-127, 128, (C2, K3, 1, 1), device="cuda").float()) if C2 else None
    b1 = torch.randint(-100, 100, (C2,), device="cuda").float() if C2 else None

    def two(i):
        _w, n = nat.conv2d_i8_add_resident(xs[i % sets], w3, b3, (1, 1), (0, 0), (1, 1), 9, 4, rs[i % sets], 5, wide, 5, True, 4, True)
        if C2:
            nat.conv2d_i8_resident(n, w1, b1, (1, 1), (0, 0), (1, 1), 10, 4, False, True, True)

    def one(i):
        nat.block_tail_i8(xs[i % sets], w3, b3, 9, 4, rs[i % sets], 5, wide, 5, not C2, 4, True, w1, b1, 10, True)
    out = []
    for f in (two, one):
        for i in range(3):
            f(i)
        reps = 5 * sets
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            f(i)
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps * 1e3)
    rb = 2 if res_dtype == torch.int16 else 1
    b_two = B * H * H * (K3 * (rb + (2 if wide else 0) + 1) + C + (K3 + C2 if C2 else 0))
    b_one = B * H * H * (K3 * (rb + (2 if wide else 0) + (0 if C2 else 1)) + C + C2)
    print("%4d -> %4d -> %3d @%2dx%-2d res %-5s: two launches %7.1f us (%5.2f TB/s) | block_tail %7.1f us (%5.2f TB/s; %6.1f us at 8 TB/s)" %
          (C, K3, C2, H, H, str(res_dtype).replace("torch.", ""), out[0], b_two / out[0] / 1e6, out[1], b_one / out[1] / 1e6, b_one / 8e6))


print("-- conv3 + add + next conv1: two launches vs fq_block_tail_i8")
run_bt(64, 56, 256, 64)
run_bt(64, 56, 256, 64, torch.int8)
run_bt(128, 28, 512, 128)
run_bt(64, 56, 256, 0)
run_bt(128, 28, 512, 0)
run_bt(256, 14, 1024, 0)
run_bt(64, 56, 256, 0, torch.int16, False)
