#!/usr/bin/env python3
"""GPU probe: fq_conv2d_i8 / fq_quantize_i8_nhwc on every distinct ResNet-50 layer shape (batch from argv)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
I8OUT = len(sys.argv) > 2 and sys.argv[2] == "i8"      # resident mode: int8 NHWC output with fused ReLU
# (C, H, K, R, stride, pad, count in the net)
LAYERS = [(3, 224, 64, 7, 2, 3, 1),
          (64, 56, 64, 1, 1, 0, 1), (64, 56, 64, 3, 1, 1, 3), (64, 56, 256, 1, 1, 0, 4), (256, 56, 64, 1, 1, 0, 2),
          (256, 56, 128, 1, 1, 0, 1), (128, 56, 128, 3, 2, 1, 1), (128, 28, 512, 1, 1, 0, 4), (256, 56, 512, 1, 2, 0, 1),
          (512, 28, 128, 1, 1, 0, 3), (128, 28, 128, 3, 1, 1, 3),
          (512, 28, 256, 1, 1, 0, 1), (256, 28, 256, 3, 2, 1, 1), (256, 14, 1024, 1, 1, 0, 6), (512, 28, 1024, 1, 2, 0, 1),
          (1024, 14, 256, 1, 1, 0, 5), (256, 14, 256, 3, 1, 1, 5),
          (1024, 14, 512, 1, 1, 0, 1), (512, 14, 512, 3, 2, 1, 1), (512, 7, 2048, 1, 1, 0, 3), (1024, 14, 2048, 1, 2, 0, 1),
          (2048, 7, 512, 1, 1, 0, 2), (512, 7, 512, 3, 1, 1, 2)]


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


tot_c = tot_q = 0.0
print("%-34s %9s %9s %8s %8s | %9s %8s" % ("layer C,H,K,R,s", "conv us", "TOP/s", "out GB/s", "floor us", "quant us", "GB/s"))
for (C, H, K, R, st, pd, cnt) in LAYERS:
    x = torch.randn(B, C, H, H, device="cuda") * 2
    w = torch.randint(-127, 128, (K, C, R, R), device="cuda").float()
    qb = torch.randint(-100, 100, (K,), device="cuda").float()
    wq = nat.pack_weight_krsc(w)
    xq = nat.quantize_i8_nhwc(x, 4, wq.shape[-1])
    P = (H + 2 * pd - R) // st + 1
    if C <= 4:
        # the stem as the model runs it: fq_conv2d_i8_stem from the fp32 image (int8 output), or the unfolded copy +
        # the general kernel (fp32 output); the "quant" column is the unfold kernel of the latter
        fold = nat.pad16(R * C)
        w_fold, w_stem = nat.pack_weight_unfold_w(w, fold), nat.pack_weight_stem(w)
        xu = nat.quantize_i8_unfold_w(x, 4, R, st, pd, 1, fold)
        if I8OUT:
            t_c = timeit(lambda: nat.conv2d_i8_stem(x, w_stem, qb, K, R, (st, st), (pd, pd), 4, 8, 4, True))
        else:
            t_c = timeit(lambda: nat.conv2d_i8(xu, w_fold, qb, (st, 1), (pd, 0), (1, 1), 8, 4))
        t_q = 0.0 if I8OUT else timeit(lambda: nat.quantize_i8_unfold_w(x, 4, R, st, pd, 1, fold))
        xq = x if I8OUT else xu
    elif I8OUT:
        t_c = timeit(lambda: nat.conv2d_i8_resident(xq, wq, qb, (st, st), (pd, pd), (1, 1), 8, 4, False, True, True))
    else:
        t_c = timeit(lambda: nat.conv2d_i8(xq, wq, qb, (st, st), (pd, pd), (1, 1), 8, 4))
    if C > 4:
        t_q = timeit(lambda: nat.quantize_i8_nhwc(x, 4, wq.shape[-1]))
    macs = B * P * P * K * C * R * R
    out_b = B * P * P * K * (1 if I8OUT else 4)
    in_b = xq.numel() * xq.element_size()
    floor = (out_b + in_b) / 5.0e12 * 1e6
    print("%-34s %9.1f %9.1f %8.0f %8.1f | %9.1f %8.0f   x%d" % ("%d,%d,%d,%d,%d" % (C, H, K, R, st), t_c, 2 * macs / t_c / 1e6,
                                                                 out_b / t_c / 1e3, floor, t_q, (x.numel() * 4 + xq.numel()) / max(t_q, 1e-9) / 1e3, cnt))
    tot_c += t_c * cnt
    tot_q += t_q * cnt
print("whole net per batch of %d: conv %.2f ms, quantize %.2f ms" % (B, tot_c / 1e3, tot_q / 1e3))
