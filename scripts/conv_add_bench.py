"""GPU box: conv3 + Eltwise + ReLU of a residual block as two kernels (fq_conv1x1_f32 max form, then fq_add_absmax_f32) against
the one-kernel form (fq_conv1x1_add_f32), per ResNet-50 stage at 256 images.  usage: python scripts/conv_add_bench.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pytorch-quantity_amd", "quantity")]
from common.quantity import _native as nat  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tot = [0.0, 0.0, 0.0, 0.0]
    for (cin, cout, hw, blocks) in ((64, 256, 56, 3), (128, 512, 28, 4), (256, 1024, 14, 6), (512, 2048, 7, 3)):
        x = torch.randn(B, cin, hw, hw, device="cuda")
        wt = torch.randn(cin, cout, device="cuda") * cin ** -0.5
        b = torch.randn(cout, device="cuda")
        res = torch.randn(B, cout, hw, hw, device="cuda")
        m = torch.zeros(4, device="cuda")
        y, s, r = torch.empty_like(res), torch.empty_like(res), torch.empty_like(res)

        def two():
            nat.conv1x1_f32(x, wt, b, 1, max_dev=m, row=0, out=y)
            nat.add_absmax(y, res, m, 1, out=s, relu_out=r)
        t2 = timed(two)
        t_none = timed(lambda: nat.conv1x1_add_f32(x, wt, b, 1, res, m, 0, 1, r))
        t_y = timed(lambda: nat.conv1x1_add_f32(x, wt, b, 1, res, m, 0, 1, r, out=y))
        t_all = timed(lambda: nat.conv1x1_add_f32(x, wt, b, 1, res, m, 0, 1, r, out=y, sum_out=s))
        gb = y.numel() * 4 / 1e9
        print("%4d->%4d @%2d x%d: two kernels %7.1f us | one kernel: nothing kept %7.1f us (%.2f TB/s), conv kept %7.1f, both kept %7.1f"
              % (cin, cout, hw, blocks, t2, t_none, 2 * gb / t_none * 1e3, t_y, t_all))
        for i, t in enumerate((t2, t_none, t_y, t_all)):
            tot[i] += t * blocks
    print("all 16 blocks: two kernels %.2f ms; one kernel %.2f / %.2f / %.2f ms" % tuple(t / 1e3 for t in tot))


if __name__ == "__main__":
    main()
