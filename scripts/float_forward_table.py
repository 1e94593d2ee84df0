"""GPU box: one row per kernel launch of ONE calibration pass-1 forward (fabu ResNet-50 @224, default 256 images, nothing cached):
microseconds (HIP events on the launch stream), TFLOP/s, GB/s of algorithmic traffic.  usage: python scripts/float_forward_table.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
os.environ.setdefault("FQ_ACT_CACHE_GB", "0")
import bench  # noqa: E402
from common.quantity import _native  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda", 0)
    sys.stdout, out = open(os.devnull, "w"), sys.stdout
    model = bench.build_model("r50", 224, dev)
    sys.stdout = out
    names = ["conv_stem_f32", "conv1x1_f32", "conv_kxk_f32", "conv_wino_f32", "conv1x1_add_f32", "conv1x1_add_hist_f32", "add_absmax", "add_hist",
             "bias_add_absmax", "maxpool2d_f32", "avgpool_global_f32", "absmax_seg", "hist2048_seg"]
    rows, on = [], {"v": False}

    def wrap(name):
        orig = getattr(_native, name)

        def f(*a, **k):
            if not on["v"]:
                return orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            x = a[0]
            fl = by = 0.0
            desc = ""
            if name in ("conv1x1_f32", "conv1x1_add_f32", "conv1x1_add_hist_f32"):
                fl, by = bench._c1_flops(a, k), (bench._c1_add_bytes(a, k) if "add" in name else bench._c1_bytes(a, k))
                desc = "%dx%dx%dx%d -> %d s%d" % (tuple(x.shape) + (a[1].shape[1], a[3]))
            elif name == "conv_kxk_f32":
                fl, by = bench._kxk_flops(a, k), bench._kxk_bytes(a, k)
                desc = "%dx%dx%dx%d -> %d k%d s%d" % (tuple(x.shape) + (a[1].shape[1], a[3][0], a[4]))
            elif name == "conv_wino_f32":                       # (TFLOP/s: the direct sum's flop count over the kernel's time)
                fl, by = bench._wino_direct_flops(a, k), bench._wino_bytes(a, k)
                desc = "%dx%dx%dx%d -> %d k3 s1 winograd" % (tuple(x.shape) + (a[3],))
            elif name == "conv_stem_f32":
                fl, by = bench._stem_flops(a, k), bench._stem_bytes(a, k)
                desc = "%dx%dx%dx%d -> %d" % (tuple(x.shape) + (a[3],))
            elif name in ("add_absmax", "add_hist"):
                by = bench._add_bytes(a, k)
                desc = "x".join(map(str, x.shape))
            elif torch.is_tensor(x):
                by = 8.0 * x.numel()
                desc = "x".join(map(str, x.shape))
            else:
                by = 4.0 * sum(t.numel() for t in x)
                desc = "%d tensors" % len(x)
            rows.append((name, desc, e0, e1, fl, by, "+relu" if k.get("relu_out") is not None else ""))
            return r
        setattr(_native, name, f)
    for n in names:
        wrap(n)
    from tools import Quantity
    bench.make_workdir(2, "1,3,224,224", 0)
    data = bench.DeviceBatches(3, B, 224, 0, 1, dev)
    q = Quantity(model)
    real = q._forward_with_stats
    count = {"n": 0}

    def fws(item, fn, feats, extra=None):
        count["n"] += 1
        on["v"] = count["n"] == 3                 # the third forward of pass 1: every module checked, everything fused
        try:
            return real(item, fn, feats, extra)
        finally:
            on["v"] = False
    q._forward_with_stats = fws
    sys.stdout, out = open(os.devnull, "w"), sys.stdout
    q.activation_quantize(data)
    sys.stdout = out
    torch.cuda.synchronize()
    tot = 0.0
    print("%-3s %-22s %-34s %9s %8s %8s" % ("#", "call", "shape", "us", "TFLOP/s", "GB/s"))
    for i, (name, desc, e0, e1, fl, by, tag) in enumerate(rows):
        us = e0.elapsed_time(e1) * 1e3
        tot += us
        print("%-3d %-22s %-34s %9.1f %8.1f %8.0f" % (i, name + tag, desc, us, fl / us / 1e6, by / us / 1e3))
    print("one pass-1 forward of %d images: %d launches, %.3f ms in kernels" % (B, len(rows), tot / 1e3))


if __name__ == "__main__":
    main()
