#!/usr/bin/env python3
"""Kernel-level microbenchmark (GPU box): achieved algorithmic GB/s of the calibration and
fake-quant kernels on R50-shaped segment lists.  Not the contract bench (that is bench.py)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat  # noqa: E402


def timeit(fn, iters=20, warmup=3, burst=5):
    """(median, best) ms per call.  Calls are timed in back-to-back bursts behind a queued blocker, so the host's
    argument marshalling for call k+1 overlaps the GPU running call k and is not counted as kernel time."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(max(1, iters // burst)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000)                     # ~1 ms of GPU idle-spin: the burst queues up behind it
        a.record()
        for _ in range(burst):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / burst)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


CARVED = "--carved" in sys.argv          # all segments are views into ONE allocation (large page-table fragments)


def r50_like_segments(batch, dist="normal"):
    # cared-tensor sizes of a fabu ResNet-50 @224 (elements per image)
    sizes = [150528, 802816] + [200704, 200704, 802816, 802816, 802816] + [200704, 200704, 802816, 802816] * 2 \
        + [401408, 100352, 401408, 401408, 401408] + [100352, 100352, 401408, 401408] * 3 \
        + [200704, 50176, 200704, 200704, 200704] + [50176, 50176, 200704, 200704] * 5 \
        + [100352, 25088, 100352, 100352, 100352] + [25088, 25088, 100352, 100352] * 2 + [1000]
    segs = []
    if CARVED:
        pool = torch.empty(sum((s * batch + 63) // 64 * 64 for s in sizes), device="cuda")
        off = 0
        for i, s in enumerate(sizes):
            n = s * batch
            t = pool[off:off + n]
            t.normal_()
            t.mul_(1.0 + (i % 5))
            if dist == "relu":
                t.relu_()
            segs.append(t)
            off += (n + 63) // 64 * 64
        return segs
    for i, s in enumerate(sizes):
        n = s * batch
        if dist == "normal":
            t = torch.randn(n, device="cuda") * (1.0 + (i % 5))
        elif dist == "outlier":
            t = torch.randn(n, device="cuda")
            t[0] = 60.0
        else:
            t = torch.relu(torch.randn(n, device="cuda"))
        segs.append(t)
    return segs


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
    print("fq version", nat.lib().fq_version(), "device", torch.cuda.get_device_name(0))
    for dist in ("normal", "outlier", "relu"):
        segs = r50_like_segments(batch, dist)
        rows = list(range(len(segs)))
        nel = sum(s.numel() for s in segs)
        mx = torch.zeros(len(segs), device="cuda")
        ms, best = timeit(lambda: nat.absmax_seg(segs, rows, mx))
        print("[%s] absmax_seg  : %d segs %.1f Melem  med %.3f ms  (%.0f GB/s, best %.0f)" %
              (dist, len(segs), nel / 1e6, ms, nel * 4 / ms / 1e6, nel * 4 / best / 1e6))
        iv = (mx / 2048 + 1e-12).float()
        hist = torch.zeros(len(segs), 2048, dtype=torch.int64, device="cuda")
        ms, best = timeit(lambda: nat.hist2048_seg(segs, rows, iv, hist))
        print("[%s] hist2048_seg: med %.3f ms  (%.0f GB/s, best %.0f)" % (dist, ms, nel * 4 / ms / 1e6, nel * 4 / best / 1e6))
        if dist == "normal":
            t0 = time.perf_counter()
            thr = nat.kl_threshold(hist)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ms, best = timeit(lambda: nat.kl_threshold(hist), iters=5, warmup=1)
            print("kl_threshold rows=%d: first %.1f ms, med %.2f ms; thr[:8]=%s" %
                  (len(segs), (t1 - t0) * 1e3, ms, thr[:8].tolist()))
        del segs
    x = torch.randn(128 * 802816, device="cuda")
    y = torch.empty_like(x)
    ms, best = timeit(lambda: nat.quandequan(x, 4, out=y))
    print("quandequan   : %.1f Melem med %.3f ms (%.0f GB/s rd+wr, best %.0f)" %
          (x.numel() / 1e6, ms, x.numel() * 8 / ms / 1e6, x.numel() * 8 / best / 1e6))
    ms, best = timeit(lambda: nat.add_sat(x, y, out=y))
    print("add_sat      : med %.3f ms (%.0f GB/s 2rd+wr)" % (ms, x.numel() * 12 / ms / 1e6))
    ms, best = timeit(lambda: torch.clamp(torch.round(x * 16), -128, 127) / 16)
    print("torch 4-op quandequan reference: med %.3f ms" % ms)
    ms, best = timeit(lambda: y.copy_(x))
    print("torch copy   : med %.3f ms (%.0f GB/s rd+wr)" % (ms, x.numel() * 8 / ms / 1e6))


def bench_channel_kernels(batch=128):
    """fq_absmax_chan / fq_hist2048_chan on ResNet-50-shaped NCHW tensors (one row per channel)."""
    shapes = [(3, 224)] + [(64, 112)] + [(64, 56)] * 7 + [(256, 56)] * 7 + [(128, 56)] + [(128, 28)] * 8 + [(512, 28)] * 9 + \
             [(256, 28)] + [(256, 14)] * 12 + [(1024, 14)] * 13 + [(512, 14)] + [(512, 7)] * 6 + [(2048, 7)] * 7
    ts = [torch.randn(batch, c, h, h, device="cuda") for c, h in shapes]
    row0, r = [], 0
    for c, _ in shapes:
        row0.append(r); r += c
    elems = sum(t.numel() for t in ts)
    mx = torch.zeros(r, device="cuda")
    ms, best = timeit(lambda: nat.absmax_chan(ts, row0, mx), iters=10)
    print("absmax_chan  : %d tensors %d rows %.1f Melem  med %.3f ms  (%.0f GB/s, best %.0f)" %
          (len(ts), r, elems / 1e6, ms, elems * 4 / ms / 1e6, elems * 4 / best / 1e6))
    iv = (mx / 2048 + 1e-12).float()
    hist = torch.zeros(r, 2048, dtype=torch.int64, device="cuda")
    ms, best = timeit(lambda: nat.hist2048_chan(ts, row0, iv, hist), iters=10)
    print("hist2048_chan: med %.3f ms  (%.0f GB/s, best %.0f)" % (ms, elems * 4 / ms / 1e6, elems * 4 / best / 1e6))


def bench_channel_shapes(batch=128):
    """The per-channel kernels one shape class at a time (enough copies of the tensor for >= 1 GB per launch)."""
    for c, h in [(3, 224), (64, 112), (64, 56), (256, 56), (128, 56), (128, 28), (512, 28), (256, 28), (256, 14), (1024, 14),
                 (512, 14), (512, 7), (2048, 7)]:
        per = batch * c * h * h * 4
        copies = max(1, min(16, (1 << 30) // per))
        ts = [torch.randn(batch, c, h, h, device="cuda") for _ in range(copies)]
        row0 = [i * c for i in range(copies)]
        rows = copies * c
        mx = torch.zeros(rows, device="cuda")
        ms_a, _ = timeit(lambda: nat.absmax_chan(ts, row0, mx), iters=8)
        iv = (mx / 2048 + 1e-12).float()
        hist = torch.zeros(rows, 2048, dtype=torch.int64, device="cuda")
        ms_h, _ = timeit(lambda: nat.hist2048_chan(ts, row0, iv, hist), iters=8)
        flat = [t.view(-1) for t in ts]
        mx1 = torch.zeros(copies, device="cuda")
        ms_s, _ = timeit(lambda: nat.hist2048_seg(flat, list(range(copies)), (mx1 + 1).float(), hist[:copies]), iters=8)
        gb = per * copies / 1e6
        print("C=%4d HW=%5d x%2d (%.2f GB): absmax_chan %.0f GB/s  hist2048_chan %.0f GB/s  [per-tensor hist %.0f GB/s]" %
              (c, h * h, copies, gb / 1e3, gb / ms_a, gb / ms_h, gb / ms_s))


def bench_rotating(batch, sets):
    """The same segment list in `sets` different places of HBM, visited in turn: every launch reads addresses
    it has not touched for sets-1 launches (cold TLB / page-table walks), as in a real calibration pass."""
    groups = [r50_like_segments(batch, "normal") for _ in range(sets)]
    rows = list(range(len(groups[0])))
    nel = sum(t.numel() for t in groups[0])
    mx = torch.zeros(len(rows), device="cuda")
    hist = torch.zeros(len(rows), 2048, dtype=torch.int64, device="cuda")
    state = {"i": 0}

    def nxt():
        state["i"] = (state["i"] + 1) % sets
        return groups[state["i"]]
    ms, best = timeit(lambda: nat.absmax_seg(nxt(), rows, mx))
    print("absmax_seg  rotating over %d sets (%.0f GB): med %.3f ms (%.0f GB/s, best %.0f)" %
          (sets, sets * nel * 4 / 1e9, ms, nel * 4 / ms / 1e6, nel * 4 / best / 1e6))
    iv = (mx / 2048 + 1e-12).float()
    ms, best = timeit(lambda: nat.hist2048_seg(nxt(), rows, iv, hist))
    print("hist2048_seg rotating: med %.3f ms (%.0f GB/s, best %.0f)" % (ms, nel * 4 / ms / 1e6, nel * 4 / best / 1e6))


def bench_after_compute(batch):
    """One statistics launch right behind ~20 ms of fp32 matrix work, as in a calibration pass (the forward)."""
    segs = r50_like_segments(batch, "normal")
    rows = list(range(len(segs)))
    nel = sum(t.numel() for t in segs)
    mx = torch.zeros(len(rows), device="cuda")
    nat.absmax_seg(segs, rows, mx)
    iv = (mx / 2048 + 1e-12).float()
    hist = torch.zeros(len(rows), 2048, dtype=torch.int64, device="cuda")
    a_mat = torch.randn(8192, 8192, device="cuda")
    def rewrite():
        for t in segs:
            t.mul_(1.0)                                 # rewrites every element: the data is "fresh" like after a forward
    for label, heat in (("idle GPU", 0), ("behind 4 fp32 8192^3 matmuls", 4), ("behind 16", 16), ("behind a rewrite of the data", -1),
                        ("rewrite, then reversed segment order", -2)):
        for name, fn in (("absmax_seg", lambda: nat.absmax_seg(segs, rows, mx)), ("hist2048_seg", lambda: nat.hist2048_seg(segs, rows, iv, hist))):
            ts = []
            for _ in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(2_000_000)
                for _ in range(max(heat, 0)):
                    torch.mm(a_mat, a_mat)
                if heat < 0:
                    rewrite()
                if heat == -2:
                    rsegs, rrows = segs[::-1], rows[::-1]
                    fn = (lambda: nat.absmax_seg(rsegs, rrows, mx)) if name == "absmax_seg" else (lambda: nat.hist2048_seg(rsegs, rrows, iv, hist))
                a.record(); fn(); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            ts.sort()
            print("%-12s %-30s med %.3f ms (%.0f GB/s)" % (name, label, ts[len(ts) // 2], nel * 4 / ts[len(ts) // 2] / 1e6))


def bench_real_activations(batch):
    """The statistics kernels on the cared activations of one real forward of the bench's ResNet-50 (their value
    distribution, their allocation pattern), timed in bursts like the synthetic segments."""
    sys.path.insert(0, ROOT)
    import bench
    from tools import Quantity
    out = sys.stdout; sys.stdout = open(os.devnull, "w")
    model = bench.build_model("r50", 224, torch.device("cuda"))
    bench.make_workdir(1, "1,3,224,224", 0)
    q = Quantity(model)
    feats, hooks = q.regist_hook_outfeature(model)
    with torch.no_grad():
        model(torch.randn(batch, 3, 224, 224, device="cuda"))
    sys.stdout = out
    names = ["image"] + list(q.net_info.keys())
    segs = [nat.dense_view(feats[n]).reshape(-1) for n in names]
    rows = list(range(len(segs)))
    nel = sum(t.numel() for t in segs)
    mx = torch.zeros(len(segs), device="cuda")
    ms, best = timeit(lambda: nat.absmax_seg(segs, rows, mx))
    print("[real] absmax_seg  : %d segs %.1f Melem  med %.3f ms  (%.0f GB/s, best %.0f)" % (len(segs), nel / 1e6, ms, nel * 4 / ms / 1e6, nel * 4 / best / 1e6))
    iv = (mx / 2048 + 1e-12).float()
    hist = torch.zeros(len(segs), 2048, dtype=torch.int64, device="cuda")
    ms, best = timeit(lambda: nat.hist2048_seg(segs, rows, iv, hist))
    print("[real] hist2048_seg: med %.3f ms  (%.0f GB/s, best %.0f)" % (ms, nel * 4 / ms / 1e6, nel * 4 / best / 1e6))
    h = hist.cpu().numpy().astype(float)
    occ = [(r > 0).sum() for r in h]
    top = [r.max() / max(r.sum(), 1) for r in h]
    print("[real] non-empty bins per row: min %d median %d; share of the fullest bin: median %.3f max %.3f" %
          (min(occ), sorted(occ)[len(occ) // 2], sorted(top)[len(top) // 2], max(top)))


def bench_per_tensor_launches(batch):
    """Would launching the statistics kernel per tensor, right behind the kernel that wrote it (while it is still in the
    Infinity Cache), beat one launch over all tensors after the whole forward?  Producer = an in-place rewrite."""
    segs = r50_like_segments(batch, "normal")
    rows = list(range(len(segs)))
    mx = torch.zeros(len(rows), device="cuda")
    nat.absmax_seg(segs, rows, mx)
    iv = (mx / 2048 + 1e-12).float()
    hist = torch.zeros(len(rows), 2048, dtype=torch.int64, device="cuda")

    def produce_only():
        for t in segs:
            t.mul_(1.0)

    def one_launch(fn):
        produce_only()
        fn(segs, rows)

    def per_tensor(fn):
        for i, t in enumerate(segs):
            t.mul_(1.0)
            fn([t], [i])

    def per_group(fn, limit=96 << 20):
        acc, ids, size = [], [], 0
        for i, t in enumerate(segs):
            t.mul_(1.0)
            acc.append(t); ids.append(i); size += t.numel() * 4
            if size >= limit:
                fn(acc, ids); acc, ids, size = [], [], 0
        if acc:
            fn(acc, ids)
    base, _ = timeit(produce_only, iters=10, burst=2)
    for name, fn in (("absmax", lambda s, r: nat.absmax_seg(s, r, mx)), ("hist", lambda s, r: nat.hist2048_seg(s, r, iv, hist))):
        a, _ = timeit(lambda: one_launch(fn), iters=10, burst=2)
        b, _ = timeit(lambda: per_tensor(fn), iters=10, burst=2)
        c, _ = timeit(lambda: per_group(fn), iters=10, burst=2)
        print("%-6s producer alone %.3f ms | + one launch after all %.3f | + one launch per tensor %.3f | + one launch per ~96 MB %.3f  (ms added)" %
              (name, base, a - base, b - base, c - base))


def bench_single_segment():
    """One 8 GiB segment: the statistics kernels without the multi-segment tiling."""
    x = torch.randn(1 << 31, device="cuda")
    mx = torch.zeros(1, device="cuda")
    ms, best = timeit(lambda: nat.absmax_seg([x], [0], mx))
    print("absmax_seg one 8 GiB segment : med %.3f ms (%.0f GB/s, best %.0f)" % (ms, x.numel() * 4 / ms / 1e6, x.numel() * 4 / best / 1e6))
    iv = (mx / 2048 + 1e-12).float()
    hist = torch.zeros(1, 2048, dtype=torch.int64, device="cuda")
    ms, best = timeit(lambda: nat.hist2048_seg([x], [0], iv, hist))
    print("hist2048_seg one 8 GiB segment: med %.3f ms (%.0f GB/s, best %.0f)" % (ms, x.numel() * 4 / ms / 1e6, x.numel() * 4 / best / 1e6))
    x.fill_(0.75)
    ms, best = timeit(lambda: nat.absmax_seg([x], [0], mx))
    print("absmax_seg, constant data     : med %.3f ms (%.0f GB/s, best %.0f)" % (ms, x.numel() * 4 / ms / 1e6, x.numel() * 4 / best / 1e6))


if __name__ == "__main__":
    if "--per-tensor" in sys.argv:
        bench_per_tensor_launches(int(sys.argv[1]))
    elif "--real" in sys.argv:
        bench_real_activations(int(sys.argv[1]))
    elif "--heat" in sys.argv:
        bench_after_compute(int(sys.argv[1]))
    elif "--rotate" in sys.argv:
        bench_rotating(int(sys.argv[1]), int(sys.argv[sys.argv.index("--rotate") + 1]))
    elif "--single" in sys.argv:
        bench_single_segment()
    elif "--chan-shapes" in sys.argv:
        bench_channel_shapes()
    elif "--chan" in sys.argv:
        bench_channel_kernels()
    else:
        main()
