#!/usr/bin/env python3
"""Headline benchmark: KL calibration throughput of a fabu ResNet-50 @224^2 on MI355X
(BASELINE.json metric "calibration images/sec + int8-sim images/sec", config[1] at N=1).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one calibration batch taken through the whole hot path: forward -> segmented abs-max
(pass 1), forward -> 2048-bin histograms (pass 2); the timed region is ONE complete
Quantity.activation_quantize() over K batches per GPU: both passes, the MAX / SUM all-reduces, the
KL threshold sweep for all 71 tensors and the feat.table write.  Inputs are synthetic fp32 images
generated on the device BEFORE the timed region.  value = images processed by all ranks / wall time
(max over ranks, bracketed by barrier + synchronize).  Weak scaling: K batches per GPU.

The JSON line also carries
  roofline     : the dominant hand-written kernel (hist2048_seg), algorithmic bytes (4 B x elements
                 per launch) / its mean launch duration measured with HIP events on the launch stream;
  cpu_baseline : the CPU oracle (oracle/fq_oracle.c, "port") + torch-CPU forwards timed on a bounded
                 sample on this box's host cores, scaled to the same workload (rank 0, N=1 only);
  (the forward-throughput keys below run --int8-batch images per forward, default 256: two calibration batches)
  int8_sim_images_per_s : ReconModel forward throughput with resident integer activations
                     (common.quantity.resident.enable(): int8/int16 NHWC between layers; logits checked
                     bit-identical to the fp32-boundary model in the same run, see int8_sim_resident);
  int8_sim_fp32_boundary_images_per_s : the same ReconModel with the reference's fp32 NCHW tensor at
                     every module boundary (the drop-in default);
  int8_sim_hipgraph_images_per_s : the resident forward replayed as one HIP graph (input copy included);
  roofline_int8_conv : 2 x MACs of the model / summed durations of the int8 conv launches of one resident forward
                     (HIP events), against the 5 000 TOP/s dense int8 MFMA peak;
  fakequant_images_per_s / float_forward_images_per_s : ReconTest and the float model.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
QUANTITY = os.path.join(ROOT, "pytorch-quantity_amd", "quantity")
sys.path.insert(0, QUANTITY)
sys.path.insert(0, ROOT)

with open(os.path.join(ROOT, "BASELINE.json")) as _fh:
    BASELINE_METRIC = json.load(_fh)["metric"]          # the reference's headline metric, verbatim
R50_CARED_ELEMS_PER_IMAGE = 16784872        # image + 53 conv + fc + 16 Eltwise outputs @224^2
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


class DeviceBatches(object):
    """Sequence of (images, None) pairs indexed by GLOBAL batch index; only this rank's batches
    (i % world == rank) exist, pre-generated in HBM from generator seed 1234 + i."""

    def __init__(self, total, batch, hw, rank, world, device, on_host=False):
        self._items = {}
        self._total = total
        for i in range(rank, total, world):
            g = torch.Generator(device=device).manual_seed(1234 + i)
            t = torch.randn(batch, 3, hw, hw, generator=g, device=device)
            self._items[i] = t.cpu() if on_host else t       # --host-inputs: pageable host memory

    def __len__(self):
        return self._total

    def __getitem__(self, i):
        return self._items[i], None

    def owned(self):
        return list(self._items.values())


def make_workdir(max_cali_img_num, input_shape, gpu):
    import yaml
    tmp = tempfile.mkdtemp(prefix="fq_bench_")
    os.makedirs(os.path.join(tmp, "tools"))
    os.makedirs(os.path.join(tmp, "test"))
    with open(os.path.join(QUANTITY, "tools", "configs.yml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg["SETTINGS"]["MAX_CALI_IMG_NUM"] = max_cali_img_num
    with open(os.path.join(tmp, "tools", "configs.yml"), "w") as fh:
        yaml.safe_dump(cfg, fh)
    with open(os.path.join(QUANTITY, "test", "user_configs.yml")) as fh:
        ucfg = yaml.safe_load(fh)
    ucfg["MODEL"]["INPUT_SHAPE"] = input_shape
    ucfg["SETTINGS"]["DEVICE"] = "gpu"
    ucfg["SETTINGS"]["GPU"] = gpu
    with open(os.path.join(tmp, "test", "user_configs.yml"), "w") as fh:
        yaml.safe_dump(ucfg, fh)
    os.chdir(os.path.join(tmp, "test"))
    return tmp


def build_model(name, hw, device):
    from common.quantity import merge_bn
    from model.resnet.ResNet_fabu import ResNet50, ResNet101
    torch.manual_seed(0)
    model = (ResNet50 if name == "r50" else ResNet101)(input_size=hw)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    model.eval()
    model = merge_bn(model)
    return model.to(device)


class KernelTimer(object):
    """Wraps a _native entry point and brackets every call with HIP events on the launch stream
    (torch's current stream is the stream the C ABI is handed)."""

    def __init__(self, native, name):
        self.native, self.name = native, name
        self.orig = getattr(native, name)
        self.events = []
        self.elems = []
        self.enabled = False

    def __enter__(self):
        def wrapped(tensors, rows, *rest):
            if not self.enabled:
                return self.orig(tensors, rows, *rest)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = self.orig(tensors, rows, *rest)
            b.record()
            self.events.append((a, b))
            self.elems.append(sum(int(t.numel()) for t in tensors))
            return r
        setattr(self.native, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.native, self.name, self.orig)

    def summary(self):
        if not self.events:
            return None
        ms = [a.elapsed_time(b) for a, b in self.events]
        return {"launches": len(ms), "mean_ms": float(np.mean(ms)), "bytes_per_launch": 4.0 * float(np.mean(self.elems))}


def int8_conv_roofline(float_model, int8_net, batch, forwards=3):
    """MFMA-side roofline of the integer contraction (the one MFMA kernel on the path): 2 x MACs of every Conv2d / Linear
    of the model per forward / the summed durations of the int8 conv launches of the resident forward, HIP events on
    the launch stream around every C-ABI call.  Peak: 5 000 TOP/s dense int8 (2 x the BF16 MFMA rate, MI355X_MICROARCH.md)."""
    from common.quantity import _native
    macs, hooks = [0], []

    def count(m, i, o):
        w = m.weight
        macs[0] += int(o.numel()) * int(w[0].numel())                 # outputs x (C_in / groups x kh x kw)
    for m in float_model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            hooks.append(m.register_forward_hook(count))
    with torch.no_grad():
        float_model(batch)
    for h in hooks:
        h.remove()
    names = ("conv2d_i8_resident", "conv2d_i8_add_resident", "conv2d_i8_stem", "conv2d_i8")
    events, saved = [], {n: getattr(_native, n) for n in names}

    def timed(fn):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            events.append((e0, e1))
            return r
        return wrapper
    try:
        for n in names:
            setattr(_native, n, timed(saved[n]))
        with torch.no_grad():
            for _ in range(forwards):
                int8_net(batch)
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(_native, n, saved[n])
    ms = sum(a.elapsed_time(b) for a, b in events) / forwards
    achieved = 2.0 * macs[0] / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "conv2d_i8 / conv2d_i8_dma / stem_conv_i8 (all integer conv + linear launches of one forward)",
            "achieved": round(achieved, 1), "peak": 5000.0, "unit": "TOP/s", "frac": round(achieved / 5000.0, 4),
            "launches_per_forward": len(events) // forwards, "ms_per_forward": round(ms, 4), "images_per_forward": int(batch.shape[0]),
            "gmac_per_image": round(macs[0] / int(batch.shape[0]) / 1e9, 3),
            "note": "latency / output-traffic bound at these layer sizes (DESIGN.md 5b); SQ_VALU_MFMA_BUSY_CYCLES 21 % on the "
                    "3x3 256->256 14x14 layer (profiles/r01j_conv_pmc_256x14x14_3x3.txt)"}


def cpu_baseline(model_cpu_ctor, sample_images, hw, n_images_full, q, log):
    """Bounded CPU run of the same workload: torch-CPU forward (what the reference does) + the CPU
    oracle for abs-max / histogram / KL, scaled to the full image count.  Reported, not a target."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import fq_oracle as orc
    orc.build()
    cores = min(os.cpu_count() or 1, 32)          # beyond one socket's worth oneDNN only gets slower
    torch.set_num_threads(cores)
    model = model_cpu_ctor()
    names = ["image"] + list(q.net_info.keys())
    feats, hooks = {}, []
    state = {"n": 0}
    cared = set(q.net_info.keys())

    def hook(m, i, o):
        if state["n"] == 0:
            feats.clear()
            feats["image"] = i[0]
        state["n"] += 1
        k = "%s_%i" % (type(m).__name__, state["n"])
        if k in cared:
            feats[k] = o
        if state["n"] >= q.layers_num:
            state["n"] = 0

    for m in model.modules():
        if type(m).__name__ in q._all_op_type:
            hooks.append(m.register_forward_hook(hook))
    x = torch.randn(sample_images, 3, hw, hw, generator=torch.Generator().manual_seed(1234))
    pool = ThreadPoolExecutor(cores)
    with torch.no_grad():
        model(x[:1])                                                 # untimed: oneDNN primitive creation
    t0 = time.perf_counter()
    with torch.no_grad():
        model(x)                                                     # pass 1 forward
    arrs = {n: feats[n].numpy().ravel() for n in names}
    maxs = dict(zip(names, pool.map(lambda n: orc.absmax(arrs[n]), names)))
    ivs = {n: orc.interval(maxs[n]) for n in names}
    with torch.no_grad():
        model(x)                                                     # pass 2 forward
    arrs = {n: feats[n].numpy().ravel() for n in names}
    hists = dict(zip(names, pool.map(lambda n: orc.hist2048(arrs[n], ivs[n]), names)))
    t_img = (time.perf_counter() - t0) / sample_images
    t1 = time.perf_counter()
    list(pool.map(lambda n: orc.kl_threshold(orc.normalize(hists[n])), names))
    t_kl = time.perf_counter() - t1
    for h in hooks:
        h.remove()
    value = n_images_full / (n_images_full * t_img + t_kl)
    log("cpu_baseline: %.3f s/image (2 forwards + absmax + hist), KL %.2f s for %d tensors" % (t_img, t_kl, len(names)))
    return {"value": round(value, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d images through torch-CPU forward x2 + oracle absmax/hist2048, + oracle KL sweep of all %d "
                      "tensors once; scaled to %d images" % (sample_images, len(names), n_images_full)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)      # 40 x 128 = 5 120 images: BASELINE configs[1] ("5k")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--model", default="r50", choices=["r50", "r101"])
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--host-inputs", action="store_true",
                    help="hand the calibrator pageable HOST batches (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recon", action="store_true")
    ap.add_argument("--int8-batch", type=int, default=256,
                    help="images per forward of the int8-sim / fake-quant throughput section (resident int8-sim forward on "
                         "one MI355X: 62 k images/s at 128, 73 k at 256, 76 k at 512)")
    ap.add_argument("--no-per-channel", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON.  Native libraries write there too (RCCL prints a version
    # banner on init), so file descriptor 1 itself is parked on /dev/null until the result is ready.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    null_fd = os.open(os.devnull, os.O_WRONLY)
    os.dup2(null_fd, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)                  # backend nccl = RCCL on ROCm

    def log(*a):
        if rank == 0:
            print(*a, file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    torch.backends.cudnn.benchmark = os.environ.get("FQ_BENCH_MIOPEN_FIND", "0") == "1"   # default: MIOpen immediate mode (find mode measured: see DESIGN.md)
    from common.quantity import _native
    from tools import Quantity, Reconstruction
    _native.lib()

    K, W, B, HW = args.steps, args.warmup, args.batch, args.image
    shape = "1,3,%d,%d" % (HW, HW)
    devnull = open(os.devnull, "w")
    real_stdout = sys.stdout
    sys.stdout = devnull                                   # the drop-in prints like the reference does (merge_bn too):
    model = build_model(args.model, HW, device)            # stdout must carry exactly one JSON line

    # ---- warmup: W batches per GPU through the same path (MIOpen algo search, RCCL init, code load)
    if W > 0:
        make_workdir(W * world - 1, shape, local_rank)
        warm = DeviceBatches(W * world, B, HW, rank, world, device)
        wq = Quantity(model)
        wq.activation_quantize(warm)
        del warm
        # Size the caching allocator's pool for the device before the clock starts: 80 % of HBM (288 GB per MI355X),
        # as a long-running calibration service would hold it.  The timed region then keeps pass 1's activations for
        # pass 2 in that pool (phases_s.cache_bytes) instead of paying 10-30 ms per GB of fresh hipMalloc; a cold
        # one-shot process keeps the engine's 96 GB rule (tools/pytorch_quantizer.py:_activation_cache_budget).
        grow = wq._activation_cache_budget()
        if grow > 0 and "FQ_ACT_CACHE_GB" not in os.environ:
            free_b, total_b = torch.cuda.mem_get_info()
            pooled_b = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
            grow = max(grow, min(int(total_b * 0.80) - torch.cuda.memory_allocated(), free_b + pooled_b - (8 << 30)))
        if grow > 0:
            pool = torch.empty(grow, dtype=torch.uint8, device=device)
            del pool
        del wq
    barrier()

    # ---- timed: K batches per GPU
    make_workdir(K * world - 1, shape, local_rank)
    data = DeviceBatches(K * world, B, HW, rank, world, device, on_host=args.host_inputs)
    q = Quantity(model)
    q.profile_phases = True
    with KernelTimer(_native, "hist2048_seg") as kt_hist, KernelTimer(_native, "absmax_seg") as kt_max:
        kt_hist.enabled = kt_max.enabled = True
        barrier()
        t0 = time.perf_counter()
        q.activation_quantize(data)
        barrier()
        elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    images = K * world * B
    value = images / elapsed
    hist_s, max_s = kt_hist.summary(), kt_max.summary()
    timings = dict(q.timings)
    timings["max_reserved_gb"] = round(torch.cuda.max_memory_reserved() / 2 ** 30, 1)
    feat_table = open("./workdir/feat.table").read() if rank == 0 else ""

    result = {
        "metric": BASELINE_METRIC,
        "metric_note": "value = calibration images/s (both passes + KL sweep + feat.table, end to end); "
                       "int8-sim images/s is reported beside it as int8_sim_images_per_s (resident integer activations, logits bit-identical "
                       "to int8_sim_fp32_boundary_images_per_s, the reference's module-boundary form)",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (host-resident, PCIe inclusive)" if args.host_inputs else ""),
        "config": {"workload": "fabu ResNet-%s per-tensor KL calibration, %d synthetic 3x%dx%d images per GPU "
                               "(batch %d x %d steps), %d histogram rows x 2048 bins" %
                               ("50" if args.model == "r50" else "101", K * B, HW, HW, B, K, len(q.net_info) + 1),
                   "batch": B, "images_total": images, "parallelism": "dp%d" % world,
                   "activation_cache": "pass-1 activations kept in a warm allocator pool (80 % of HBM, grown during warm-up); "
                                       "bytes used: phases_s.cache_bytes",
                   "int8_sim_images_per_forward": args.int8_batch},
        "phases_s": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in timings.items()},
    }
    traffic = None
    try:        # HBM bytes per launch from the PMC counters of the committed profile of this same configuration
        with open(os.path.join(ROOT, "profiles", "traffic_hist2048.json")) as fh:
            tj = json.load(fh)
        if (tj["batch"], tj["model"], tj["image"]) == (B, args.model, HW):
            traffic = tj["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    if hist_s:
        ach = hist_s["bytes_per_launch"] / (hist_s["mean_ms"] * 1e-3) / 1e9
        result["roofline"] = {"bound": "hbm", "kernel": "hist2048_seg_kernel", "achieved": round(ach, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                              "traffic": traffic, "launches": hist_s["launches"],
                              "mean_launch_ms": round(hist_s["mean_ms"], 4),
                              "algorithmic_bytes_per_launch": hist_s["bytes_per_launch"]}
        if max_s:
            ach2 = max_s["bytes_per_launch"] / (max_s["mean_ms"] * 1e-3) / 1e9
            result["roofline_absmax"] = {"kernel": "absmax_seg_kernel", "achieved": round(ach2, 1), "unit": "GB/s",
                                         "frac": round(ach2 / HBM_PEAK_GBS, 4), "mean_launch_ms": round(max_s["mean_ms"], 4),
                                         "algorithmic_bytes_per_launch": max_s["bytes_per_launch"],
                                         "note": "pass 1 takes the abs-max of conv and Eltwise outputs inside their own bias add / "
                                                 "residual add (phases_s.fused_*): only what is left (the input image) goes "
                                                 "through this kernel, a launch of tens of microseconds; its streaming rate on "
                                                 "the full 8.6 GB tensor set is 6.2-6.8 TB/s (DESIGN.md section 5)"}

    # ---- the fused fake-quant kernel on its own (north_star: >= 60 % of the HBM roofline)
    try:
        xq = torch.empty(802816 * B, device=device)                      # the largest ResNet-50 activation
        xq.normal_()
        yq = torch.empty_like(xq)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        _native.quandequan(xq, 4, 8, out=yq)
        for a, b in evs:
            a.record()
            _native.quandequan(xq, 4, 8, out=yq)
            b.record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        ach = xq.numel() * 8 / (ms * 1e-3) / 1e9
        result["roofline_fakequant"] = {"bound": "hbm", "kernel": "unary_vec_kernel<QuanDequanOp>", "achieved": round(ach, 1),
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                        "mean_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": xq.numel() * 8.0}
        del xq, yq
    except Exception as e:
        result["roofline_fakequant"] = {"error": repr(e)}

    # ---- int8-sim / fake-quant forward throughput (BASELINE config[2]); replicas only, no collective
    if not args.no_recon:
        try:
            q.weight_quantize()
            barrier()
            batches = [b.to(device) for b in data.owned()[:min(K, 8)]]
            if args.int8_batch > B:                        # larger forwards: concatenated calibration batches
                per = (args.int8_batch + B - 1) // B
                src = batches if len(batches) >= per else batches * per
                batches = [torch.cat(src[i:i + per]) for i in range(0, len(src) - per + 1, per)][:4]
            FB = int(batches[0].shape[0])

            def fwd_rate(net, passes=1):
                """images/s of net over the resident batches; the fast models take several passes so that one
                host hiccup (a GC pause is longer than a whole int8 forward) does not decide the number."""
                with torch.no_grad():
                    net(batches[0])
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(passes):
                        for xb in batches:
                            net(xb)
                    barrier()
                dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
                if distributed:
                    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
                return passes * len(batches) * FB * world / float(dt.item())

            result["float_forward_images_per_s"] = round(fwd_rate(model), 1)
            rec = Reconstruction(build_model(args.model, HW, device))
            info = rec.get_quantity_information()
            result["fakequant_images_per_s"] = round(fwd_rate(rec.ReconTest(info, "./workdir/recontest.pth")), 1)
            rec2 = Reconstruction(build_model(args.model, HW, device))
            info2 = rec2.get_quantity_information()
            int8_net = rec2.ReconModel(info2, "./workdir/recon.pth")
            result["int8_sim_fp32_boundary_images_per_s"] = round(fwd_rate(int8_net, 3), 1)
            # same model, same logits, activations kept as int8/int16 NHWC between the integer layers
            from common.quantity import resident
            with torch.no_grad():
                logits_fp32_boundary = int8_net(batches[0])
            plan = resident.enable(int8_net, batches[0])
            with torch.no_grad():
                same = bool(torch.equal(int8_net(batches[0]), logits_fp32_boundary))
            result["int8_sim_images_per_s"] = round(fwd_rate(int8_net, 8), 1)
            result["int8_sim_resident"] = {"bit_identical_logits": same, "plan": plan, "images_per_forward": FB}
            if not same:                                   # never report a rate for a model that computes something else
                result["int8_sim_images_per_s"] = result["int8_sim_fp32_boundary_images_per_s"]
            else:
                try:
                    result["roofline_int8_conv"] = int8_conv_roofline(model, int8_net, batches[0])
                except Exception as e:
                    result["roofline_int8_conv"] = {"error": repr(e)}
                try:                                       # the same forward replayed as one HIP graph (input copy included)
                    graphed = resident.capture(int8_net, batches[0])
                    with torch.no_grad():
                        same_g = bool(torch.equal(graphed(batches[0]), logits_fp32_boundary))
                    if same_g:
                        result["int8_sim_hipgraph_images_per_s"] = round(fwd_rate(graphed, 8), 1)
                    del graphed
                except Exception as e:
                    result["int8_sim_hipgraph_error"] = repr(e)
        except Exception as e:  # the headline number above stands on its own
            result["recon_error"] = repr(e)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(lambda: build_model(args.model, HW, torch.device("cpu")),
                                                  4, HW, images, q, log)
        except Exception as e:
            result["cpu_baseline"] = {"error": repr(e)}

    # ---- per-channel rows (extension; BASELINE configs[1] words the workload "per-channel"): same two passes with
    # one histogram row per (tensor, channel), read in place by fq_absmax_chan / fq_hist2048_chan, on a bounded
    # sample -- the KL sweep of all rows is a fixed cost of the same order as the passes themselves
    if world == 1 and not args.no_per_channel:
        try:
            pc_batches = data.owned()[:min(K, 8)]
            pc_data = [(b, 0) for b in pc_batches]
            make_workdir(len(pc_data) - 1, shape, local_rank)
            pq = Quantity(model)
            barrier()
            t0 = time.perf_counter()
            pq.activation_quantize_per_channel(pc_data)
            barrier()
            dt = time.perf_counter() - t0
            result["per_channel_calibration"] = {"images": len(pc_data) * B, "rows": int(pq._channel_collector.rows),
                                                 "seconds": round(dt, 3), "images_per_s": round(len(pc_data) * B / dt, 1)}
            del pq
        except Exception as e:
            result["per_channel_calibration"] = {"error": repr(e)}

    sys.stdout = real_stdout
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    if rank == 0:
        log("feat.table head:", feat_table.split("\n")[:4])
        print(json.dumps(result), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
