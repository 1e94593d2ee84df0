#!/usr/bin/env python3
"""GPU fuzz of the fused calibration forward on RANDOM model topologies, three calibrations of the same weights and batches:
  A  every switch on (own float kernels, statistics in the producers' epilogues, conv + Eltwise + ReLU as one kernel, outputs nobody
     reads not written, own pools, activation cache);
  B  the same without the two fusions that rest on a PROOF about the model's dataflow (fuse_conv_add, skip_unread_outputs): the same
     kernels compute the same sums in the same order, so maxima, histograms and table must equal A's BIT FOR BIT -- a chain wrongly
     taken hands somebody a tensor nobody wrote, and that shows here;
  D  the same as A without the producers' histograms in pass 2 (fuse_hist): the plain own kernels + one statistic launch, = A bit for bit;
  C  every switch off (library convolutions, one streaming statistic launch per tensor): another summation order, so the maxima
     agree to that bound and a histogram differs by the few elements the bound moves across a bin edge; a table line may then differ
     where the KL search has a near tie (seen: 1 model in 300, one line, one bit) -- reported as a note, not as a finding.
What it is for: the dataflow proofs (deferral, relu-only, keepers) on graphs nobody wrote a test for -- two consumers of one tensor,
a shortcut that is itself a convolution output, concatenations, in-place ReLUs, pools in odd places.
usage: model_fuzz.py [models=40] [seed=1]"""
import os, sys, random
import numpy as np
import torch
import torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from tools import Quantity
from workdir_util import product_workdir

SWITCHES = ("fuse_bias_absmax", "fuse_relu", "fuse_hist", "own_pools", "fuse_conv_add", "skip_unread_outputs", "own_conv1x1")


from cases import RandomNet as Net, random_net      # (tests/golden/cases.py: the generator is shared with golden G11)


def fold(drawn):
    """random_net()'s tuple with the model's BatchNorm2d layers folded into their convolutions (the flow's first step), if it has any."""
    from common.quantity import merge_bn
    model = drawn[0]
    if getattr(model, "has_bn", False):
        out = sys.stdout; sys.stdout = open(os.devnull, "w")
        try:
            model = merge_bn(model)
        finally:
            sys.stdout = out
    return (model,) + tuple(drawn[1:])


def calibrate(model, size, batches, off=(), cache_gb=None, plan=None):
    """One calibration.  cache_gb: what pass 1 may keep for pass 2 (FQ_ACT_CACHE_GB; None: the engine's own rule -- nothing in a
    process without a warm pool); plan: "A" / "B" forces the cache plan (FQ_CACHE_PLAN)."""
    saved = {k: os.environ.get(k) for k in ("FQ_ACT_CACHE_GB", "FQ_CACHE_PLAN")}
    for k, v in (("FQ_ACT_CACHE_GB", cache_gb), ("FQ_CACHE_PLAN", plan)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        return _calibrate(model, size, batches, off)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _calibrate(model, size, batches, off=()):
    with product_workdir(input_shape="1,%d,%d,%d" % (getattr(model, "cin", 3), size, size), device="gpu", max_cali_img_num=len(batches) - 1) as tmp:
        q = Quantity(model)
        for s in off:
            setattr(q, s, False)
        out = sys.stdout; sys.stdout = open(os.devnull, "w")
        try:
            bits = q.activation_quantize(batches)
        finally:
            sys.stdout = out
        c = q._collector
        return (dict(bits), {k: float(v) for k, v in c.max_vals.items()}, c.hist_device.clone(),
                open(os.path.join(tmp, "test", "workdir", "feat.table")).read(), dict(q.timings), list(c._tensor_list))


def calibrate_channels(model, size, batches):
    """The per-(tensor, channel) extension on the same model: (maxima [rows], histograms [rows, 2048], row ranges, table text)."""
    with product_workdir(input_shape="1,%d,%d,%d" % (getattr(model, "cin", 3), size, size), device="gpu", max_cali_img_num=len(batches) - 1) as tmp:
        q = Quantity(model)
        out = sys.stdout; sys.stdout = open(os.devnull, "w")
        try:
            q.activation_quantize_per_channel(batches)
        finally:
            sys.stdout = out
        c = q._channel_collector
        mx, hist = c._stat_tensors()
        names = ["image"] + list(q.net_info.keys())
        return mx.clone(), hist.clone(), {n: c.row_range(n) for n in names}, open(os.path.join(tmp, "test", "workdir", "feat_channel.table")).read()


def run_channels(n, seed, log=print, odd=False, share=False, bn=False):
    """The per-channel calibration of n random models: twice (equal bit for bit), and against the per-tensor calibration of the same
    model -- a tensor's maximum is the largest of its channels' maxima, its histogram holds as many elements as theirs together."""
    torch.backends.cudnn.deterministic = bool(odd)
    bad = 0
    for i in range(n):
        model, size, bs, rng = fold(random_net(i, seed, odd, "cuda", share, bn))
        batches = [(torch.randn(bs, model.cin, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(3)]
        try:
            t = calibrate(model, size, batches)
            c1 = calibrate_channels(model, size, batches)
            c2 = calibrate_channels(model, size, batches)
        except Exception as e:
            bad += 1
            log("model %d (seed %d): %s: %s" % (i, seed, type(e).__name__, str(e)[:300]))
            continue
        problems = []
        if not (torch.equal(c1[0], c2[0]) and torch.equal(c1[1], c2[1]) and c1[3] == c2[3]):
            problems.append("two per-channel calibrations differ")
        for r, name in enumerate(t[5]):
            lo, hi = c1[2][name]
            if float(c1[0][lo:hi].max()) != t[1][name]:
                problems.append("%s: largest channel maximum %.8g, tensor maximum %.8g" % (name, float(c1[0][lo:hi].max()), t[1][name]))
            if int(c1[1][lo:hi].sum()) != int(t[2][r].sum()):
                problems.append("%s: %d elements in the channel histograms, %d in the tensor's" % (name, int(c1[1][lo:hi].sum()), int(t[2][r].sum())))
        if problems:
            bad += 1
            log("model %d (seed %d, %d modules): %s" % (i, seed, model.n, "; ".join(problems[:4])))
    return bad


def run_cache(n, seed, log=print, odd=False, share=False, bn=False):
    """The activation cache on n random models: nothing kept / a few MB (the deepest tensors of every batch: plan B, pass 2 re-runs
    a prefix of the network and stops) / more (whole batches: plan A) / everything, each plan also forced -- all must give the
    statistics of the calibration without a cache bit for bit (the kept tensors ARE the ones pass 1 took the maxima of)."""
    torch.backends.cudnn.deterministic = bool(odd)
    bad, plans = 0, {}
    for i in range(n):
        model, size, bs, rng = fold(random_net(i, seed, odd, "cuda", share, bn))
        batches = [(torch.randn(bs, model.cin, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(4)]
        try:
            base = calibrate(model, size, batches, cache_gb=0)
            problems = []
            for gb, plan in ((0.002, None), (0.002, "B"), (0.01, None), (0.01, "A"), (0.03, "B"), (4, None)):
                c = calibrate(model, size, batches, cache_gb=gb, plan=plan)
                kind = (c[4].get("cache_plan") or {}).get("kind")
                plans[(kind, c[4].get("cache_bytes", 0) > 0)] = plans.get((kind, c[4].get("cache_bytes", 0) > 0), 0) + 1
                # (sums pass 1 left to pass 2's pair / chain kernels -- Quantity.pair_hist, pair_chain: only a cache makes them)
                plans["sums_left_to_pairs"] = plans.get("sums_left_to_pairs", 0) + int(c[4].get("sums_left_to_pass2_pairs", 0) or 0)
                if c[3] != base[3] or c[1] != base[1] or not torch.equal(c[2], base[2]):
                    rows = [k for r, k in enumerate(base[5]) if not torch.equal(c[2][r], base[2][r]) or c[1][k] != base[1][k]]
                    problems.append("cache %s GB plan %s (%s, %d bytes kept): rows %s differ" % (gb, plan, c[4].get("cache_plan"), c[4].get("cache_bytes", 0), rows[:5]))
        except Exception as e:
            bad += 1
            log("model %d (seed %d): %s: %s" % (i, seed, type(e).__name__, str(e)[:300]))
            continue
        if problems:
            bad += 1
            log("model %d (seed %d, %d modules): %s" % (i, seed, model.n, "; ".join(problems[:3])))
    return bad, plans


def run(n, seed, log=print, odd=False, share=False, bn=False):
    """n random models; returns (models with a finding, what the fused forwards launched in all).  odd: with depthwise / dilated
    convolutions and nearest-neighbour upsampling here and there."""
    # (layers the own kernels do not take run on the convolution library, whose default kernels do not give the same bits from call
    #  to call; its deterministic mode makes A = B a meaningful test for such models too)
    torch.backends.cudnn.deterministic = bool(odd)
    bad, seen = 0, {"conv_add_launches": 0, "conv_add_hist_launches": 0, "conv_add_chains_proven": 0, "relu_only_chains_proven": 0,
                    "launches_without_own_output": 0, "own_conv1x1_launches": 0, "fused_hist_launches": 0, "refused": 0}
    for i in range(n):
        model, size, bs, rng = fold(random_net(i, seed, odd, "cuda", share, bn))
        batches = [(torch.randn(bs, model.cin, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(3)]   # (data, label)
        if odd and rng.random() < 0.5:                           # a ragged last batch
            batches[-1] = (batches[-1][0][:3].contiguous(), batches[-1][1][:3])
        try:
            a = calibrate(model, size, batches)
            b = calibrate(model, size, batches, off=("fuse_conv_add", "skip_unread_outputs"))
            c = calibrate(model, size, batches, off=SWITCHES)
            d = calibrate(model, size, batches, off=("fuse_hist",))     # pass 2 on the plain own kernels, statistics from the hooks
        except Exception as e:                                   # a crash is a finding too
            bad += 1
            log("model %d (seed %d): %s: %s" % (i, seed, type(e).__name__, str(e)[:300]))
            continue
        for k in seen:
            v = a[4].get(k if k != "refused" else "chains_refused_for_keepers", 0)
            seen[k] += len(v) if isinstance(v, dict) else int(v or 0)
        problems = []
        # A against B: bit for bit
        if a[3] != b[3] or a[1] != b[1] or not torch.equal(a[2], b[2]):
            rows = [k for r, k in enumerate(a[5]) if not torch.equal(a[2][r], b[2][r]) or a[1][k] != b[1][k]]
            problems.append("with / without the proven chains: rows %s differ" % rows[:6])
        if a[3] != d[3] or a[1] != d[1] or not torch.equal(a[2], d[2]):
            rows = [k for r, k in enumerate(a[5]) if not torch.equal(a[2][r], d[2][r]) or a[1][k] != d[1][k]]
            problems.append("with / without the producers' histograms: rows %s differ" % rows[:6])
        # A against C: to the summation-order bound
        for k, v in a[1].items():
            if abs(v - c[1][k]) > 2e-5 * max(abs(v), abs(c[1][k]), 1e-6):
                problems.append("max of %s: %.8g vs %.8g" % (k, v, c[1][k]))
        ha, hc = a[2].double(), c[2].double()
        if float((ha.sum(1) - hc.sum(1)).abs().max()) > 2:       # (an element at exactly zero after one of the two roundings)
            problems.append("histogram totals differ by up to %d" % int((ha.sum(1) - hc.sum(1)).abs().max()))
        moved = (ha - hc).abs().sum(1)
        # (elements next to a bin edge move to the NEIGHBOURING bin when the bin width changes in its last bit -- a handful, or a
        #  whole cluster of equal values at once: a depthwise layer over a rectified input writes its bias wherever the window is all
        #  zeros, 6 180 equal values in one model.  What a wrong tensor would do is move mass FAR: the distance between the two
        #  cumulative histograms, in bins per element, stays below 0.1 for neighbours and is tens of bins for garbage.)
        shift = (ha.cumsum(1) - hc.cumsum(1)).abs().sum(1) / ha.sum(1).clamp(min=1)
        if bool((shift > 0.1).any()):
            r = int(shift.argmax())
            problems.append("histogram of %s differs from the library path: %d elements moved, %.2f bins per element" % (a[5][r], int(moved[r]), float(shift[r])))
        if a[3] != c[3] and not problems:
            lines = [k for k in a[0] if a[0][k] != c[0].get(k)]
            # (the histograms passed the test above: what is left is the KL search deciding between two nearly equal candidates)
            if len(lines) > 3 or any(abs(a[0][k] - c[0][k]) > 1 for k in lines):
                problems.append("tables differ from the library path: %s" % lines[:6])
            else:
                seen["near_ties"] = seen.get("near_ties", 0) + len(lines)
                log("  (model %d, seed %d: %s one bit apart from the library path -- a near tie of the KL search; %s elements moved)"
                    % (i, seed, ", ".join(lines), ", ".join("%d of %d" % (int(moved[a[5].index(k)]), int(ha.sum(1)[a[5].index(k)])) for k in lines)))
        if problems:
            bad += 1
            log("model %d (seed %d, %d modules): %s" % (i, seed, model.n, "; ".join(problems[:4])))
    return bad, seen


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    odd, share = "odd" in sys.argv[3:], "share" in sys.argv[3:]         # share: nn.ReLU modules that serve several places of the graph
    bn = "bn" in sys.argv[3:]                                            # bn: BatchNorm2d behind half of the convolutions, folded by merge_bn first
    if "cache" in sys.argv[3:]:
        bad, plans = run_cache(n, seed, odd=odd, share=share, bn=bn)
        print("model_fuzz cache%s: %d random models (seed %d), %d with a finding; (plan, something kept) -> calibrations: %s" % (" odd" if odd else "", n, seed, bad, plans))
        return
    if "channels" in sys.argv[3:]:
        print("model_fuzz channels%s: %d random models (seed %d), %d with a finding" % (" odd" if odd else "", n, seed, run_channels(n, seed, odd=odd, share=share, bn=bn)))
        return
    bad, seen = run(n, seed, odd=odd, share=share, bn=bn)
    print("model_fuzz%s: %d random models (seed %d), %d with a finding; fused launches seen: %s" % ((" odd" if odd else "") + (" share" if share else "") + (" bn" if bn else ""), n, seed, bad, seen))


if __name__ == "__main__":
    main()
