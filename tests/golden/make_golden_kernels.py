#!/usr/bin/env python3
"""Capture kernel-level golden vectors (G1, G2, G5, G8 of SURVEY.md section 8c) from the imported
reference.  Run in the build container only:

    python tests/golden/make_golden_kernels.py

Writes tests/golden/g1_hist.npz, g2_kl.npz, g5_ops.npz, g8_merge_bn.npz; `... g14` writes g14_interval_num.npz (the collector
and the KL search at INTERVAL_NUM 512 / 1024 / 4096).  The fixtures are data
(inputs or their seeds+sha256, and the reference's outputs); no reference source is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import _refenv  # noqa: E402


def capture_g1(cq):
    out = {}
    meta = {}
    for name, case in cases.g1_cases().items():
        coll = cq.DistributionCollector([name], interval_num=2048, statistic=1, worker_num=1)
        for b in case["p1"]:
            coll.refresh_max_val({name: b})
        mv = coll.max_vals[name]
        iv = coll.distribution_intervals[name]
        for b in case["p2"]:
            coll.add_to_distributions({name: b})
        hist = coll.distributions[name]
        assert hist.dtype == np.int32
        out[name + "/hist"] = hist
        out[name + "/max"] = np.float64(mv)
        out[name + "/interval"] = np.float64(iv)
        meta[name] = dict(
            max_type=type(mv).__name__, interval_type=type(iv).__name__,
            p1_sha=[cases.sha(b) for b in case["p1"]], p2_sha=[cases.sha(b) for b in case["p2"]],
            p1_n=[int(b.size) for b in case["p1"]], p2_n=[int(b.size) for b in case["p2"]])
        print("G1", name, "max", mv, type(mv).__name__, "iv", iv, type(iv).__name__, "sum", hist.sum())
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "g1_hist.npz"), **out)


def capture_g2(cq):
    hs = cases.g2_cases()
    ivs = cases.g2_intervals()
    out = {}
    meta = {}
    for name, h in hs.items():
        q = cq.Quantizer([name], worker_num=1)
        curve = []
        orig = q.compute_kl_divergence

        def rec(a, b, _orig=orig, _curve=curve):
            v = _orig(a, b)
            _curve.append(float(v))
            return v

        q.compute_kl_divergence = rec
        p = q.normalize_distribution(h)
        thr = q.threshold_distribution(p)
        # bits / threshold_value exactly as quantize_worker computes them
        _, bits, tv = q.quantize_worker([name], {name: h}, {name: ivs[name]})
        out[name + "/hist"] = h
        out[name + "/p"] = np.asarray(p)
        out[name + "/kl"] = np.array(curve[:1920], dtype=np.float64)
        out[name + "/thr"] = np.int32(thr)
        out[name + "/bits"] = np.int32(bits[0])
        out[name + "/thr_val"] = np.float64(tv[0])
        out[name + "/interval"] = np.float32(ivs[name])
        meta[name] = dict(p_dtype=str(np.asarray(p).dtype), tv_type=type(tv[0]).__name__,
                          hist_dtype=str(h.dtype))
        print("G2", name, "thr", thr, "bits", bits[0], "tv", tv[0], type(tv[0]).__name__,
              "p", np.asarray(p).dtype, "ncurve", len(curve))
    # the all-zero-tensor interval is a Python float 1e-12 in the reference
    q = cq.Quantizer(["z"], worker_num=1)
    _, bits, tv = q.quantize_worker(["z"], {"z": hs["empty"]}, {"z": 1e-12})
    out["empty_pyfloat/bits"] = np.int32(bits[0])
    out["empty_pyfloat/thr_val"] = np.float64(tv[0])
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "g2_kl.npz"), **out)


def capture_g14(cq):
    """G14 (g14_interval_num.npz): the collector and the KL search of the reference at INTERVAL_NUM 512 / 1024 / 4096."""
    out, meta = {}, {}
    for bins in cases.G14_BINS:
        for name, case in cases.g14_tensor_cases().items():
            coll = cq.DistributionCollector([name], interval_num=bins, statistic=1, worker_num=1)
            for b in case["p1"]:
                coll.refresh_max_val({name: b})
            mv = coll.max_vals[name]
            iv = coll.distribution_intervals[name]
            for b in case["p2"]:
                coll.add_to_distributions({name: b})
            hist = coll.distributions[name]
            assert hist.shape == (bins,)
            out["%d/t/%s/hist" % (bins, name)] = hist
            out["%d/t/%s/max" % (bins, name)] = np.float64(mv)
            out["%d/t/%s/interval" % (bins, name)] = np.float64(iv)
            meta["%d/%s" % (bins, name)] = dict(interval_type=type(iv).__name__)
        for name, h in cases.g14_hists(bins).items():
            q = cq.Quantizer([name], worker_num=1)
            curve = []
            orig = q.compute_kl_divergence

            def rec(a, b, _orig=orig, _curve=curve):
                v = _orig(a, b)
                _curve.append(float(v))
                return v
            q.compute_kl_divergence = rec
            p = q.normalize_distribution(h)
            thr = q.threshold_distribution(p)
            n_curve = len(curve)
            iv = np.float32(3.0 / bins)
            _, bits, tv = q.quantize_worker([name], {name: h}, {name: iv})
            assert n_curve == bins - 128
            out["%d/k/%s/hist" % (bins, name)] = h
            out["%d/k/%s/p" % (bins, name)] = np.asarray(p)
            out["%d/k/%s/kl" % (bins, name)] = np.array(curve[:n_curve], dtype=np.float64)
            out["%d/k/%s/thr" % (bins, name)] = np.int32(thr)
            out["%d/k/%s/bits" % (bins, name)] = np.int32(bits[0])
            out["%d/k/%s/thr_val" % (bins, name)] = np.float64(tv[0])
            out["%d/k/%s/interval" % (bins, name)] = iv
            print("G14", bins, name, "thr", thr, "bits", bits[0], flush=True)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "g14_interval_num.npz"), **out)


def capture_g5(cq):
    import torch
    x = torch.from_numpy(cases.g5_inputs())
    out = {"x": x.numpy()}
    for bit in (-3, -1, 0, 1, 3, 5, 7, 11):
        out["quantity/%d" % bit] = cq.Quantity(bit)(x).numpy()
        out["dequantity/%d" % bit] = cq.DeQuantity(bit)(x).numpy()
        for bw in (8, 16):
            out["quandequan/%d/%d" % (bw, bit)] = cq.QuanDequan(bw, bit)(x).numpy()
    for rs in (-2, -1, 0, 1, 2, 5, 9, 12):
        for bw in (8, 16):
            out["rightshift/%d/%d" % (bw, rs)] = cq.RightShift(bw, rs)(x).numpy()
    for bw in (8, 16):
        out["sp/%d" % bw] = cq.Sp(bw)(x).numpy()
    y = torch.flip(x, dims=[0])
    out["newadd"] = cq.NewAdd()(x, y).numpy()
    out["biasadd"] = cq.BiasAdd()(x, y).numpy()
    np.savez_compressed(os.path.join(HERE, "g5_ops.npz"), **out)
    print("G5 ops captured", len(out))


def capture_g8(cq):
    import torch
    import torch.nn as nn
    torch.manual_seed(0)
    out = {}
    for tag, bias in (("nobias", False), ("bias", True)):
        seq = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=bias), nn.BatchNorm2d(8), nn.ReLU(False))
        bn = seq[1]
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.1)
        seq.eval()
        pre = {k: v.clone().numpy() for k, v in seq.state_dict().items()}
        x = torch.randn(2, 3, 8, 8)
        y0 = seq(x).detach().numpy()
        merged = cq.merge_bn(seq)
        y1 = merged(x).detach().numpy()
        for k, v in pre.items():
            out["%s/pre/%s" % (tag, k)] = v
        out["%s/w" % tag] = merged[0].weight.detach().numpy()
        out["%s/b" % tag] = merged[0].bias.detach().numpy()
        out["%s/x" % tag] = x.numpy()
        out["%s/y_bn" % tag] = y0
        out["%s/y_merged" % tag] = y1
        out["%s/bn_type_after" % tag] = np.array(type(merged[1]).__name__)
    np.savez_compressed(os.path.join(HERE, "g8_merge_bn.npz"), **out)
    print("G8 merge_bn captured")


def main():
    cq, _ = _refenv.import_reference()
    which = sys.argv[1:] or ["g1", "g2", "g5", "g8"]
    if "g1" in which:
        capture_g1(cq)
    if "g5" in which:
        capture_g5(cq)
    if "g8" in which:
        capture_g8(cq)
    if "g2" in which:
        capture_g2(cq)
    if "g14" in which:
        capture_g14(cq)


if __name__ == "__main__":
    main()
