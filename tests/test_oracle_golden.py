"""Pin the CPU oracle (oracle/fq_oracle.c) against golden vectors captured from the imported
reference (tests/golden/make_golden_kernels.py).  CPU only."""
import json
import os

import numpy as np
import pytest

import cases


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


# ---------------------------------------------------------------- G1: absmax / interval / histogram
@pytest.mark.parametrize("name", list(cases.g1_cases().keys()))
def test_g1_absmax_interval_hist(oracle, golden_dir, name):
    g = _load(golden_dir, "g1_hist.npz")
    meta = json.loads(str(g["meta"]))[name]
    case = cases.g1_cases()[name]
    assert [cases.sha(b) for b in case["p1"]] == meta["p1_sha"], "input generator drifted"
    assert [cases.sha(b) for b in case["p2"]] == meta["p2_sha"], "input generator drifted"
    m = np.float32(0)
    for b in case["p1"]:
        m = oracle.absmax(b, m)
    assert np.float32(g[name + "/max"]) == m
    iv = oracle.interval(m)
    # the reference yields a Python float 1e-12 for an all-zero tensor; its fp32 image is ours
    assert np.float32(g[name + "/interval"]) == iv
    hist = np.zeros(2048, dtype=np.int64)
    for b in case["p2"]:
        oracle.hist2048(b, iv, hist)
    ref = g[name + "/hist"]
    assert ref.dtype == np.int32
    np.testing.assert_array_equal(hist, ref.astype(np.int64))


# ---------------------------------------------------------------- numpy pairwise-sum order
def test_np_pairwise_sum_bit_exact(oracle):
    rng = np.random.default_rng(42)
    for n in list(range(0, 40)) + [63, 64, 65, 127, 128, 129, 130, 135, 136, 255, 256, 257, 300, 511,
                                   777, 1000, 1023, 1024, 1025, 1919, 1920, 2047, 2048, 4097]:
        a = rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n))
        assert oracle.np_sum(a) == float(np.sum(a)), n
        b = np.abs(a) * 1e-3
        assert oracle.np_sum(b) == float(b.sum()), n


# ---------------------------------------------------------------- G2: KL threshold sweep
G2_NAMES = list(cases.g2_cases().keys())


def _assert_curves_close(a, b, rel=1e-12):
    """KL(t) curves: NaNs (the reference's incremental tail can go slightly negative, log of a
    negative is NaN, and NaN never wins the argmin) must sit at the same t; finite values agree to
    a few ulp (np.log in the capture container is SVML, not correctly rounded)."""
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    f = ~np.isnan(a)
    scale = np.maximum(np.abs(b[f]), 1e-300)
    assert np.max(np.abs(a[f] - b[f]) / scale, initial=0.0) < rel


@pytest.mark.parametrize("name", G2_NAMES)
def test_g2_normalize_and_threshold(oracle, golden_dir, name):
    g = _load(golden_dir, "g2_kl.npz")
    h = cases.g2_cases()[name]
    np.testing.assert_array_equal(h, g[name + "/hist"])
    p = oracle.normalize(h)
    np.testing.assert_array_equal(p, g[name + "/p"])          # float64, bit exact
    thr, curve = oracle.kl_threshold(p, want_curve=True)
    assert thr == int(g[name + "/thr"])
    ref_curve = g[name + "/kl"]
    # np.log in the capture container is SVML (not correctly rounded): allow a few ulp on KL(t)
    _assert_curves_close(curve, ref_curve)
    iv = np.float32(g[name + "/interval"])
    bits, tv = oracle.bits_from_threshold(thr, iv)
    assert bits == int(g[name + "/bits"])
    assert tv == np.float32(g[name + "/thr_val"])


@pytest.mark.parametrize("name", G2_NAMES)
def test_g2_fq_log_variant_agrees(oracle, golden_dir, name):
    """The oracle run with include/fq_log.h (the log the HIP kernel uses) picks the same threshold."""
    g = _load(golden_dir, "g2_kl.npz")
    p = oracle.normalize(cases.g2_cases()[name])
    thr, curve = oracle.kl_threshold(p, want_curve=True, use_fq_log=True)
    assert thr == int(g[name + "/thr"])
    thr2, curve2 = oracle.kl_threshold(p, want_curve=True, use_fq_log=False)
    _assert_curves_close(curve, curve2)


def test_g2_empty_pyfloat_interval(oracle, golden_dir):
    g = _load(golden_dir, "g2_kl.npz")
    bits, _ = oracle.bits_from_threshold(128, np.float32(1e-12))
    assert bits == int(g["empty_pyfloat/bits"])


# ---------------------------------------------------------------- G5: element-wise ops
def test_g5_ops(oracle, golden_dir):
    g = _load(golden_dir, "g5_ops.npz")
    x = g["x"]
    np.testing.assert_array_equal(x, cases.g5_inputs())
    for key in g.files:
        parts = key.split("/")
        if parts[0] == "quantity":
            got = oracle.quantity(x, int(parts[1]))
        elif parts[0] == "dequantity":
            got = oracle.dequantity(x, int(parts[1]))
        elif parts[0] == "quandequan":
            got = oracle.quandequan(x, int(parts[2]), int(parts[1]))
        elif parts[0] == "rightshift":
            got = oracle.rightshift(x, int(parts[2]), int(parts[1]))
        elif parts[0] == "sp":
            got = oracle.sp(x, int(parts[1]))
        elif parts[0] == "newadd":
            got = oracle.add_sat(x, x[::-1].copy())
        else:
            continue
        np.testing.assert_array_equal(got.view(np.uint32), g[key].view(np.uint32), err_msg=key)
