"""INTERVAL_NUM other than 2048 (reference tools/configs.yml:24; distribution_collector.py:9-14 takes it as `interval_num`, and
quantizer.py:98-167 sweeps t = 128 .. distribution.size - 1 whatever the size).  Golden G14 (tests/golden/make_golden_kernels.py
g14): the reference's collector on seeded tensors and its KL search on seeded histograms at 512 / 1024 / 4096 bins.
CPU: the oracle's bins-generic entry points against G14, exact (KL curves to 1e-12 relative -- NumPy's SVML log -- same argmin).
GPU: fq_hist_seg_n / fq_kl_threshold_n and the drop-in's DistributionCollector / Quantizer against G14 and the oracle; the KL
curves bit for bit against the oracle's fq_log form."""
import os

import numpy as np
import pytest

import cases


@pytest.fixture(scope="module")
def g14(golden_dir):
    return np.load(os.path.join(golden_dir, "g14_interval_num.npz"))


@pytest.mark.parametrize("bins", cases.G14_BINS)
def test_oracle_histograms_at_other_interval_nums_equal_the_reference(oracle, g14, bins):
    for name, case in cases.g14_tensor_cases().items():
        m = np.float32(0)
        for b in case["p1"]:
            m = oracle.absmax(b, m)
        iv = oracle.interval(m, 1, bins) if m != 0 else np.float32(1e-12)
        assert np.float64(m) == g14["%d/t/%s/max" % (bins, name)], name
        assert np.float32(g14["%d/t/%s/interval" % (bins, name)]) == iv, name
        hist = np.zeros(bins, dtype=np.int64)
        for b in case["p2"]:
            oracle.hist2048(b, iv, hist)
        np.testing.assert_array_equal(hist, g14["%d/t/%s/hist" % (bins, name)].astype(np.int64), err_msg="%d %s" % (bins, name))


@pytest.mark.parametrize("bins", cases.G14_BINS)
def test_oracle_kl_search_at_other_interval_nums_equals_the_reference(oracle, g14, bins):
    for name, h in cases.g14_hists(bins).items():
        p = oracle.normalize(h)
        np.testing.assert_array_equal(p, g14["%d/k/%s/p" % (bins, name)], err_msg=name)
        thr, curve = oracle.kl_threshold(p, want_curve=True)
        ref = g14["%d/k/%s/kl" % (bins, name)]
        assert curve.shape == ref.shape == (bins - 128,)
        both = np.isfinite(ref) & np.isfinite(curve)
        assert np.array_equal(np.isnan(ref), np.isnan(curve))
        assert np.all(np.abs(curve[both] - ref[both]) <= 1e-12 * np.maximum(1.0, np.abs(ref[both]))), name
        assert thr == int(g14["%d/k/%s/thr" % (bins, name)]), name
        iv = np.float32(g14["%d/k/%s/interval" % (bins, name)])
        bits, tv = oracle.bits_from_threshold(thr, iv)
        assert bits == int(g14["%d/k/%s/bits" % (bins, name)]) and np.float64(tv) == g14["%d/k/%s/thr_val" % (bins, name)]


def test_an_unsupported_interval_num_is_refused():
    from common.quantity import _native
    assert _native.SUPPORTED_BINS == (512, 1024, 2048, 4096)


@pytest.mark.gpu
@pytest.mark.parametrize("bins", cases.G14_BINS)
def test_gpu_collector_and_quantizer_at_other_interval_nums_equal_the_reference(oracle, g14, bins):
    """The drop-in's own classes with interval_num = bins (what tools.Quantity builds from configs.yml's INTERVAL_NUM)."""
    import torch
    from common.quantity import DistributionCollector, Quantizer, _native
    for name, case in cases.g14_tensor_cases().items():
        coll = DistributionCollector([name], interval_num=bins, statistic=1, worker_num=1)
        assert not coll.fused_hist_ok and not coll.supports_pairs
        for b in case["p1"]:
            coll.refresh_max_val({name: torch.from_numpy(b).cuda()})
        assert np.float64(coll.max_vals[name]) == g14["%d/t/%s/max" % (bins, name)], name
        assert np.float64(coll.distribution_intervals[name]) == g14["%d/t/%s/interval" % (bins, name)], name
        for b in case["p2"]:
            coll.add_to_distributions({name: torch.from_numpy(b).cuda()})
        got = coll.hist_device.cpu().numpy()
        assert got.shape == (1, bins)
        np.testing.assert_array_equal(got[0], g14["%d/t/%s/hist" % (bins, name)].astype(np.int64), err_msg="%d %s" % (bins, name))
    hs = cases.g14_hists(bins)
    names = list(hs)
    hist = torch.from_numpy(np.stack([np.asarray(hs[n]).astype(np.int64) for n in names])).cuda()
    thr, curve = _native.kl_threshold(hist, want_curve=True)
    thr, curve = thr.cpu().numpy(), curve.cpu().numpy()
    for i, n in enumerate(names):
        assert int(thr[i]) == int(g14["%d/k/%s/thr" % (bins, n)]), n
        _t, want = oracle.kl_threshold(oracle.normalize(hs[n]), want_curve=True, use_fq_log=True)
        assert np.array_equal(curve[i].view(np.int64), want.view(np.int64)), n           # bit for bit, NaNs included
    q = Quantizer(names, worker_num=1)
    ivs = {n: np.float32(g14["%d/k/%s/interval" % (bins, n)]) for n in names}
    q.quantize({n: hs[n] for n in names}, ivs)
    for n in names:
        assert q.bits[n] == int(g14["%d/k/%s/bits" % (bins, n)]) and q.threshold_bins[n] == int(g14["%d/k/%s/thr" % (bins, n)])
        assert np.float64(q.threshold_value[n]) == g14["%d/k/%s/thr_val" % (bins, n)]


@pytest.mark.gpu
def test_gpu_calibration_with_interval_num_1024_equals_the_oracle_engine_on_the_same_activations(oracle, monkeypatch):
    """End to end through tools.Quantity with INTERVAL_NUM 1024 in configs.yml: the HIP engine (histograms through fq_hist_seg_n,
    pass 2 without the producers' fused 2048-bin epilogues, KL through fq_kl_threshold_n) against the oracle engine fed the
    very same activations -- every histogram and the table."""
    import torch
    from common.quantity import DistributionCollector, merge_bn
    from engine_doubles import OracleQuantizer
    from model.resnet.ResNet_18_fabu import ResNet18
    from tools import Quantity
    from workdir_util import product_workdir
    from common.quantity import Quantizer
    seen = {}

    class SpyQuantizer(Quantizer):                       # what the KL search is handed (merge groups already pooled)
        def quantize(self, distributions, distribution_intervals):
            seen["hist"] = distributions.cpu().numpy().copy()
            seen["iv"] = dict(distribution_intervals)
            return super().quantize(distributions, distribution_intervals)

    class Spy(Quantity):
        quantizer_cls = SpyQuantizer

    with product_workdir(device="gpu", max_cali_img_num=1, interval_num=1024) as tmp:
        model = merge_bn(cases.seed_model(ResNet18()).eval()).cuda()
        q = Spy(model)
        bits = q.activation_quantize(cases.calib_batches(2, (4, 3, 32, 32)))
        assert seen["hist"].shape == (30, 1024) and q.timings["fused_hist_launches"] == 0
        assert type(q._collector).__module__ == "common.quantity.distribution_collector" and q._collector.hist_device.shape == (30, 1024)
        names = ["image"] + list(q.net_info.keys())
        oq = OracleQuantizer(names, worker_num=1)
        oq.quantize({n: seen["hist"][i] for i, n in enumerate(names)}, seen["iv"])
        # (bit tying of merge groups happens after the search: compare the raw thresholds and the bits the search itself found)
        assert {n: int(q._quantizer.threshold_bins[n]) for n in names} == {n: int(oq.threshold_bins[n]) for n in names}
        assert all(int(seen["hist"][i].sum()) > 0 for i in range(30)) and len(bits) == 30
        # ... and every histogram row is the oracle's 1024-bin histogram of the very tensors: a second calibration of the same
        # batches with every tensor materialised, taped and replayed through the oracle
        taped = {}

        class TapeCollector(DistributionCollector):
            def add_to_distributions(self, tensors):
                for k, v in tensors.items():
                    taped.setdefault(k, []).append(v.detach().cpu().numpy().copy())
                super().add_to_distributions(tensors)

        class TapeQuantity(Quantity):
            collector_cls = TapeCollector
            fuse_bias_absmax = False
            fuse_relu = False

        q2 = TapeQuantity(model)
        q2.activation_quantize(cases.calib_batches(2, (4, 3, 32, 32)))
        got = q2._collector.hist_device.cpu().numpy()
        ivs = q2._collector._distribution_intervals
        for i, n in enumerate(names):
            ref = np.zeros(1024, dtype=np.int64)
            for x in taped[n]:
                oracle.hist2048(x, np.float32(ivs[n]), ref)
            np.testing.assert_array_equal(got[i], ref, err_msg=n)
