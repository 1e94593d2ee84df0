"""Per-channel calibration extension: rows = (tensor, channel).  Same kernels, same per-row arithmetic;
checked against the CPU oracle channel by channel on the activations the GPU produced.  pytest -m gpu"""
import os

import numpy as np
import pytest
import torch

import cases
from workdir_util import product_workdir

pytestmark = pytest.mark.gpu


def test_channel_collector_vs_oracle(oracle):
    from common.quantity.channel_collector import ChannelCollector
    rng = np.random.default_rng(5)
    feats = [{"a": torch.from_numpy(rng.standard_normal((3, 5, 7, 6), dtype=np.float32) * np.float32(b + 1)).cuda(),
              "b": torch.from_numpy(np.maximum(rng.standard_normal((3, 4), dtype=np.float32), 0)).cuda()}
             for b in range(2)]
    coll = ChannelCollector({"a": 5, "b": 4})
    for f in feats:
        coll.refresh_max_val(f)
    iv = coll.intervals()
    for f in feats:
        coll.add_to_distributions(f)
    bits = coll.quantize()
    mx = coll.max_device.cpu().numpy()
    hist = coll.hist_device.cpu().numpy()
    for name, C in (("a", 5), ("b", 4)):
        lo, hi = coll.row_range(name)
        for c in range(C):
            ref_m = np.float32(0)
            for f in feats:
                ref_m = oracle.absmax(f[name][:, c].cpu().numpy(), ref_m)
            assert mx[lo + c] == ref_m
            ref_iv = oracle.interval(ref_m)
            assert iv[lo + c] == ref_iv
            ref_h = np.zeros(2048, dtype=np.int64)
            for f in feats:
                oracle.hist2048(f[name][:, c].cpu().numpy(), ref_iv, ref_h)
            np.testing.assert_array_equal(hist[lo + c], ref_h)
            t = oracle.kl_threshold(oracle.normalize(ref_h))
            assert coll.threshold_bins[lo + c] == t
            assert bits[name][c] == oracle.bits_from_threshold(t, ref_iv)[0]


@pytest.mark.parametrize("shape", [(45, 3, 40, 40), (2, 5, 33, 33), (130, 7, 5, 5), (3, 2, 224, 224), (9, 1000), (1, 4, 1, 1),
                                   (128, 64, 14, 14),      # small planes, 8-channel blocks, 16-byte loads
                                   (40, 20, 7, 7),         # 7x7 planes (196 B), ragged last block of 4 channels
                                   (33, 10, 7, 7),         # image stride not a multiple of 16 bytes: 4-byte loads
                                   (200, 16, 28, 28),      # 5 image groups per channel block (<= 32 768 elements per row)
                                   (40, 9, 33, 31),        # HW = 1023: just below the big-plane switch, odd everything
                                   (5, 3, 300, 300),       # big planes, 90 000 elements, one image group of 2 + ...
                                   (1, 2, 512, 512),       # config-5 sized plane: one image per workgroup, no division
                                   (6, 12, 32, 32)])       # HW = 1024: the switch itself
def test_channel_kernels_large_and_ragged_planes(oracle, shape):
    """fq_absmax_chan / fq_hist2048_chan on planes above and below the 1024-element switch (one channel per workgroup
    with 32-bit bins above, blocks of 8 channels with packed 16-bit bins below), planes whose start is not 16-byte
    aligned, several image groups per channel, ragged channel blocks, [N, F] inputs -- row by row against the oracle."""
    from common.quantity import _native as nat
    rng = np.random.default_rng(sum(shape))
    x = (rng.standard_normal(shape, dtype=np.float32) * np.float32(2.5)).astype(np.float32)
    x[x < -3] = 0
    C = shape[1]
    row0 = 3
    dev = torch.from_numpy(x).cuda()
    mx = torch.zeros(row0 + C + 2, device="cuda")
    nat.absmax_chan([dev], [row0], mx)
    mxh = mx.cpu().numpy()
    ref_m = np.array([oracle.absmax(np.ascontiguousarray(x[:, c]).ravel()) for c in range(C)], dtype=np.float32)
    np.testing.assert_array_equal(mxh[row0:row0 + C], ref_m)
    assert not mxh[:row0].any() and not mxh[row0 + C:].any()
    iv = np.full(row0 + C + 2, 1.0, dtype=np.float32)
    iv[row0:row0 + C] = [oracle.interval(m) for m in ref_m]
    hist = torch.zeros(row0 + C + 2, 2048, dtype=torch.int64, device="cuda")
    nat.hist2048_chan([dev], [row0], torch.from_numpy(iv).cuda(), hist)
    nat.hist2048_chan([dev], [row0], torch.from_numpy(iv).cuda(), hist)          # accumulates
    hh = hist.cpu().numpy()
    for c in range(C):
        ref = oracle.hist2048(np.ascontiguousarray(x[:, c]).ravel(), iv[row0 + c])
        np.testing.assert_array_equal(hh[row0 + c], 2 * ref)
    assert not hh[:row0].any() and not hh[row0 + C:].any()


def test_channel_histogram_owned_rows_and_shared_rows(oracle, monkeypatch):
    """A row that one workgroup owns for the whole call (all images in one group, no other segment on the row) is flushed with
    plain 16-byte read-add-write instead of 64-bit atomics.  Same counts either way (FQ_CHAN_OWN_FLUSH=0: atomics everywhere);
    two tensors that feed the SAME rows in one call are seen on the host and keep the atomics (their sum is exact); a histogram
    buffer that is only 8-byte aligned keeps the atomics; rows next to the touched ones stay untouched."""
    from common.quantity import _native as nat
    rng = np.random.default_rng(77)
    shapes = [(64, 24, 14, 14), (64, 20, 7, 7), (16, 5, 28, 28)]       # one-channel form, 8-channel blocks (ragged), several per CU
    xs = [torch.from_numpy((rng.standard_normal(s, dtype=np.float32) * np.float32(1.5))).cuda() for s in shapes]
    row0s = [2, 2 + 24 + 1, 2 + 24 + 1 + 20 + 3]
    rows = row0s[-1] + 5 + 2
    iv = torch.from_numpy(rng.uniform(0.002, 0.004, rows).astype(np.float32)).cuda()

    def ref_rows(x, r0, acc):
        xh = x.cpu().numpy()
        for c in range(xh.shape[1]):
            acc[r0 + c] += oracle.hist2048(np.ascontiguousarray(xh[:, c]).ravel(), np.float32(iv[r0 + c].item()))
    want = np.zeros((rows, 2048), dtype=np.int64)
    for x, r0 in zip(xs, row0s):
        ref_rows(x, r0, want)
    got = {}
    for env in ("1", "0"):
        monkeypatch.setenv("FQ_CHAN_OWN_FLUSH", env)
        hist = torch.full((rows, 2048), 5, dtype=torch.int64, device="cuda")      # earlier batches' counts stay
        nat.hist2048_chan(xs, row0s, iv, hist)
        nat.hist2048_chan(xs, row0s, iv, hist)
        got[env] = hist.cpu().numpy()
        np.testing.assert_array_equal(got[env], 2 * want + 5)
    # the same rows fed by two segments of one call: no owner, exact sum
    monkeypatch.setenv("FQ_CHAN_OWN_FLUSH", "1")
    a, b = xs[1], torch.flip(xs[1], dims=(0,)) * 0.5
    both = np.zeros((rows, 2048), dtype=np.int64)
    ref_rows(a, 4, both); ref_rows(b, 4, both)
    hist = torch.zeros(rows, 2048, dtype=torch.int64, device="cuda")
    nat.hist2048_chan([a, b], [4, 4], iv, hist)
    np.testing.assert_array_equal(hist.cpu().numpy(), both)
    # partly overlapping row ranges
    both = np.zeros((rows, 2048), dtype=np.int64)
    ref_rows(a, 4, both); ref_rows(xs[0], 10, both)
    hist = torch.zeros(rows, 2048, dtype=torch.int64, device="cuda")
    nat.hist2048_chan([a, xs[0]], [4, 10], iv, hist)
    np.testing.assert_array_equal(hist.cpu().numpy(), both)
    # a histogram buffer 8 bytes off a 16-byte boundary
    flat = torch.zeros(rows * 2048 + 1, dtype=torch.int64, device="cuda")
    off = flat[1:].view(rows, 2048)
    assert off.data_ptr() % 16 == 8
    nat.hist2048_chan(xs, row0s, iv, off)
    np.testing.assert_array_equal(off.cpu().numpy(), want)
    assert int(flat[0]) == 0


def test_channel_histogram_packed_bins_do_not_carry(oracle):
    """Small planes count in 16-bit halves of a dword: a workgroup hands one row at most 32 768 elements, so even when
    they all fall into ONE bin (a constant tensor: every element lands in bin 2047) nothing carries into the
    neighbouring bin.  Odd and even bins, two image groups."""
    from common.quantity import _native as nat
    N, C, H, W = 70, 8, 31, 33                             # HW = 1023 -> 32 images per group: 32 736 elements per row
    x = torch.ones(N, C, H, W, device="cuda")
    x[:, 1] *= 1023.5 / 2048                                # bin 1023 (odd half): iv = fl32(1/2048 + 1e-12) = 2^-11
    x[:, 2] *= 512.5 / 2048                                 # bin 512 (even half)
    iv = torch.full((C,), float(oracle.interval(np.float32(1.0))), device="cuda")
    hist = torch.zeros(C, 2048, dtype=torch.int64, device="cuda")
    nat.hist2048_chan([x], [0], iv, hist)
    hh = hist.cpu().numpy()
    for c in range(C):
        ref = oracle.hist2048(x[:, c].cpu().numpy().ravel(), np.float32(iv[c].item()))
        np.testing.assert_array_equal(hh[c], ref)
    assert hh[0, 2047] == N * H * W and hh[1, 1023] == N * H * W and hh[2, 512] == N * H * W      # odd, odd, even halves


def test_per_channel_calibration_of_a_model():
    """Orchestrator level: per-channel table of the Concat net; pooling all channels of a tensor gives
    back the per-tensor maxima and histogram mass of the reference-compatible path."""
    from tools import Quantity
    with product_workdir(input_shape="1,3,8,8", device="gpu", max_cali_img_num=2) as tmp:
        model = cases.seed_model(cases.tiny_concat_net(), base_seed=7).eval().cuda()
        q = Quantity(model)
        # both calibrations must see the same float forward for the pooled rows to match bit for bit: the per-channel path
        # runs torch's own modules, so the per-tensor path keeps the library's 1x1 convolutions here as well
        q.own_conv1x1 = False
        batches = cases.calib_batches(4, (4, 3, 8, 8), seed=4321)
        per_tensor_bits = q.activation_quantize(batches)
        t_max = {n: float(v) for n, v in q._collector.max_vals.items()}
        t_mass = {n: int(v.sum()) for n, v in q._collector.distributions.items()}
        by_module = q.activation_quantize_per_channel(batches)
        cc = q._channel_collector
        names = ["image"] + list(q.net_info.keys())
        for n in names:
            lo, hi = cc.row_range(n)
            assert float(cc.max_device[lo:hi].max()) == t_max[n]
            assert int(cc.hist_device[lo:hi].sum()) == t_mass[n]
        table = open(os.path.join(tmp, "test", "workdir", "feat_channel.table")).read().strip().split("\n")
        assert len(table) == len(names) and table[0].startswith("image ")
        assert [len(v) for v in by_module.values()] == [3, 8, 8, 8, 16, 8, 8, 8, 5]
        # a channel never needs fewer fractional bits than its whole tensor's range allows
        for (module, b), n in zip(by_module.items(), names):
            assert min(b) >= per_tensor_bits[n] - 1, (module, b, per_tensor_bits[n])


def test_per_channel_calibration_sees_hook_time_values_of_inplace_models():
    """A conv output that an in-place ReLU overwrites is calibrated on the values the hook saw: same per-channel
    table as the same network with out-of-place ReLUs."""
    from test_gpu_e2e import _InplaceNet
    from tools import Quantity
    tables = []
    for inplace in (False, True):
        with product_workdir(input_shape="1,3,16,16", device="gpu", max_cali_img_num=2) as tmp:
            q = Quantity(_InplaceNet(inplace).eval().cuda())
            q.activation_quantize_per_channel(cases.calib_batches(3, (4, 3, 16, 16)))
            tables.append(open(os.path.join(tmp, "test", "workdir", "feat_channel.table")).read())
            assert q._stats_limit == (0 if inplace else 1 << 62)
    assert tables[0] == tables[1] and tables[0].count("\n") == 5


def test_per_channel_activation_cache_does_not_change_the_table(monkeypatch):
    """Pass 2 may histogram the tensors kept from pass 1 instead of running the forward again (whole batches, inside
    the allocator's warm pool): same per-channel table, histograms and maxima either way."""
    from tools import Quantity
    out = []
    for cache_gb in ("0", "1"):
        monkeypatch.setenv("FQ_ACT_CACHE_GB", cache_gb)
        with product_workdir(input_shape="1,3,8,8", device="gpu", max_cali_img_num=3) as tmp:
            q = Quantity(cases.seed_model(cases.tiny_concat_net(), base_seed=7).eval().cuda())
            q.activation_quantize_per_channel(cases.calib_batches(4, (4, 3, 8, 8), seed=4321))
            out.append((open(os.path.join(tmp, "test", "workdir", "feat_channel.table")).read(),
                        q._channel_collector.hist_device.cpu().numpy(), q._channel_collector.max_device.cpu().numpy(),
                        q.timings["per_channel_cache_bytes"]))
    assert out[0][0] == out[1][0]
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])
    assert out[0][3] == 0 and out[1][3] > 0
