"""Resident integer activations (common.quantity.resident): the integer-simulation model with int8 /
int16 NHWC hand-offs between layers must reproduce the fp32-boundary model -- and therefore the
reference -- bit for bit.   pytest -m gpu"""
import os

import numpy as np
import pytest
import torch
from torch import nn

import cases
from workdir_util import product_workdir

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from common.quantity import _native
    _native.lib()
    return _native


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


RES_CASES = [
    # N, C, H, W, K, R, S, stride, pad
    (2, 16, 8, 8, 64, 3, 3, 1, 1),             # TK = 64, general path
    (3, 32, 9, 7, 40, 3, 3, 2, 1),             # K = 40 -> Kpad 48, ragged pixel tile
    (2, 128, 7, 7, 200, 3, 3, 1, 1),           # two k tiles, second partial, Kpad 208
    (1, 256, 6, 6, 512, 1, 1, 2, 0),           # C % 128 fast path, TK = 64 (few workgroups)
    (40, 128, 14, 14, 256, 1, 1, 1, 0),        # C % 128 fast path, TK = 128
    (2, 3, 20, 20, 10, 3, 3, 1, 1),            # K = 10 -> Kpad 16
]


@pytest.mark.parametrize("case", RES_CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_conv_resident_outputs(nat, oracle, case, relu):
    """q is what the NEXT layer's Quantity(ib = ob) recovers from the reference's fp32 output (after the
    ReLU when fused); y is the fp32 output itself; both-outputs mode gives the same two arrays."""
    N, C, H, W, K, R, S, st, pd = case
    rng = np.random.default_rng(sum(case) + int(relu))
    xq = rng.integers(-128, 128, size=(N, C, H, W)).astype(np.int32)
    wq = rng.integers(-128, 128, size=(K, C, R, S)).astype(np.int32)
    qb = rng.integers(-128, 128, size=K).astype(np.float32)
    acc = oracle.conv2d_int(xq, wq, (st, st), (pd, pd), (1, 1))
    cpad = (C + 15) // 16 * 16
    kpad = (K + 15) // 16 * 16
    x_nhwc = np.zeros((N, H, W, cpad), dtype=np.int8)
    x_nhwc[..., :C] = xq.transpose(0, 2, 3, 1)
    w_dev = nat.pack_weight_krsc(_dev(wq.astype(np.float32)))
    x_dev, b_dev = _dev(x_nhwc), _dev(qb)
    for rs, ob in ((12, 5), (9, 2), (16, -1)):
        ref = oracle.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
        if relu:
            ref = np.maximum(ref, np.float32(0))
        ref_q = oracle.quantity(ref, ob).astype(np.int8).transpose(0, 2, 3, 1)
        for want_f32, want_i8 in ((False, True), (True, True), (True, False)):
            y, q = nat.conv2d_i8_resident(x_dev, w_dev, b_dev, (st, st), (pd, pd), (1, 1), rs, ob, want_f32, want_i8, relu)
            if want_f32:
                np.testing.assert_array_equal(y.cpu().numpy(), ref)
            else:
                assert y is None
            if want_i8:
                got = q.cpu().numpy()
                assert got.shape == ref_q.shape[:3] + (kpad,)
                np.testing.assert_array_equal(got[..., :K], ref_q)
                assert not got[..., K:].any()
            else:
                assert q is None


STEM_CASES = [
    # N, C, H, W, K, R, S, stride, pad
    (2, 3, 224, 224, 64, 7, 7, 2, 3),          # the ResNet stem
    (3, 3, 32, 32, 64, 3, 3, 1, 1),            # the 32x32 ResNet-18 stem
    (2, 1, 28, 28, 20, 5, 5, 1, 0),            # one channel, K = 20 -> Kpad 32, no padding
    (1, 4, 37, 53, 33, 7, 5, 2, 2),            # ragged tiles on both axes, R != S, Kpad 48
    (2, 2, 19, 23, 16, 8, 8, 2, 4),            # the limits: 8 x 8 taps
    (1, 3, 9, 9, 64, 3, 3, 2, 0),              # a single partial tile
]


@pytest.mark.parametrize("case", STEM_CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_stem_kernel_equals_oracle_and_the_unfold_path(nat, oracle, case, relu):
    """fq_conv2d_i8_stem (fp32 image -> int8 NHWC in one kernel) against the reference chain Quantity -> integer
    conv -> RightShift -> BiasAdd -> Sp (-> ReLU) -> next Quantity, and against the two-kernel unfold path."""
    N, C, H, W, K, R, S, st, pd = case
    rng = np.random.default_rng(sum(case) + 7 * int(relu))
    x = (rng.standard_normal((N, C, H, W)) * 1.7).astype(np.float32)
    x.flat[::97] = 0.0
    x.flat[5::211] *= 40.0                                                 # saturating pixels
    wq = rng.integers(-128, 128, size=(K, C, R, S)).astype(np.int32)
    qb = rng.integers(-128, 128, size=K).astype(np.float32)
    assert nat.stem_supported(C, K, R, S, (st, st), (1, 1), 9)
    w_stem = nat.pack_weight_stem(_dev(wq.astype(np.float32)))
    x_dev, b_dev = _dev(x), _dev(qb)
    kpad = (K + 15) // 16 * 16
    for ib, rs, ob in ((5, 12, 4), (6, 9, 2), (4, 16, -1), (7, 1, 6)):
        xq = oracle.quantity(x, ib).astype(np.int32)
        acc = oracle.conv2d_int(xq, wq, (st, st), (pd, pd), (1, 1))
        ref = oracle.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
        if relu:
            ref = np.maximum(ref, np.float32(0))
        ref_q = np.zeros(ref.shape[:1] + ref.shape[2:] + (kpad,), dtype=np.int8)
        ref_q[..., :K] = oracle.quantity(ref, ob).astype(np.int8).transpose(0, 2, 3, 1)
        got = nat.conv2d_i8_stem(x_dev, w_stem, b_dev, K, S, (st, st), (pd, pd), ib, rs, ob, relu).cpu().numpy()
        assert got.shape == ref_q.shape
        assert np.array_equal(got, ref_q), (case, ib, rs, ob, int((got != ref_q).sum()))
        # the two-kernel path the general layers use computes the same integers
        fold = nat.pad16(S * C)
        w_fold = nat.pack_weight_unfold_w(_dev(wq.astype(np.float32)), fold)
        xu = nat.quantize_i8_unfold_w(x_dev, ib, S, st, pd, 1, fold)
        _y, q2 = nat.conv2d_i8_resident(xu, w_fold, b_dev, (st, 1), (pd, 0), (1, 1), rs, ob, False, True, relu)
        assert torch.equal(q2.cpu(), torch.from_numpy(got))


def test_stem_kernel_refuses_what_it_does_not_cover(nat):
    x = torch.zeros(1, 5, 8, 8, device="cuda")
    w = torch.zeros(3, 64, 32, dtype=torch.int8, device="cuda")
    b = torch.zeros(8, device="cuda")
    assert not nat.stem_supported(5, 8, 3, 3, (1, 1), (1, 1), 9)           # five input channels
    assert not nat.stem_supported(3, 8, 3, 3, (1, 1), (2, 2), 9)           # dilation
    assert not nat.stem_supported(3, 8, 3, 3, (1, 1), (1, 1), 0)           # shift outside the integer tail
    assert not nat.stem_supported(3, 8, 7, 7, (6, 6), (1, 1), 9)           # patch larger than the LDS budget
    with pytest.raises(nat.FqError):
        nat.conv2d_i8_stem(x, w, b, 8, 3, (1, 1), (1, 1), 5, 9, 3, False)
    with pytest.raises(nat.FqError):
        nat.conv2d_i8_stem(torch.zeros(1, 3, 8, 8), w, b, 8, 3, (1, 1), (1, 1), 5, 9, 3, False)    # CPU tensor


@pytest.mark.parametrize("xb,gx,yb,gy,ib,relu", [(1, 3, 1, 5, 4, True), (1, 5, 2, 5, 3, True), (2, 6, 1, 2, 6, False),
                                                 (2, 8, 2, 7, 5, True), (1, -1, 1, 2, 0, False), (1, 0, 2, 0, 1, True)])
def test_add_resident_equals_fp32_chain(nat, oracle, xb, gx, yb, gy, ib, relu):
    """DeQuantity -> NewAdd -> ReLU -> Quantity of the reference, evaluated by the oracle on fp32, against
    the fused integer kernel; the int16 output must be the exact sum."""
    rng = np.random.default_rng(xb * 1000 + gx * 100 + yb * 10 + gy)
    shape = (3, 5, 7, 32)
    x = rng.integers(-128, 128, size=shape).astype(np.int8 if xb == 1 else np.int16)
    if xb == 2:
        x = (x.astype(np.int32) * rng.integers(1, 2 ** gx + 1, size=shape)).clip(-128 * 2 ** gx, 127 * 2 ** gx).astype(np.int16)
    y = rng.integers(-128, 128, size=shape).astype(np.int8 if yb == 1 else np.int16)
    if yb == 2:
        y = (y.astype(np.int32) * rng.integers(1, 2 ** gy + 1, size=shape)).clip(-128 * 2 ** gy, 127 * 2 ** gy).astype(np.int16)
    g = max(0, gx, gy)
    xf = oracle.dequantity(x.astype(np.float32), gx)
    yf = oracle.dequantity(y.astype(np.float32), gy)
    s = oracle.add_sat(xf, yf)
    if relu:
        s = np.maximum(s, np.float32(0))
    exact = s.astype(np.float64) * 2.0 ** g
    assert np.all(exact == np.rint(exact)) and np.abs(exact).max() <= 32768
    wide, narrow = nat.add_resident(_dev(x), gx, _dev(y), gy, True, g, True, ib, relu)
    np.testing.assert_array_equal(wide.cpu().numpy(), exact.astype(np.int16))
    np.testing.assert_array_equal(narrow.cpu().numpy(), oracle.quantity(s, ib).astype(np.int8))
    w2, n2 = nat.add_resident(_dev(x), gx, _dev(y), gy, True, g, False, 0, relu)
    assert n2 is None
    np.testing.assert_array_equal(w2.cpu().numpy(), wide.cpu().numpy())
    w3, n3 = nat.add_resident(_dev(x), gx, _dev(y), gy, False, g, True, ib, relu)
    assert w3 is None
    np.testing.assert_array_equal(n3.cpu().numpy(), narrow.cpu().numpy())


@pytest.mark.parametrize("case", [(2, 64, 9, 9, 256, 1, 1, 1, 0), (3, 128, 7, 7, 200, 3, 3, 1, 1), (1, 1024, 6, 6, 512, 1, 1, 1, 0)])
@pytest.mark.parametrize("res_dtype,g_res,relu", [(np.int8, 3, True), (np.int16, 6, True), (np.int16, 5, False)])
def test_conv_add_fused_equals_conv_then_add(nat, oracle, case, res_dtype, g_res, relu):
    """fq_conv2d_i8_add_resident == fq_conv2d_i8_resident followed by fq_add_resident == the oracle's fp32 chain
    DeQuantity -> NewAdd -> ReLU -> Quantity, for both conv kernels (register-staged and LDS-DMA)."""
    N, C, H, W, K, R, S, st, pd = case
    rng = np.random.default_rng(sum(case) + g_res)
    xq = rng.integers(-128, 128, size=(N, C, H, W)).astype(np.int32)
    wq = rng.integers(-8, 9, size=(K, C, R, S)).astype(np.int32)
    qb = rng.integers(-20, 20, size=K).astype(np.float32)
    rs, ob, ib = 10, 4, 4
    acc = oracle.conv2d_int(xq, wq, (st, st), (pd, pd), (1, 1))
    conv_f = oracle.recon_epilogue(acc.astype(np.float32), qb, rs, ob)                       # [N,K,P,Q] fp32
    kpad = (K + 15) // 16 * 16
    P, Q = conv_f.shape[2], conv_f.shape[3]
    info = np.iinfo(res_dtype)
    lim = 128 * 2 ** g_res if res_dtype == np.int16 else 128
    res = np.zeros((N, P, Q, kpad), dtype=res_dtype)
    res[..., :K] = rng.integers(max(info.min, -lim), min(info.max, lim - 1) + 1, size=(N, P, Q, K))
    res_f = oracle.dequantity(res[..., :K].astype(np.float32), g_res).transpose(0, 3, 1, 2)
    s = oracle.add_sat(conv_f, res_f)
    if relu:
        s = np.maximum(s, np.float32(0))
    g = max(0, ob, g_res)
    exact = (s.astype(np.float64) * 2.0 ** g)
    assert np.all(exact == np.rint(exact))
    x_nhwc = np.ascontiguousarray(xq.transpose(0, 2, 3, 1)).astype(np.int8)
    w_dev = nat.pack_weight_krsc(_dev(wq.astype(np.float32)))
    wide, narrow = nat.conv2d_i8_add_resident(_dev(x_nhwc), w_dev, _dev(qb), (st, st), (pd, pd), (1, 1), rs, ob, _dev(res), g_res,
                                              True, g, True, ib, relu)
    np.testing.assert_array_equal(wide.cpu().numpy()[..., :K].transpose(0, 3, 1, 2), exact.astype(np.int16))
    np.testing.assert_array_equal(narrow.cpu().numpy()[..., :K].transpose(0, 3, 1, 2), oracle.quantity(s, ib).astype(np.int8))
    assert not wide.cpu().numpy()[..., K:].any() and not narrow.cpu().numpy()[..., K:].any()
    # and the two-kernel form gives the same bytes
    _, cq = nat.conv2d_i8_resident(_dev(x_nhwc), w_dev, _dev(qb), (st, st), (pd, pd), (1, 1), rs, ob, False, True, False)
    w2, n2 = nat.add_resident(cq, ob, _dev(res), g_res, True, g, True, ib, relu)
    assert torch.equal(w2, wide) and torch.equal(n2, narrow)


def test_add_resident_refuses_a_sum_that_does_not_fit(nat):
    x = torch.zeros(2, 2, 2, 16, dtype=torch.int8, device="cuda")
    with pytest.raises(nat.FqError):
        nat.add_resident(x, 9, x, 2, True, 9, False, 0, False)
    with pytest.raises(nat.FqError):
        nat.add_resident(x, 3, x, 2, True, 4, False, 0, False)          # wrong grid for the exact sum


@pytest.mark.parametrize("dtype,g,N,C,H,W", [(np.int8, 4, 2, 64, 7, 7), (np.int16, 7, 3, 40, 5, 9), (np.int8, -1, 1, 10, 1, 1),
                                             (np.int16, 0, 2, 130, 13, 3)])
def test_dequant_nhwc_to_nchw(nat, oracle, dtype, g, N, C, H, W):
    rng = np.random.default_rng(C + g)
    cpad = (C + 15) // 16 * 16
    info = np.iinfo(dtype)
    q = rng.integers(info.min, info.max + 1, size=(N, H, W, cpad)).astype(dtype)
    y = nat.dequant_nhwc_to_nchw(_dev(q), g, C).cpu().numpy()
    ref = oracle.dequantity(q[..., :C].astype(np.float32), g).transpose(0, 3, 1, 2)
    np.testing.assert_array_equal(y, ref)


@pytest.mark.parametrize("N,C,H,W,k,st,pd", [(2, 64, 17, 17, 3, 2, 1), (1, 20, 9, 12, 2, 2, 0), (3, 16, 8, 8, 3, 1, 1), (2, 48, 7, 7, 5, 3, 2),
                                             # the banded 3x3 / 2 / 1 kernel (one workgroup per 8 output rows of an image): ResNet's
                                             # stem plane, odd and even planes, a partial last band, one-row and two-pixel planes, a
                                             # plane too wide for one workgroup (generic kernel)
                                             (3, 64, 112, 112, 3, 2, 1), (2, 32, 35, 20, 3, 2, 1), (1, 16, 2, 2, 3, 2, 1),
                                             (2, 48, 3, 9, 3, 2, 1), (1, 256, 40, 40, 3, 2, 1), (1, 64, 30, 200, 3, 2, 1)])
def test_maxpool_i8_equals_torch_on_the_values(nat, N, C, H, W, k, st, pd):
    """Pooling the integers == quantising torch's max-pool of the de-quantised tensor."""
    rng = np.random.default_rng(C * H)
    cpad = (C + 15) // 16 * 16
    q = np.zeros((N, H, W, cpad), dtype=np.int8)
    q[..., :C] = rng.integers(-128, 128, size=(N, H, W, C))
    g = 3
    xf = torch.from_numpy(q[..., :C].astype(np.float32) / 2 ** g).permute(0, 3, 1, 2).contiguous().cuda()
    ref = torch.nn.functional.max_pool2d(xf, k, st, pd)
    got = nat.maxpool_i8_nhwc(_dev(q), (k, k), (st, st), (pd, pd))
    assert tuple(got.shape) == (N, ref.shape[2], ref.shape[3], cpad)
    np.testing.assert_array_equal(got.cpu().numpy()[..., :C].astype(np.float32) / 2 ** g, ref.permute(0, 2, 3, 1).cpu().numpy())


@pytest.mark.parametrize("dtype,g,N,C,H", [(np.int16, 6, 4, 2048, 7), (np.int8, 3, 2, 100, 4), (np.int16, 0, 3, 64, 16), (np.int16, 8, 2, 48, 13),
                                           # one phase, 256 groups of 8 channels; more groups than threads (two sweeps); a single group
                                           (np.int16, 5, 300, 2048, 7), (np.int16, 2, 2, 4000, 3), (np.int8, 4, 3, 8200, 2), (np.int8, 1, 5, 10, 5)])
def test_avgpool_global_equals_torch_avg_pool2d(nat, dtype, g, N, C, H):
    """Bit-identical to torch's AvgPool2d(H) on the de-quantised fp32 tensor (exact partial sums, one division)."""
    rng = np.random.default_rng(C + H)
    cpad = (C + 15) // 16 * 16
    info = np.iinfo(dtype)
    q = rng.integers(info.min, info.max + 1, size=(N, H, H, cpad)).astype(dtype)
    xf = torch.from_numpy(q[..., :C].astype(np.float32) * np.float32(2.0 ** -g)).permute(0, 3, 1, 2).contiguous().cuda()
    ref = torch.nn.AvgPool2d(H)(xf)
    got = nat.avgpool_global_nhwc(_dev(q), g, C)
    assert torch.equal(got, ref), float((got - ref).abs().max())


def _r18_recon(g3, tmp):
    from model.resnet.ResNet_18_fabu import ResNet18
    from tools import Reconstruction
    wd = os.path.join(tmp, "test", "workdir")
    os.makedirs(wd, exist_ok=True)
    with open(os.path.join(wd, "feat.table"), "w") as fh:
        fh.write(g3["feat_table"])
    with open(os.path.join(wd, "weight.table"), "w") as fh:
        fh.write(g3["weight_table_after_second_rewrite"])
    rec = Reconstruction(cases.seed_model(ResNet18()).eval())
    rec.merge_bn()
    return rec.ReconModel(rec.get_quantity_information(), os.path.join(wd, "recon.pth")).cuda()


@pytest.fixture(scope="module")
def g3(golden_dir):
    import json
    with open(os.path.join(golden_dir, "g3_r18_e2e.json")) as fh:
        return json.load(fh)


def test_r18_resident_logits_equal_reference_golden(golden_dir, g3):
    """ResNet-18 (basic blocks, identity and projection shortcuts, folded stem): the resident model must
    give the logits the reference's CPU ReconModel gave (golden G4), and the plan must be non-trivial."""
    from common.quantity import resident
    g4 = np.load(os.path.join(golden_dir, "g4_r18_recon.npz"))
    with product_workdir(device="gpu") as tmp:
        net = _r18_recon(g3, tmp)
        x = torch.from_numpy(g4["x"]).cuda()
        with torch.no_grad():
            plain = net(x).cpu().numpy()
        np.testing.assert_array_equal(plain, g4["logits_recon"])
        summary = resident.enable(net, x)
        assert resident.is_enabled(net)
        assert summary["resident_convs"] >= 15 and summary["fused_relus"] >= 10, summary
        with torch.no_grad():
            got = net(x).cpu().numpy()
            again = net(torch.cat([x, x])).cpu().numpy()          # another batch size, same plan
        np.testing.assert_array_equal(got, g4["logits_recon"])
        np.testing.assert_array_equal(again[:x.shape[0]], g4["logits_recon"])
        np.testing.assert_array_equal(again[x.shape[0]:], g4["logits_recon"])
        resident.disable(net)
        assert not resident.is_enabled(net) and not resident.describe(net)
        with torch.no_grad():
            np.testing.assert_array_equal(net(x).cpu().numpy(), plain)


def _calibrated_recon(model_fn, image, batch, channels=3):
    """Calibrate a seeded model on the GPU and rebuild it as ReconModel; returns (net, input batch)."""
    from tools import Quantity, Reconstruction
    from common.quantity import merge_bn
    dev = torch.device("cuda")
    model = merge_bn(cases.seed_model(model_fn()).eval()).to(dev)
    gen = np.random.default_rng(5)
    data = [(torch.from_numpy(gen.standard_normal((batch, channels, image, image), dtype=np.float32)).to(dev), 0) for _ in range(2)]
    q = Quantity(model)
    q.activation_quantize(data)
    q.weight_quantize()
    q.rewrite_weight()
    rec = Reconstruction(merge_bn(cases.seed_model(model_fn()).eval()).to(dev))
    net = rec.ReconModel(rec.get_quantity_information(), "./workdir/recon.pth")
    return net, data[0][0]


def test_r50_resident_equals_fp32_boundary_model():
    """Bottleneck ResNet-50 @64x64: every conv but the classifier and every Eltwise goes resident, all
    49 ReLUs are fused, and logits (and an intermediate stage output) are bit-identical."""
    from common.quantity import resident
    from model.resnet.ResNet_fabu import ResNet50
    with product_workdir(input_shape="1,3,64,64", device="gpu"):
        net, x = _calibrated_recon(lambda: ResNet50(num_classes=100, input_size=64), 64, 8)
        with torch.no_grad():
            plain = net(x)
            stage_plain = net.layer2[:3](net.layer1(net.maxpool(net.relu(net.conv1(x)))))
        summary = resident.enable(net, x)
        assert summary["resident_convs"] == 53 and summary["resident_adds"] == 16 and summary["fused_relus"] == 49, summary
        assert summary["resident_pools"] == 2 and summary["fp32_outputs"] == 0, summary     # max-pool and global average pool
        assert summary["fused_conv_adds"] == 16, summary                                    # every conv3 runs inside its add
        plans = resident.describe(net)
        assert plans["layer1.0.conv1"].emit_f32 is False and plans["layer1.0.conv1"].relu is True
        assert plans["conv1"].emit_f32 is False and plans["conv1"].emit_int is True and plans["conv1"].relu is True
        assert plans["maxpool"].emit_f32 is False and plans["maxpool"].emit_int is True
        assert plans["layer4.2.Eltwise"].emit_f32 is False and plans["layer4.2.Eltwise"].want_wide is True   # -> average pool
        assert plans["layer1.1.Eltwise"].emit_f32 is False and plans["layer1.1.Eltwise"].want_wide is True
        with torch.no_grad():
            got = net(x)
            stage = net.layer2[:3](net.layer1(net.maxpool(net.relu(net.conv1(x)))))   # feeds a conv and an add
        assert type(stage).__name__ == "QHandle"
        assert torch.equal(got, plain)
        assert torch.equal(stage.to_f32(), stage_plain)
        # the whole resident forward captures as one HIP graph (no sync, no allocation outside torch's pool)
        graphed = resident.capture(net, x)
        assert torch.equal(graphed(x), plain)
        x2 = torch.flip(x, dims=[0])
        with torch.no_grad():
            want2 = net(x2)
        assert torch.equal(graphed(x2), want2)
        # ... and as two graphs of half the batch replayed on two streams (the kernels of one half start while the other
        # half's drain): same logits, also on another input
        dual = resident.capture(net, x, streams=2)
        assert torch.equal(dual(x), plain) and torch.equal(dual(x2), want2) and torch.equal(dual(x), plain)
        with pytest.raises(Exception):
            resident.capture(net, x[:3], streams=2)
        # a handle that reaches code outside the plan fails loudly instead of computing garbage
        with pytest.raises(Exception):
            torch.relu(stage)
        # the plan survives pickling of the whole model (how the reference stores models)
        torch.save(net, "./workdir/recon_resident.pth")
        again = torch.load("./workdir/recon_resident.pth", weights_only=False)
        with torch.no_grad():
            assert torch.equal(again(x), plain)


def test_foreign_consumers_and_shared_relu():
    """A net that mixes integer layers with things the plan does not own: a functional op on a conv
    output, one nn.ReLU instance used at three call sites, a Concat, a value used by both a conv and a
    user op.  Resident mode must give identical outputs."""
    from common.quantity import NewConv2d, NewAdd, resident

    def info(i, o, w=6):
        return {"weight_bit": w, "bias_bit": o, "input_bit": i, "output_bit": o}

    class Net(nn.Module):
        def __init__(self):
            super(Net, self).__init__()
            torch.manual_seed(3)
            self.c1 = NewConv2d(nn.Conv2d(3, 32, 3, padding=1), info(5, 4))
            self.c2 = NewConv2d(nn.Conv2d(32, 32, 3, padding=1), info(4, 3))
            self.c3 = NewConv2d(nn.Conv2d(32, 32, 1), info(3, 3))
            self.c4 = NewConv2d(nn.Conv2d(32, 48, 1), info(3, 2))
            self.c5 = NewConv2d(nn.Conv2d(96, 16, 1), info(2, 2))
            self.add = NewAdd()
            self.relu = nn.ReLU()                             # shared by every call site

        def forward(self, x):
            a = self.relu(self.c1(x))                         # fused, int8 only
            b = self.relu(self.c2(a))                         # fused; consumed by c3 AND by a user op below
            c = self.c3(b)                                    # no ReLU; feeds the add
            d = self.relu(self.add(c, b))                     # resident add, fused ReLU
            e = self.c4(d)
            f = torch.cat((e, e * 0.5 + b.mean()), dim=1)     # foreign consumers of e and b
            return self.c5(self.relu(f))                      # ReLU on a foreign tensor: runs normally

    net = Net().cuda().eval()
    x = torch.randn(4, 3, 12, 12, device="cuda")
    with torch.no_grad():
        plain = net(x)
    summary = resident.enable(net, x)
    plans = resident.describe(net)
    assert plans["c1"].emit_f32 is False and plans["c1"].relu
    assert plans["c2"].emit_f32 is True and plans["c2"].emit_int is True and plans["c2"].relu
    assert plans["c3"].emit_f32 is False and not plans["c3"].relu
    assert plans["add"].resident_add and plans["add"].relu
    assert plans["c4"].emit_f32 is True and plans["c4"].emit_int is False
    assert summary["fused_relus"] == 3
    with torch.no_grad():
        assert torch.equal(net(x), plain)
        assert torch.equal(net(x[:1]), plain[:1])


@pytest.mark.parametrize("tag", ["concat", "lenet", "vgg", "separable"])
def test_small_nets_with_concat_and_pool_only_topologies(tag, monkeypatch):
    """Nets whose integer layers feed things the plan does not own (a Concat marker layer, View, Linear
    chains), a plain VGG-like stack (2x2 max-pools between integer layers) and depthwise-separable blocks (grouped convolutions
    keep the reference's fp32 form): whatever goes resident, the logits must not change, and Concat operands must stay fp32."""
    from common.quantity import resident
    if tag == "concat":
        fn, shape, image, ch = cases.tiny_concat_net, "1,3,8,8", 8, 3
        monkeypatch.setattr(torch, "save", lambda *a, **k: None)       # the fixture net is a local class: not picklable
    elif tag in ("vgg", "separable"):
        fn, shape, image, ch = (cases.tiny_vgg_net if tag == "vgg" else cases.tiny_separable_net), "1,3,16,16", 16, 3
        monkeypatch.setattr(torch, "save", lambda *a, **k: None)
    else:
        fn, shape, image, ch = (lambda: __import__("model.lenet.lenet", fromlist=["Cnn"]).Cnn(1, 10)), "1,1,28,28", 28, 1
    with product_workdir(input_shape=shape, device="gpu"):
        net, x = _calibrated_recon(fn, image, 4, channels=ch)
        with torch.no_grad():
            plain = net(x)
        summary = resident.enable(net, x)                     # verifies bit-identity on x itself
        plans = resident.describe(net)
        if tag == "concat":
            assert plans["branch_a"].emit_f32 and plans["branch_b"].emit_f32, plans          # Concat is foreign code
            assert plans["stem"].relu and plans["Eltwise"].resident_add, plans
        if tag == "vgg":
            assert plans["p1"].emit_int and plans["p2"].emit_int and summary["resident_pools"] >= 2, plans     # int8 max-pools
        if tag == "separable":
            assert "dw1" not in plans or plans["dw1"].emit_f32, plans                     # a grouped convolution is not an integer layer
        assert summary["resident_convs"] >= 2
        with torch.no_grad():
            assert torch.equal(net(x), plain)
            assert torch.equal(net(torch.flip(x, dims=[0])), torch.flip(plain, dims=[0]))


def test_hipgraph_capture_of_a_model_whose_first_conv_is_not_a_stem_layer():
    """A first integer conv with 16 input channels (and one with a 1-wide kernel) quantises its fp32 input through
    the one-entry fq_quantize_i8_nhwc memo instead of the stem kernel.  The memo must not serve, inside a capture, a
    copy made before it (the quantise launch would be missing from the graph and every replay would convolve the
    capture-time input): replaying on a DIFFERENT input must give that input's logits."""
    from common.quantity import NewConv2d, NewAdd, resident, new_quantity_op

    def info(i, o, w=6):
        return {"weight_bit": w, "bias_bit": o, "input_bit": i, "output_bit": o}

    class Net(nn.Module):
        def __init__(self, cin, k):
            super(Net, self).__init__()
            torch.manual_seed(11)
            self.c1 = NewConv2d(nn.Conv2d(cin, 32, k, padding=k // 2), info(5, 4))
            self.sc = NewConv2d(nn.Conv2d(cin, 32, 1), info(5, 4))        # shortcut on the SAME input: served by the memo
            self.c2 = NewConv2d(nn.Conv2d(32, 32, 3, padding=1), info(4, 4))
            self.add = NewAdd()
            self.relu = nn.ReLU()

        def forward(self, x):
            return self.relu(self.add(self.c2(self.relu(self.c1(x))), self.sc(x)))

    for cin, k in ((16, 3), (3, 1)):
        net = Net(cin, k).cuda().eval()
        x = torch.randn(4, cin, 10, 10, device="cuda")
        x2 = torch.randn(4, cin, 10, 10, device="cuda") * 2
        for use_plan in (False, True):
            if use_plan:
                resident.enable(net, x)
            with torch.no_grad():
                want, want2 = net(x).clone(), net(x2).clone()
            assert not torch.equal(want, want2)
            graphed = resident.capture(net, x)
            assert torch.equal(graphed(x), want)
            assert torch.equal(graphed(x2), want2)
            assert torch.equal(graphed(x), want)
            with torch.no_grad():                                         # eager calls after a capture: no pool-private copy served
                assert torch.equal(net(x2), want2)
        assert new_quantity_op._xq_cache._val is None or new_quantity_op._xq_cache._ref() is not None
