"""int8 MFMA implicit-GEMM convolution / linear with fused tail, and the fp32->int8 NHWC quantiser,
against the exact integer oracle.   pytest -m gpu"""
import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from common.quantity import _native
    _native.lib()
    return _native


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N,C,H,W,ib", [(2, 3, 7, 9, 5), (1, 64, 8, 8, 3), (3, 20, 5, 5, 0), (2, 130, 3, 4, 6),
                                        (4, 16, 1, 1, 4), (5, 48, 1, 1, 2), (1, 3, 224, 224, 5)])
def test_quantize_i8_nhwc(nat, oracle, N, C, H, W, ib):
    rng = np.random.default_rng(N * 1000 + C)
    x = (rng.standard_normal((N, C, H, W), dtype=np.float32) * np.float32(3.0)).astype(np.float32)
    x.flat[::17] = np.float32(0.5) / np.float32(2.0 ** ib)          # ties
    y = nat.quantize_i8_nhwc(_dev(x), ib).cpu().numpy()
    cpad = (C + 15) // 16 * 16
    assert y.shape == (N, H, W, cpad)
    ref = oracle.quantity(x, ib).astype(np.int8).transpose(0, 2, 3, 1)
    np.testing.assert_array_equal(y[..., :C], ref)
    assert not y[..., C:].any()


CONV_CASES = [
    # N, C, H, W, K, R, S, stride, pad, dil
    (2, 16, 8, 8, 64, 3, 3, 1, 1, 1),
    (1, 64, 14, 14, 64, 1, 1, 1, 0, 1),
    (3, 32, 9, 7, 40, 3, 3, 2, 1, 1),          # K not a tile multiple, ragged pixels
    (2, 3, 33, 31, 64, 7, 7, 2, 3, 1),         # stem-like, C padded 3 -> 16
    (2, 128, 7, 7, 200, 3, 3, 1, 1, 1),        # K > 128: two k tiles, second partial
    (1, 256, 6, 6, 512, 1, 1, 2, 0, 1),
    (2, 48, 10, 10, 96, 3, 3, 1, 2, 2),        # dilation 2
    (5, 80, 5, 6, 130, 2, 3, 1, 0, 1),         # asymmetric kernel
    (64, 64, 4, 4, 256, 1, 1, 1, 0, 1),        # many images, pixel tiles crossing images
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_i8_vs_integer_oracle(nat, oracle, case):
    N, C, H, W, K, R, S, st, pd, dl = case
    rng = np.random.default_rng(sum(case))
    xq = rng.integers(-128, 128, size=(N, C, H, W)).astype(np.int32)
    wq = rng.integers(-128, 128, size=(K, C, R, S)).astype(np.int32)
    qb = rng.integers(-128, 128, size=K).astype(np.float32)
    acc = oracle.conv2d_int(xq, wq, (st, st), (pd, pd), (dl, dl))
    cpad = (C + 15) // 16 * 16
    x_nhwc = np.zeros((N, H, W, cpad), dtype=np.int8)
    x_nhwc[..., :C] = xq.transpose(0, 2, 3, 1)
    w_dev = nat.pack_weight_krsc(_dev(wq.astype(np.float32)))
    assert w_dev.shape == (K, R, S, cpad)
    for rs, ob in ((8, 3), (12, 5), (0, 0), (15, -1)):
        got = nat.conv2d_i8(_dev(x_nhwc), w_dev, _dev(qb), (st, st), (pd, pd), (dl, dl), rs, ob).cpu().numpy()
        ref = oracle.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
        np.testing.assert_array_equal(got, ref, err_msg="rs=%d ob=%d" % (rs, ob))


def test_a_bias_far_outside_the_output_range_saturates_like_the_reference(nat, oracle):
    """The quantised bias is "integer valued" by contract, not bounded.  The four-instruction tail folds it into the rounding constant
    (qb << rs): a bias of 10^5 or 3 x 10^9 would leave int32 there -- tail_consts first clamps it to the range beyond which the output
    is the Sp bound whatever the accumulator holds.  fp32 output, int8 output (+ ReLU) and the residual-add epilogue, 8- and 16-bit."""
    rng = np.random.default_rng(404)
    N, C, H, W, K = 2, 64, 9, 9, 128
    xq = rng.integers(-128, 128, size=(N, C, H, W)).astype(np.int32)
    wq = rng.integers(-128, 128, size=(K, C, 1, 1)).astype(np.int32)
    qb = rng.integers(-128, 128, size=K).astype(np.float32)
    qb[::7] = [(-1) ** i * v for i, v in enumerate(np.resize([300.0, 1e5, 3e9, 40000.0, 255.0, 256.0], len(qb[::7])))]
    acc = oracle.conv2d_int(xq, wq, (1, 1), (0, 0), (1, 1))
    x_nhwc = np.ascontiguousarray(xq.transpose(0, 2, 3, 1)).astype(np.int8)
    w_dev = nat.pack_weight_krsc(_dev(wq.astype(np.float32)))
    for rs, ob, bw in ((8, 3, 8), (16, 2, 8), (12, 4, 16), (16, 0, 16)):
        got = nat.conv2d_i8(_dev(x_nhwc), w_dev, _dev(qb), (1, 1), (0, 0), (1, 1), rs, ob, bw).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.recon_epilogue(acc.astype(np.float32), qb, rs, ob, bw), err_msg="rs=%d bits=%d" % (rs, bw))
    # ... and on a 1 x 1 plane, where the call runs on the linear layer's kernel (linear_i8_wave_kernel: its own copy of the tail)
    x1 = rng.integers(-128, 128, size=(37, C, 1, 1)).astype(np.int32)
    acc1 = oracle.conv2d_int(x1, wq, (1, 1), (0, 0), (1, 1))
    x1_nhwc = np.ascontiguousarray(x1.transpose(0, 2, 3, 1)).astype(np.int8)
    for rs, ob, bw in ((8, 3, 8), (16, 2, 8), (12, 4, 16), (16, 0, 16)):
        nat.conv_variant_log = seen = {}
        try:
            got = nat.conv2d_i8(_dev(x1_nhwc), w_dev, _dev(qb), (1, 1), (0, 0), (1, 1), rs, ob, bw).cpu().numpy()
        finally:
            nat.conv_variant_log = None
        assert any(k.startswith("linear_wave") for k in seen), seen
        np.testing.assert_array_equal(got, oracle.recon_epilogue(acc1.astype(np.float32), qb, rs, ob, bw), err_msg="1x1 plane rs=%d bits=%d" % (rs, bw))
    # a 1 x 1 plane with padding 1 and stride 3 still has ONE output pixel, but it samples the zero border: bias only.  That call must
    # not take the linear kernel (which ignores the geometry)
    got = nat.conv2d_i8(_dev(x1_nhwc), w_dev, _dev(qb), (3, 3), (1, 1), (1, 1), 8, 3, 8).cpu().numpy()
    accz = oracle.conv2d_int(x1, wq, (3, 3), (1, 1), (1, 1))
    assert accz.shape[2:] == (1, 1) and not accz.any()
    np.testing.assert_array_equal(got, oracle.recon_epilogue(accz.astype(np.float32), qb, 8, 3, 8))
    for relu in (False, True):
        _, q = nat.conv2d_i8_resident(_dev(x_nhwc), w_dev, _dev(qb), (1, 1), (0, 0), (1, 1), 9, 3, False, True, relu)
        ref = oracle.recon_epilogue(acc.astype(np.float32), qb, 9, 0, 8)
        ref = np.maximum(ref, 0) if relu else ref
        np.testing.assert_array_equal(q.cpu().numpy().astype(np.float32), ref.transpose(0, 2, 3, 1))
    # the block-tail kernel's three tails (conv3, the projection, the next conv1)
    C2 = 64
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randint(-128, 128, (2, 6, 6, 64), dtype=torch.int8, device="cuda", generator=g)
    xp = torch.randint(-128, 128, (2, 6, 6, 64), dtype=torch.int8, device="cuda", generator=g)
    w3 = nat.pack_weight_krsc(torch.randint(-127, 128, (256, 64, 1, 1), device="cuda", generator=g).float())
    wp = nat.pack_weight_krsc(torch.randint(-127, 128, (256, 64, 1, 1), device="cuda", generator=g).float())
    w1 = nat.pack_weight_krsc(torch.randint(-127, 128, (C2, 256, 1, 1), device="cuda", generator=g).float())
    big = lambda n: torch.from_numpy(np.resize(np.array([7.0, -1e5, 3e9, -300.0, 90.0, 1e4], dtype=np.float32), n)).cuda()
    b3, bp, b1 = big(256), big(256).flip(0), big(C2)
    _, res = nat.conv2d_i8_resident(xp, wp, bp, (1, 1), (0, 0), (1, 1), 9, 4, False, True, False)
    wide, narrow = nat.conv2d_i8_add_resident(x, w3, b3, (1, 1), (0, 0), (1, 1), 9, 4, res, 4, True, 4, True, 4, True)
    _, q1 = nat.conv2d_i8_resident(narrow, w1, b1, (1, 1), (0, 0), (1, 1), 10, 4, False, True, True)
    got = nat.block_tail_proj_i8(x, w3, b3, 9, 4, xp, wp, bp, 9, 4, 1, True, 4, True, 4, True, w1, b1, 10, True)
    for a, b_ in zip(got, (wide, narrow, q1)):
        assert torch.equal(a, b_)
    # ... and the general kernels' own results against the oracle's chain for the projection (the other two follow from it above)
    accp = oracle.conv2d_int(xp.cpu().numpy().astype(np.int32).transpose(0, 3, 1, 2), wp.cpu().numpy().astype(np.int32).reshape(256, 64, 1, 1))
    np.testing.assert_array_equal(res.cpu().numpy().astype(np.float32),
                                  oracle.recon_epilogue(accp.astype(np.float32), bp.cpu().numpy(), 9, 0, 8).transpose(0, 2, 3, 1))


def test_linear_i8_vs_oracle(nat, oracle):
    rng = np.random.default_rng(9)
    # (a linear layer runs as one wave per 32 x 32 output tile, operands straight from L2 -- linear_i8_wave_kernel; 400 and 84
    #  input features: a last sub-step of 16 bytes; 300 rows, 37 columns: partial tiles both ways)
    for N, F, K in ((4, 512, 10), (64, 2048, 1000), (3, 400, 120), (1, 84, 10), (300, 272, 37), (256, 2048, 1000)):
        xq = rng.integers(-128, 128, size=(N, F)).astype(np.int32)
        wq = rng.integers(-128, 128, size=(K, F)).astype(np.int32)
        qb = rng.integers(-128, 128, size=K).astype(np.float32)
        acc = (xq.astype(np.int64) @ wq.astype(np.int64).T)
        fpad = (F + 15) // 16 * 16
        xp = np.zeros((N, fpad), dtype=np.int8)
        xp[:, :F] = xq
        w_dev = nat.pack_weight_krsc(_dev(wq.astype(np.float32)))
        for rs, ob in ((9, 2), (0, 0), (13, -1), (20, 3)):                         # (0 and 20: the fp32 tail)
            nat.conv_variant_log = log = {}
            try:
                got = nat.conv2d_i8(_dev(xp), w_dev, _dev(qb), (1, 1), (0, 0), (1, 1), rs, ob).cpu().numpy()
            finally:
                nat.conv_variant_log = None
            assert log == {"linear_wave/32": 1}, log
            ref = oracle.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
            np.testing.assert_array_equal(got, ref, err_msg="N=%d F=%d K=%d rs=%d ob=%d" % (N, F, K, rs, ob))


def test_newconv2d_int8_path_equals_float_path(nat):
    """Module level: the MFMA path and the fp32-conv path of NewConv2d agree exactly (|acc| < 2^24)."""
    import torch.nn as nn
    from common.quantity import new_quantity_op as nq
    torch.manual_seed(3)
    for cin, cout, k, st, pd in ((3, 64, 7, 2, 3), (64, 64, 3, 1, 1), (64, 256, 1, 1, 0), (128, 128, 3, 2, 1)):
        conv = nn.Conv2d(cin, cout, k, stride=st, padding=pd, bias=True)
        info = dict(weight_bit=8, bias_bit=4, input_bit=4, output_bit=4)
        m = nq.NewConv2d(conv, info).cuda()
        x = torch.randn(6, cin, 20, 20, device="cuda") * 2
        with torch.no_grad():
            m.use_int8_mfma = True
            a = m(x)
            m.use_int8_mfma = False
            b = m(x)
        assert torch.equal(a, b)
    lin = nn.Linear(512, 10)
    ml = nq.NewLinear(lin, dict(weight_bit=8, bias_bit=3, input_bit=3, output_bit=3)).cuda()
    x = torch.randn(32, 512, device="cuda")
    with torch.no_grad():
        ml.use_int8_mfma = True
        a = ml(x)
        ml.use_int8_mfma = False
        b = ml(x)
    assert torch.equal(a, b)


@pytest.mark.parametrize("C,H,W,K,R,S,st,pd,dl", [(3, 33, 31, 64, 7, 7, 2, 3, 1), (1, 28, 28, 6, 3, 3, 1, 1, 1), (4, 16, 20, 10, 5, 3, 1, 2, 2),
                                                   (3, 224, 224, 64, 7, 7, 2, 3, 1),
                                                   (2, 12, 40, 8, 3, 11, 1, 5, 1)])        # S > 8: generic unfold kernel
def test_stem_unfold_path_vs_integer_oracle(nat, oracle, C, H, W, K, R, S, st, pd, dl):
    """Kernel width folded into the channel axis (stem layers): same integers as the plain convolution."""
    rng = np.random.default_rng(C * 100 + H)
    N = 2
    x = (rng.standard_normal((N, C, H, W), dtype=np.float32) * np.float32(3)).astype(np.float32)
    wq = rng.integers(-128, 128, size=(K, C, R, S)).astype(np.int32)
    qb = rng.integers(-128, 128, size=K).astype(np.float32)
    ib = 4
    xq_ref = oracle.quantity(x, ib).astype(np.int32)
    acc = oracle.conv2d_int(xq_ref, wq, (st, st), (pd, pd), (dl, dl))
    cpad2 = (S * C + 15) // 16 * 16
    xq = nat.quantize_i8_unfold_w(_dev(x), ib, S, st, pd, dl, cpad2)
    w_dev = nat.pack_weight_unfold_w(_dev(wq.astype(np.float32)), cpad2)
    got = nat.conv2d_i8(xq, w_dev, _dev(qb), (st, 1), (pd, 0), (dl, 1), 9, 3).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.recon_epilogue(acc.astype(np.float32), qb, 9, 3))


@pytest.mark.parametrize("rs", [-1, 0, 1, 2, 3, 7, 8, 15, 16, 17, 20])
def test_conv_tail_rounding_ties_and_saturation(nat, oracle, rs):
    """The fused tail runs in integer arithmetic for 1 <= rs <= 16 and as the reference's fp32 chain
    otherwise; both must equal the oracle on accumulators that sit exactly on rounding ties (+-half),
    on both sides of them, and far beyond the int8 range."""
    C, K, H, W = 16, 64, 16, 16
    xq = np.zeros((1, C, H, W), dtype=np.int32)
    xq[0, 0] = np.arange(-128, 128).reshape(H, W)                     # every int8 value once
    xq[0, 1] = 127
    wq = np.zeros((K, C, 1, 1), dtype=np.int32)
    wq[:, 0, 0, 0] = np.arange(K) - 32                                # acc = x * (k - 32) + 127 * w1
    wq[:, 1, 0, 0] = np.where(np.arange(K) % 3 == 0, 127, 0)          # pushes some rows far out of range
    qb = ((np.arange(K) % 7) - 3).astype(np.float32)
    acc = oracle.conv2d_int(xq, wq, (1, 1), (0, 0), (1, 1))
    x_nhwc = np.ascontiguousarray(xq.transpose(0, 2, 3, 1)).astype(np.int8)
    w_dev = nat.pack_weight_krsc(_dev(wq.astype(np.float32)))
    for ob in (0, 4):
        ref = oracle.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
        got = nat.conv2d_i8(_dev(x_nhwc), w_dev, _dev(qb), (1, 1), (0, 0), (1, 1), rs, ob).cpu().numpy()
        np.testing.assert_array_equal(got, ref)
        y, q = nat.conv2d_i8_resident(_dev(x_nhwc), w_dev, _dev(qb), (1, 1), (0, 0), (1, 1), rs, ob, True, True, True)
        np.testing.assert_array_equal(y.cpu().numpy(), np.maximum(ref, np.float32(0)))
        np.testing.assert_array_equal(q.cpu().numpy().transpose(0, 3, 1, 2),
                                      oracle.quantity(np.maximum(ref, np.float32(0)), ob).astype(np.int8))


def test_grouped_and_circular_convolutions_take_the_reference_shaped_path(nat, oracle):
    """fq_conv2d_i8 covers groups == 1 with zero padding.  A grouped (depth-wise) convolution or a circular padding
    mode must take the reference-shaped forward instead (Quantity kernel -> fp32 conv on integer-valued data -> fused
    tail kernel), also inside a resident plan, and give the oracle's integers."""
    from common.quantity import new_quantity_op as nq
    from common.quantity import resident
    torch.manual_seed(4)
    info = dict(weight_bit=7, bias_bit=4, input_bit=4, output_bit=4)
    for conv in (nn.Conv2d(16, 32, 3, padding=1, groups=4), nn.Conv2d(8, 8, 3, padding=1, groups=8),
                 nn.Conv2d(16, 16, 3, padding=1, padding_mode="circular")):
        w_float, b_float = conv.weight.detach().clone(), conv.bias.detach().clone()
        m = nq.NewConv2d(conv, info).cuda()
        assert not m._int8_ok(m.Conv)
        x = torch.randn(3, conv.in_channels, 9, 9, device="cuda") * 2
        with torch.no_grad():
            got = m(x).cpu().numpy()
        # oracle: the reference's chain with an fp32 convolution over the integer-valued operands (exact here)
        xq = torch.from_numpy(oracle.quantity(x.cpu().numpy(), 4))
        wq = torch.clamp(torch.round(w_float * 2 ** 7), -128, 127)
        qb = torch.clamp(torch.round(b_float * 2 ** 4), -128, 127).numpy()
        ref_conv = nn.Conv2d(conv.in_channels, conv.out_channels, 3, padding=1, groups=conv.groups,
                             padding_mode=conv.padding_mode, bias=False)
        with torch.no_grad():
            ref_conv.weight.copy_(wq)
            acc = ref_conv(xq).numpy()
        np.testing.assert_array_equal(got, oracle.recon_epilogue(acc, qb, 7, 4))
    # inside a resident plan the layer stays an fp32-boundary producer and the model still reproduces itself
    net = nn.Sequential(nq.NewConv2d(nn.Conv2d(16, 16, 3, padding=1), dict(info)), nn.ReLU(),
                        nq.NewConv2d(nn.Conv2d(16, 16, 3, padding=1, groups=16), dict(info)), nn.ReLU(),
                        nq.NewConv2d(nn.Conv2d(16, 8, 1), dict(info))).cuda().eval()
    x = torch.randn(2, 16, 10, 10, device="cuda")
    with torch.no_grad():
        plain = net(x)
    resident.enable(net, x)
    with torch.no_grad():
        assert torch.equal(net(x), plain)
