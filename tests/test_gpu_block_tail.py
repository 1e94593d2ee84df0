"""fq_block_tail_i8: conv3 (1x1 expand) + NewAdd with the shortcut (+ ReLU) + the next block's conv1 (1x1 reduce + ReLU) as ONE
kernel (reference: new_quantity_op.py:124-133 twice and :166-174).  It must leave bit for bit what the two launches it
replaces leave -- fq_conv2d_i8_add_resident (itself pinned to the CPU oracle's fp32 chain in tests/test_gpu_resident.py) followed
by fq_conv2d_i8_resident -- and, independently, what the oracle's chain gives on the same integers: ragged pixel counts, int8
and int16 shortcuts, every subset of {sum, re-quantisation} outputs, with and without the fused next convolution, all grids
the integer add takes (packed int16, 32-bit) and the float fallback.   pytest -m gpu"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    assert torch.cuda.is_available()
    from common.quantity import _native
    _native.lib()
    return _native


def _operands(nat, N, H, W, C, K3, C2, res_dtype, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randint(-128, 128, (N, H, W, C), dtype=torch.int8, device="cuda", generator=g)
    w3 = nat.pack_weight_krsc(torch.randint(-127, 128, (K3, C, 1, 1), device="cuda", generator=g).float())
    b3 = torch.randint(-100, 101, (K3,), device="cuda", generator=g).float()
    lim = 3000 if res_dtype == torch.int16 else 128
    res = torch.randint(-lim, lim, (N, H, W, K3), dtype=res_dtype, device="cuda", generator=g)
    w1 = b1 = None
    if C2:
        w1 = nat.pack_weight_krsc(torch.randint(-127, 128, (C2, K3, 1, 1), device="cuda", generator=g).float())
        b1 = torch.randint(-100, 101, (C2,), device="cuda", generator=g).float()
    return x, w3, b3, res, w1, b1


def _two_launches(nat, x, w3, b3, rs3, ob3, res, g_res, want_wide, g_wide, want_narrow, ib, relu, w1, b1, rs1, relu1):
    wide, narrow = nat.conv2d_i8_add_resident(x, w3, b3, (1, 1), (0, 0), (1, 1), rs3, ob3, res, g_res, want_wide, g_wide,
                                              True, ib, relu)
    q1 = None
    if w1 is not None:
        _, q1 = nat.conv2d_i8_resident(narrow, w1, b1, (1, 1), (0, 0), (1, 1), rs1, ib, False, True, relu1)
    return wide, (narrow if want_narrow else None), q1


CASES = [
    # N, H, W, C, K3, C2, shortcut, (ob3, g_res, ib), rs3, rs1, relu, relu1, want_wide, want_narrow
    (3, 9, 7, 64, 256, 64, torch.int16, (4, 5, 4), 9, 10, True, True, True, False),      # stage-1 shape, 189 pixels (ragged tile)
    (2, 12, 12, 64, 256, 64, torch.int8, (4, 4, 4), 9, 10, True, True, True, False),     # a stage's first block: int8 shortcut
    (2, 8, 8, 128, 512, 128, torch.int16, (4, 5, 4), 10, 11, True, True, True, False),   # stage-2 shape: four slices
    (1, 16, 16, 128, 512, 128, torch.int8, (3, 5, 4), 10, 11, True, True, True, True),
    (2, 7, 9, 64, 256, 128, torch.int16, (4, 5, 3), 8, 9, True, False, True, True),
    (2, 7, 9, 128, 384, 64, torch.int16, (5, 5, 5), 9, 9, False, True, False, True),     # k = 0: the 32-bit integer add, no ReLU
    (2, 10, 10, 64, 256, 0, torch.int16, (4, 5, 4), 9, 0, True, False, True, True),      # conv3 + NewAdd alone
    (2, 10, 10, 128, 128, 0, torch.int8, (4, 5, 4), 9, 0, True, False, False, True),
    (5, 14, 14, 64, 256, 64, torch.int16, (4, 6, 2), 9, 10, True, True, True, False),    # k = 4
    (1, 1, 1, 64, 128, 64, torch.int16, (4, 5, 4), 9, 10, True, True, True, True),       # one pixel
    (3, 7, 7, 256, 1024, 0, torch.int16, (4, 5, 4), 10, 0, True, False, True, True),     # stage-3 shape: eight slices, 256-byte weight rows
    (2, 5, 5, 256, 384, 0, torch.int8, (3, 4, 4), 11, 0, False, False, True, False),
    (23, 56, 56, 64, 256, 64, torch.int16, (4, 5, 4), 9, 10, True, True, True, False),   # 564 tiles: more than two per CU
    (9, 56, 56, 64, 256, 0, torch.int8, (4, 4, 4), 9, 0, True, False, True, True),
    (5, 33, 31, 64, 128, 64, torch.int16, (4, 5, 4), 9, 10, True, True, True, True),     # K3 = 128: one slice per tile
]


UNSUPPORTED_GRIDS = []          # (every grid the two launches take is taken: packed, 32-bit and fp32 forms of NewAdd)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%dx%d_%dto%dto%d_%s" % (c[0], c[1], c[2], c[3], c[4], c[5], str(c[6])[-5:]))
def test_block_tail_equals_the_two_launches_it_replaces(nat, case):
    N, H, W, C, K3, C2, res_dtype, (ob3, g_res, ib), rs3, rs1, relu, relu1, want_wide, want_narrow = case
    x, w3, b3, res, w1, b1 = _operands(nat, N, H, W, C, K3, C2, res_dtype, seed=N * 1000 + K3 + C2)
    g_wide = max(0, ob3, g_res)
    rb = 2 if res_dtype == torch.int16 else 1
    ref = _two_launches(nat, x, w3, b3, rs3, ob3, res, g_res, want_wide, g_wide, want_narrow, ib, relu, w1, b1, rs1, relu1)
    if not nat.block_tail_supported(C, K3, C2, rs3, rs1, ob3, g_res, rb, ib):
        # grids outside NewAdd's packed-int16 form: the kernel refuses, callers keep the two launches (which take every grid)
        assert case in UNSUPPORTED_GRIDS
        with pytest.raises(nat.FqError):
            nat.block_tail_i8(x, w3, b3, rs3, ob3, res, g_res, want_wide, g_wide, want_narrow, ib, relu, w1, b1, rs1, relu1)
        return
    assert case not in UNSUPPORTED_GRIDS
    nat.conv_variant_log = log = {}
    try:
        got = nat.block_tail_i8(x, w3, b3, rs3, ob3, res, g_res, want_wide, g_wide, want_narrow, ib, relu, w1, b1, rs1, relu1)
    finally:
        nat.conv_variant_log = None
    assert log == {"block_tail/128": 1}
    for name, a, b in zip(("wide", "narrow", "q1"), got, ref):
        assert (a is None) == (b is None), name
        if a is not None:
            assert torch.equal(a, b), "%s differs in %d of %d" % (name, int((a != b).sum()), a.numel())
    if C2:
        assert int((got[2] != 0).sum()) > 0                       # (not a vacuous comparison)


PROJ_CASES = [
    # N, H, W (of the block's output), stride of the projection, K3, C2, (ob3, obp, ib), rs3, rsp, rs1, relu, want_wide, want_narrow
    (2, 12, 12, 1, 256, 64, (4, 4, 4), 9, 8, 10, True, True, False),       # ResNet-50 stage 1, block 0
    (3, 9, 7, 1, 256, 64, (4, 5, 3), 9, 9, 10, True, True, True),          # 189 pixels: a ragged tile; different grids
    (2, 7, 5, 2, 256, 64, (3, 5, 4), 8, 10, 9, True, True, True),          # stride 2: the shortcut reads every other pixel of a 13 x 9 plane
    (1, 6, 6, 2, 384, 0, (5, 5, 5), 9, 9, 0, False, True, True),           # no next conv1, no ReLU, 32-bit form of the add, 12 x 11 plane
    (2, 10, 10, 1, 128, 0, (4, 5, 4), 9, 9, 0, True, False, True),         # one slice, narrow only
    (9, 56, 56, 1, 256, 64, (4, 4, 4), 9, 9, 10, True, True, False),       # 221 tiles
    (1, 1, 1, 1, 128, 64, (4, 5, 4), 9, 10, 10, True, True, True),         # one pixel
]


@pytest.mark.parametrize("case", PROJ_CASES, ids=lambda c: "%dx%dx%d_s%d_to%dto%d" % (c[0], c[1], c[2], c[3], c[4], c[5]))
def test_block_tail_with_the_projection_inside_equals_the_launches_it_replaces(nat, case):
    """fq_block_tail_proj_i8 (the tail of a stage's FIRST block: conv3 + NewAdd + ReLU + the next conv1, with the projection
    shortcut -- a 1x1 NewConv2d of the block's input, stride 1 or 2 -- computed in the kernel) against fq_conv2d_i8_resident for
    the projection followed by fq_block_tail_i8 on its output, and against the general kernels' chain: every output bit for bit."""
    N, H, W, sp, K3, C2, (ob3, obp, ib), rs3, rsp, rs1, relu, want_wide, want_narrow = case
    C = CP = 64
    x, w3, b3, _res, w1, b1 = _operands(nat, N, H, W, C, K3, C2, torch.int8, seed=N * 100 + K3 + C2 + sp)
    g = torch.Generator(device="cuda").manual_seed(N * 7 + K3 + sp)
    Hp, Wp = (H - 1) * sp + 1 + (sp - 1) * (N % 2), (W - 1) * sp + 1                 # (an even plane too: its last row is never read)
    xp = torch.randint(-128, 128, (N, Hp, Wp, CP), dtype=torch.int8, device="cuda", generator=g)
    wp = nat.pack_weight_krsc(torch.randint(-127, 128, (K3, CP, 1, 1), device="cuda", generator=g).float())
    bp = torch.randint(-100, 101, (K3,), device="cuda", generator=g).float()
    g_wide = max(0, ob3, obp)
    assert nat.block_tail_proj_supported(C, K3, C2, CP, rs3, rs1, rsp, sp)
    _, res = nat.conv2d_i8_resident(xp, wp, bp, (sp, sp), (0, 0), (1, 1), rsp, obp, False, True, False)
    assert tuple(res.shape) == (N, H, W, K3)
    ref = nat.block_tail_i8(x, w3, b3, rs3, ob3, res, obp, want_wide, g_wide, want_narrow, ib, relu, w1, b1, rs1, True)
    ref2 = _two_launches(nat, x, w3, b3, rs3, ob3, res, obp, want_wide, g_wide, want_narrow, ib, relu, w1, b1, rs1, True)
    nat.conv_variant_log = log = {}
    try:
        got = nat.block_tail_proj_i8(x, w3, b3, rs3, ob3, xp, wp, bp, rsp, obp, sp, want_wide, g_wide, want_narrow, ib, relu, w1, b1, rs1,
                                     True)
    finally:
        nat.conv_variant_log = None
    assert log == {"block_tail_proj/128": 1}
    for name, a, b, c in zip(("wide", "narrow", "q1"), got, ref, ref2):
        assert (a is None) == (b is None), name
        if a is not None:
            assert torch.equal(a, b) and torch.equal(a, c), "%s differs in %d of %d" % (name, int((a != b).sum()), a.numel())
    assert int((res != 0).sum()) > res.numel() // 2                              # (the shortcut is not vacuous)


def test_block_tail_proj_argument_errors(nat):
    L = nat.lib()
    assert L.fq_block_tail_proj_i8_supported(64, 256, 64, 64, 9, 10, 9, 1) == 1 and L.fq_block_tail_proj_i8_supported(64, 256, 0, 64, 9, 0, 9, 2) == 1
    assert L.fq_block_tail_proj_i8_supported(128, 512, 128, 256, 9, 10, 9, 2) == 0       # deeper stages keep the projection's own launch
    assert L.fq_block_tail_proj_i8_supported(64, 256, 64, 64, 9, 10, 0, 1) == 0          # no integer tail for the projection
    assert L.fq_block_tail_proj_i8_supported(64, 256, 64, 64, 9, 10, 9, 3) == 0
    x = torch.zeros(4096, dtype=torch.int8, device="cuda")
    P = x.data_ptr()
    call = lambda *a: L.fq_block_tail_proj_i8(*a)
    # plane sizes that are not the 1x1 convolution's: (Hp - 1) / stride + 1 != H
    assert call(P, P, P, 9, 4, P, P, P, 9, 4, 2, 9, 9, P, 4, None, 4, 1, None, None, 0, 0, None, 1, 4, 4, 64, 256, 0, 64, None) == -1
    assert call(P, P, P, 9, 4, None, P, P, 9, 4, 1, 4, 4, P, 4, None, 4, 1, None, None, 0, 0, None, 1, 4, 4, 64, 256, 0, 64, None) == -1
    assert call(P, P, P, 9, 4, P, P, P, 9, 4, 1, 4, 4, P, 4, None, 4, 1, None, None, 0, 0, None, 1, 4, 4, 128, 512, 0, 256, None) == -4
    assert call(P, P, P, 9, 4, P, P, P, 9, 4, 1, 4, 4, P, 4, None, 4, 1, None, None, 0, 0, None, 0, 4, 4, 64, 256, 0, 64, None) == 0   # no images


def test_block_tail_against_the_oracle_chain(nat, oracle):
    """The same chain through the CPU oracle's element-wise ops on the exact integer convolution (no GPU kernel involved in the
    expected values): conv -> RightShift -> + bias -> Sp -> DeQuantity | add | clamp | ReLU | Quantity -> conv -> tail -> ReLU."""
    N, H, W, C, K3, C2 = 2, 6, 5, 64, 256, 64
    ob3, g_res, ib, rs3, rs1 = 4, 5, 4, 9, 10
    x, w3, b3, res, w1, b1 = _operands(nat, N, H, W, C, K3, C2, torch.int16, seed=77)
    wide, narrow, q1 = nat.block_tail_i8(x, w3, b3, rs3, ob3, res, g_res, True, 5, True, ib, True, w1, b1, rs1, True)
    xi = x.cpu().numpy().astype(np.int32).transpose(0, 3, 1, 2)
    w3i = w3.cpu().numpy().astype(np.int32).reshape(K3, C, 1, 1)
    acc = oracle.conv2d_int(xi, w3i).astype(np.float32)                              # [N, K3, H, W]
    conv_val = oracle.recon_epilogue(acc, b3.cpu().numpy(), rs3, ob3, 8)            # RightShift + bias + Sp + DeQuantity
    res_val = res.cpu().numpy().astype(np.float32).transpose(0, 3, 1, 2) * np.float32(2.0 ** -g_res)
    s = np.maximum(oracle.add_sat(conv_val.ravel(), res_val.ravel(), 8), 0).astype(np.float32)        # NewAdd, then nn.ReLU
    want_wide = (s * np.float32(2.0 ** 5)).reshape(N, K3, H, W).transpose(0, 2, 3, 1)
    assert np.array_equal(wide.cpu().numpy().astype(np.float32), want_wide)
    qn = oracle.quantity(s, ib, 8).reshape(N, K3, H, W)                              # the next layer's Quantity(ib)
    assert np.array_equal(narrow.cpu().numpy().astype(np.float32), qn.transpose(0, 2, 3, 1))
    acc1 = oracle.conv2d_int(qn.astype(np.int32), w1.cpu().numpy().astype(np.int32).reshape(C2, K3, 1, 1)).astype(np.float32)
    t1 = np.maximum(oracle.recon_epilogue(acc1, b1.cpu().numpy(), rs1, 0, 8), 0)     # ob = 0: the integers in front of DeQuantity
    assert np.array_equal(q1.cpu().numpy().astype(np.float32), t1.transpose(0, 2, 3, 1))


def test_block_tail_argument_errors(nat):
    L = nat.lib()
    ok = (4, 5, 2, 4)                                                        # ob3, g_res, int16 shortcut, ib: the packed add's grids
    assert L.fq_block_tail_i8_supported(64, 256, 64, 9, 10, *ok) == 1
    assert L.fq_block_tail_i8_supported(256, 1024, 256, 9, 10, *ok) == 0     # no fused next conv behind 256 -> 1024
    assert L.fq_block_tail_i8_supported(256, 1024, 0, 9, 0, *ok) == 1 and L.fq_block_tail_i8_supported(512, 2048, 0, 9, 0, *ok) == 0
    assert L.fq_block_tail_i8_supported(64, 192, 64, 9, 10, *ok) == 0        # K3 not a multiple of 128
    assert L.fq_block_tail_i8_supported(64, 256, 64, 0, 10, *ok) == 0        # no integer tail
    assert L.fq_block_tail_i8_supported(64, 256, 64, 9, 10, 5, 5, 2, 5) == 1  # k = 0: the 32-bit form of the add
    assert L.fq_block_tail_i8_supported(64, 256, 64, 9, 10, 5, 4, 3, 4) == 0  # a 3-byte shortcut
    x = torch.zeros(1, 1, 1, 64, dtype=torch.int8, device="cuda")
    assert L.fq_block_tail_i8(x.data_ptr(), x.data_ptr(), x.data_ptr(), 9, 4, None, 2, 5, None, 5, None, 4, 1, None, None, 0, 0, None,
                              1, 64, 256, 0, None) == -1                      # no shortcut
    assert L.fq_block_tail_i8(None, None, None, 9, 4, x.data_ptr(), 2, 5, None, 5, None, 4, 1, None, None, 0, 0, None, 0, 64, 256, 0,
                              None) == 0                                      # no pixels: nothing to do
