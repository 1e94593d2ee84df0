"""fq_conv3x3_wino_f32 -- the stride-1 3x3 convolutions of the float calibration forward as Winograd F(2x2, 3x3) on the fp32
matrix cores -- through the C ABI: exact agreement with a float64 convolution on integer-valued data (U = G g Gt is then exact
in quarters and every sum is exact, so a wrong index, tile, mask or transform coefficient shows as a wrong value), agreement
within the product's bound on Gaussian data and closeness to the direct kernel, the folded abs-max / histogram / ReLU copy bit
for bit on the SAME output, odd planes, 1x1 planes, tiles that cross images, partial tile blocks, independence of the batch
size, the packed weights against their definition, error codes; and the product: ResNet-50 calibrated with the Winograd
kernels and with the direct ones gives the same feat.table.    pytest -m gpu"""
import numpy as np
import pytest
import torch

import cases
from workdir_util import product_workdir

pytestmark = pytest.mark.gpu

# N, Cin, Cout, H, W
SHAPES = [
    (2, 64, 64, 56, 56),          # ResNet-50 layer1: 1 568 tiles, 8 steps
    (3, 128, 128, 28, 28),        # layer2: two k-blocks
    (5, 256, 256, 14, 14),        # layer3: 7x7 tiles per image, 245 tiles = 3.8 blocks
    (4, 512, 512, 7, 7),          # layer4: odd plane, 4x4 tiles with a missing row and column; 64 tiles = one block
    (3, 8, 64, 5, 9),             # one step; odd both ways
    (70, 16, 128, 1, 4),          # one-row planes: two tiles per image, half of their pixels missing, six taps padding
    (1, 8, 64, 2, 3),             # a single partial block
    (7, 24, 64, 13, 6),           # three steps, odd height
    (33, 16, 192, 10, 10),        # 825 tiles = 12.9 blocks x 3 k-blocks: more work items than one round of a persistent grid
    (2, 40, 64, 6, 20),           # wide plane
    # (Cin >= 128: the half-size work item -- 64 channels x 32 tiles, two workgroups per CU, two channels per transform wave)
    (3, 128, 64, 5, 9),           # odd both ways, 45 tiles = 1.4 blocks
    (70, 128, 128, 2, 2),         # one tile per image
    (1, 136, 64, 2, 3),           # a single partial block, 17 steps
    # the end of the tensor: the last tile's row H - 1 is loaded from the plane's last four pixels and shifted (tile_in)
    (1, 8, 64, 2, 2),             # tile 0 is also the last tile: both shifts in one tile
    (1, 8, 64, 1, 4),
    (2, 16, 64, 4, 1),            # W = 1: every tile is the last of its row
    (1, 8, 64, 3, 3),             # odd H: row H - 1 is in two tile rows
    (2, 8, 64, 7, 2),
    (3, 128, 64, 3, 5),           # the half-size work item
    (40, 8, 64, 7, 9),            # odd H, the two tile rows with row H - 1 in different tile blocks
]


@pytest.fixture(scope="module")
def nat():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from common.quantity import _native
    _native.lib()
    return _native


def _case(shape, seed, integer):
    N, cin, cout, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(seed)
    if integer:
        x = torch.randint(-8, 9, (N, cin, H, W), device="cuda", generator=g).float()
        w = torch.randint(-8, 9, (cout, cin, 3, 3), device="cuda", generator=g).float()
        b = torch.randint(-100, 101, (cout,), device="cuda", generator=g).float()
    else:
        x = torch.randn(N, cin, H, W, device="cuda", generator=g)
        w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (cin * 9) ** -0.5
        b = torch.randn(cout, device="cuda", generator=g)
    return x, w, b


def _ref64(x, w, b):
    return torch.nn.functional.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_exact_on_integer_valued_data(nat, shape):
    # |U| <= 18 in quarters, |V| <= 32, 512 channels: every product and partial sum is a multiple of 1/4 below 2^22 -- exact
    x, w, b = _case(shape, 51, integer=True)
    u = nat.pack_wino_weight(w)
    assert torch.equal(nat.conv_wino_f32(x, u, b, shape[2]).double(), _ref64(x, w, b))
    assert torch.equal(nat.conv_wino_f32(x, u, None, shape[2]).double(), _ref64(x, w, None))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_gaussian_data_statistics_and_relu(nat, shape):
    x, w, b = _case(shape, 52, integer=False)
    cout = shape[2]
    u = nat.pack_wino_weight(w)
    ref = _ref64(x, w, b)
    bound = torch.nn.functional.conv2d(x.abs().double(), w.abs().double(), b.abs().double(), padding=1)
    y = nat.conv_wino_f32(x, u, b, cout)
    # the product's once-per-module bound (_float_conv.TOL), and -- measured, DESIGN.md section 4 -- a tenth of it in practice
    assert bool(((y.double() - ref).abs() <= 1e-5 * bound).all())
    assert float(((y.double() - ref).abs() / bound).max()) < 1e-6
    assert torch.equal(y, nat.conv_wino_f32(x, u, b, cout))                     # deterministic
    if shape[1] % 16 == 0:
        direct = nat.conv_kxk_f32(x, nat.pack_kxk_weight(w), b, (3, 3), 1, 1)
        assert float(((y.double() - direct.double()).abs() / bound).max()) < 1e-6
    mx = torch.tensor([0.0, 1e9, 0.0], device="cuda")
    r = torch.empty_like(y)
    y1 = nat.conv_wino_f32(x, u, b, cout, max_dev=mx, row=2, relu_out=r)
    assert torch.equal(y1, y) and torch.equal(r, torch.relu(y)) and mx.tolist() == [0.0, 1e9, float(y.abs().max())]
    iv = torch.tensor([1.0, float(y.abs().max()) / 2048 + 1e-12], device="cuda")
    hist = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
    hist[1, 3] = 11
    want = hist.clone()
    y2 = nat.conv_wino_f32(x, u, b, cout, interval_dev=iv, hist_dev=hist, row=1, relu_out=r)
    nat.hist2048_seg([y], [1], iv, want)
    assert torch.equal(y2, y) and torch.equal(r, torch.relu(y)) and torch.equal(hist, want)
    # only the ReLU's output wanted: y is not written, its statistic is still the convolution output's
    mx2 = torch.zeros(1, device="cuda")
    r2 = torch.full_like(y, 7.0)
    assert nat.conv_wino_f32(x, u, b, cout, max_dev=mx2, row=0, relu_out=r2, out=False) is None
    assert torch.equal(r2, torch.relu(y)) and float(mx2[0]) == float(y.abs().max())


@pytest.mark.parametrize("bit,bitwidth", [(5, 8), (-2, 8), (9, 16)])
def test_quandequan_epilogue_equals_the_two_pass_form(nat, bit, bitwidth):
    """TestConv.forward (new_quantity_op.py:283-292) on a stride-1 3x3 layer as one kernel: the value QuanDequan sees is the
    kernel's own sum, so the result is fq_quandequan_f32 of the plain output bit for bit -- saturation, odd planes included."""
    for shape in ((3, 64, 64, 14, 14), (4, 24, 128, 7, 7)):
        x, w, b = _case(shape, 57, integer=False)
        x *= 40.0                                               # (some outputs beyond the 8-bit range at bit 5)
        u = nat.pack_wino_weight(w)
        plain = nat.conv_wino_f32(x, u, b, shape[2])
        want = nat.quandequan(plain.clone(), bit, bitwidth)
        got = nat.conv_wino_f32(x, u, b, shape[2], qd=(bit, bitwidth))
        assert torch.equal(got, want)
    with pytest.raises(nat.FqError):
        nat.conv_wino_f32(x, u, b, shape[2], qd=(bit, bitwidth), relu_out=torch.empty_like(plain))
    assert nat.lib().fq_conv3x3_wino_qd_f32(x.data_ptr(), u.data_ptr(), None, plain.data_ptr(), 4, 24, 7, 7, 128, 3, 12, None) == -1


def test_x_may_end_where_its_allocation_ends():
    """The kernel reads nothing outside [x, x + x_bytes): x is put at the very end of an allocation of its own (hipMalloc, not
    the caching allocator, whose blocks have neighbours), for an even and an odd W -- the 16-byte row load of the last tile
    used to end 4 / 8 bytes behind the tensor for the last input channel (the channel rides in the scalar offset, which the
    address unit's range check does not see).  In a child process: a fault would end the process, not the test run."""
    import os, subprocess, sys
    code = r'''
import ctypes, sys, torch
sys.path.insert(0, sys.argv[1])
from common.quantity import _native as nat
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
L = nat.lib()
for (N, cin, cout, H, W) in ((2, 64, 64, 32, 32), (2, 64, 64, 8, 31), (1, 128, 64, 32, 32), (1, 128, 64, 31, 8)):
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randint(-8, 9, (N, cin, H, W), device="cuda", generator=g).float()
    w = torch.randint(-8, 9, (cout, cin, 3, 3), device="cuda", generator=g).float()
    u = nat.pack_wino_weight(w)
    nbytes = x.numel() * 4
    total = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    base = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(base), total) == 0
    xp = base.value + total - nbytes                      # x ends where the allocation ends
    torch.cuda.synchronize()
    assert hip.hipMemcpy(xp, x.data_ptr(), nbytes, 3) == 0
    y = torch.empty(N, cout, H, W, device="cuda")
    rc = L.fq_conv3x3_wino_f32(xp, u.data_ptr(), None, y.data_ptr(), None, N, cin, H, W, cout, None, None, None, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, padding=1)
    assert torch.equal(y.double(), ref), (N, cin, cout, H, W)
print("ok")
'''
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch-quantity_amd", "quantity")
    out = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]


def test_an_image_computes_the_same_bits_in_any_batch(nat):
    """No K split, no workspace: what a pixel is does not depend on how many images ride in the launch or where in it the
    image sits (the direct kernel's tail split does make the last bit depend on the tile count -- include/fq.h)."""
    x, w, b = _case((37, 256, 256, 14, 14), 53, integer=False)
    u = nat.pack_wino_weight(w)
    whole = nat.conv_wino_f32(x, u, b, 256)
    for lo, hi in ((0, 1), (5, 6), (3, 20), (30, 37)):
        assert torch.equal(nat.conv_wino_f32(x[lo:hi].contiguous(), u, b, 256), whole[lo:hi])


def test_nan_and_inf_stay_where_the_direct_sum_puts_them_or_next_to_it(nat):
    """An Inf or NaN input reaches every output whose 4x4 input tile holds it (a transform adds and subtracts it): a superset
    of the outputs the direct sum poisons, never a finite value where the direct sum has none; the abs-max ignores NaN."""
    x, w, b = _case((2, 16, 64, 8, 8), 54, integer=False)
    x[1, 3, 4, 4] = float("nan")
    u = nat.pack_wino_weight(w)
    mx = torch.zeros(1, device="cuda")
    y = nat.conv_wino_f32(x, u, b, 64, max_dev=mx, row=0)
    direct = torch.nn.functional.conv2d(x, w, b, padding=1)
    assert bool(torch.isnan(y)[torch.isnan(direct)].all()) and not bool(torch.isnan(y[0]).any())
    assert float(mx[0]) == float(torch.nan_to_num(y, nan=0.0).abs().max())


def test_packed_weights_are_g_gt_in_fp64_rounded_once(nat):
    g = torch.Generator(device="cuda").manual_seed(55)
    w = torch.randn(64, 24, 3, 3, device="cuda", generator=g)
    u = nat.pack_wino_weight(w).cpu().numpy()
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
    wd = w.cpu().numpy().astype(np.float64)                                     # [k][c][3][3]
    U = np.einsum("ar,kcrs,bs->kcab", G, wd, G)                                  # [k][c][4][4]
    U = U.reshape(64, 24, 16)
    want = np.zeros(16 * 24 * 64, dtype=np.float32).reshape(3, 16, 2, 64, 4)    # [c / 8][e][c % 2][k][c % 8 / 2]
    for c in range(24):
        want[c // 8, :, c % 2, :, (c % 8) // 2] = U[:, c, :].T.astype(np.float32)
    # (the kernel sums the three terms of a row of G in a fixed order in fp64: allow one fp32 ulp against einsum's order)
    assert np.allclose(u.reshape(want.shape), want, rtol=2e-7, atol=0)
    assert nat.lib().fq_conv3x3_wino_f32_packed_floats(24, 64) == 16 * 24 * 64


def test_argument_errors(nat):
    L = nat.lib()
    x = torch.zeros(1, 16, 4, 4, device="cuda")
    u = torch.zeros(16 * 16 * 64, device="cuda")
    y = torch.zeros(1, 64, 4, 4, device="cuda")
    mx = torch.zeros(1, device="cuda")
    hist = torch.zeros(2048, dtype=torch.int64, device="cuda")
    P = lambda t: t.data_ptr()
    ok = lambda *a: L.fq_conv3x3_wino_f32(*a)
    assert ok(P(x), P(u), None, P(y), None, 1, 16, 4, 4, 64, None, None, None, None) == 0
    assert ok(P(x), P(u), None, P(y), None, 0, 16, 4, 4, 64, None, None, None, None) == 0          # no images: nothing to do
    assert ok(P(x), P(u), None, P(y), None, 1, 12, 4, 4, 64, None, None, None, None) == -4         # Cin % 8
    assert ok(P(x), P(u), None, P(y), None, 1, 16, 4, 4, 32, None, None, None, None) == -4         # Cout % 64
    assert ok(P(x), P(u) + 4, None, P(y), None, 1, 16, 4, 4, 64, None, None, None, None) == -4     # u not 16-byte aligned
    assert ok(P(x), P(u), None, None, None, 1, 16, 4, 4, 64, None, None, None, None) == -1         # neither y nor relu_out
    assert ok(None, P(u), None, P(y), None, 1, 16, 4, 4, 64, None, None, None, None) == -1
    assert ok(P(x), P(u), None, P(y), None, 1, 16, 4, 4, 64, P(mx), P(mx), P(hist), None) == -1    # both statistics
    assert ok(P(x), P(u), None, P(y), None, 1, 16, 4, 4, 64, None, None, P(hist), None) == -1      # histogram without its interval
    assert ok(P(x), P(u), None, P(y), None, 1, 16, 0, 4, 64, None, None, None, None) == -1
    assert L.fq_conv3x3_wino_f32_supported(256, 64, 56, 56, 64) == 1
    assert L.fq_conv3x3_wino_f32_supported(1024, 256, 56, 56, 256) == 0                            # x beyond 2^31 bytes
    assert L.fq_conv3x3_wino_f32_pack(None, P(u), 8, 64, None) == -1
    assert L.fq_conv3x3_wino_f32_pack(P(x), P(u), 4, 64, None) == -4
    assert L.fq_conv3x3_wino_f32_supported(4, 12, 8, 8, 64) == 0 and L.fq_conv3x3_wino_f32_supported(4, 8, 8, 8, 64) == 1
    # planes of fewer than four pixels stay on the direct kernel (the end-of-tensor row is loaded from a plane's last four)
    assert L.fq_conv3x3_wino_f32_supported(4, 8, 1, 3, 64) == 0 and L.fq_conv3x3_wino_f32_supported(4, 8, 1, 4, 64) == 1


def test_resnet50_tables_do_not_depend_on_the_3x3_kernel(monkeypatch):
    """The product: the fabu ResNet-50 calibrated with its stride-1 3x3 layers on the Winograd kernel and on the direct one --
    same feat.table (the two differ in the last bits of an activation, a table entry is a power of two)."""
    from common.quantity import merge_bn
    from model.resnet.ResNet_fabu import ResNet50
    from tools import Quantity
    model = merge_bn(cases.seed_model(ResNet50(input_size=64)).eval()).cuda()
    batches = cases.calib_batches(4, (4, 3, 64, 64), seed=56)
    tables, used = [], []
    for env in ("1", "0"):
        monkeypatch.setenv("FQ_CONV_WINO", env)
        calls = []
        real = __import__("common.quantity._native", fromlist=["x"]).conv_wino_f32
        monkeypatch.setattr("common.quantity._native.conv_wino_f32", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with product_workdir(input_shape="1,3,64,64", device="gpu", max_cali_img_num=3) as tmp:
            q = Quantity(model)
            bits = q.activation_quantize(batches)
            tables.append((dict(bits), open(tmp + "/test/workdir/feat.table").read()))
        used.append(len(calls))
        monkeypatch.undo()
    assert tables[0] == tables[1]
    assert used[0] > 0 and used[1] == 0
