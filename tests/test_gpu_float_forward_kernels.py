"""The kernels the float calibration forward runs instead of torch's (fq_conv1x1_f32, fq_conv_stem_f32 further down,
fq_maxpool2d_f32 / fq_avgpool_global_f32 at the end).  First fq_conv1x1_f32 -- the float 1x1 convolution on the fp32
matrix cores with the calibration's statistic in its epilogue -- through the C ABI: exact agreement with a float64 reference on integer-valued data (every partial sum is exact, so any
indexing or tiling mistake shows as a wrong bit), agreement within the summation-order bound on Gaussian data (tolerance
1e-5 * (|W| * |x| + |b|), the bound the product's own once-per-module check uses), the folded abs-max / histogram / ReLU
copy bit for bit against the streaming kernels and torch on the SAME output, ragged shapes (K tails, partial tiles in both
directions, strides, 7x7 and 1x1 planes, one image), the non-temporal form, determinism, error codes; and the product:
a ResNet-style net calibrated with and without the kernel gives the same tables.    pytest -m gpu"""
import ctypes

import numpy as np
import pytest
import torch

import cases
from workdir_util import product_workdir

pytestmark = pytest.mark.gpu

# N, Cin, Cout, H, W, stride
SHAPES = [
    (9, 256, 640, 28, 28, 1),     # tail split: 560 narrow tiles = 2 rounds of 256 + 48, each of the 48 cut into 2 K slices
    (130, 256, 1024, 8, 16, 1),   # tail split: 1 040 tiles of 128 x 128, the last 16 cut into 4 K slices
    (256, 2048, 512, 7, 7, 1),    # ResNet-50 layer4 conv1 at the bench's batch: 784 tiles, the last 16 in 16 slices
    (2, 64, 64, 56, 56, 1),       # the narrow 64 x 256 tile
    (3, 64, 256, 28, 28, 1),      # 128 x 128 tiles, 2352 columns = 18.4 tiles
    (5, 256, 64, 14, 14, 1),
    (4, 128, 512, 7, 7, 1),       # 7x7 planes: runs of 49, columns cross images inside a 32-lane group
    (7, 32, 100, 5, 3, 1),        # Cout = 100: partial m tile, K = 32
    (1, 3, 8, 9, 11, 1),          # K tail (3), Cout 8, one image
    (2, 20, 36, 6, 6, 1),         # K tail (20 = 16 + 4)
    (3, 33, 132, 8, 8, 1),        # K tail (33), partial second m tile
    (2, 256, 512, 56, 56, 2),     # the downsample branch
    (3, 16, 64, 9, 7, 2),         # odd plane, stride 2
    (2, 16, 128, 10, 10, 3),      # stride 3
    (9, 512, 16, 1, 1, 1),        # 1x1 planes: 9 columns
    (1, 2048, 512, 7, 7, 1),      # K = 2048
]


@pytest.fixture(scope="module")
def nat():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from common.quantity import _native
    _native.lib()
    return _native


def _ref64(x, w, bias, s):
    y = torch.nn.functional.conv2d(x.double(), w.double(), None if bias is None else bias.double(), stride=s)
    return y


def _case(shape, seed, integer):
    N, cin, cout, H, W, s = shape
    g = torch.Generator(device="cuda").manual_seed(seed)
    if integer:
        x = torch.randint(-8, 9, (N, cin, H, W), device="cuda", generator=g).float()
        w = torch.randint(-8, 9, (cout, cin, 1, 1), device="cuda", generator=g).float()
        b = torch.randint(-100, 101, (cout,), device="cuda", generator=g).float()
    else:
        x = torch.randn(N, cin, H, W, device="cuda", generator=g)
        w = torch.randn(cout, cin, 1, 1, device="cuda", generator=g) * cin ** -0.5
        b = torch.randn(cout, device="cuda", generator=g)
    return x, w, b, w.view(cout, cin).t().contiguous(), s


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_exact_on_integer_valued_data(nat, shape):
    x, w, b, wt, s = _case(shape, 11, integer=True)       # |sum| <= 2048 * 64 + 100 < 2^24: fp32 is exact in any order
    y = nat.conv1x1_f32(x, wt, b, s)
    assert torch.equal(y.double(), _ref64(x, w, b, s))
    assert torch.equal(nat.conv1x1_f32(x, wt, None, s).double(), _ref64(x, w, None, s))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_gaussian_data_statistics_and_relu(nat, shape):
    x, w, b, wt, s = _case(shape, 12, integer=False)
    ref = _ref64(x, w, b, s)
    bound = _ref64(x.abs(), w.abs(), b.abs(), s)
    y = nat.conv1x1_f32(x, wt, b, s)
    assert bool(((y.double() - ref).abs() <= 1e-5 * bound).all())
    assert torch.equal(y, nat.conv1x1_f32(x, wt, b, s))                      # same bits from run to run
    # pass 1: the abs-max folded into an existing maximum, the ReLU copy
    mx = torch.tensor([0.0, 1e9, 0.0], device="cuda")
    r = torch.empty_like(y)
    y1 = nat.conv1x1_f32(x, wt, b, s, max_dev=mx, row=2, relu_out=r)
    assert torch.equal(y1, y) and torch.equal(r, torch.relu(y))
    assert mx.tolist() == [0.0, 1e9, float(y.abs().max())]
    nat.conv1x1_f32(x, wt, b, s, max_dev=mx, row=1)
    assert float(mx[1]) == 1e9                                               # a larger running maximum stays
    # pass 2: the histogram, accumulated onto existing counts, against the streaming kernel on the same tensor
    iv = torch.tensor([1.0, float(y.abs().max()) / 2048 + 1e-12], device="cuda")
    hist = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
    hist[1, 5] = 7
    want = hist.clone()
    y2 = nat.conv1x1_f32(x, wt, b, s, interval_dev=iv, hist_dev=hist, row=1, relu_out=r)
    nat.hist2048_seg([y], [1], iv, want)
    assert torch.equal(y2, y) and torch.equal(r, torch.relu(y)) and torch.equal(hist, want)
    assert int(hist[1].sum()) - 7 == int((y != 0).sum()) and int(hist[0].sum()) == 0


def test_against_the_oracles_exact_integer_convolution(nat, oracle):
    """The CPU oracle's integer convolution (oracle/fq_oracle.c, the checker of the int8 path) is exact; on integer-valued
    fp32 data so are these kernels: equal results, element for element."""
    rng = np.random.default_rng(5)
    xi = rng.integers(-8, 9, (3, 48, 10, 9)).astype(np.int32)
    wi = rng.integers(-8, 9, (72, 48, 1, 1)).astype(np.int32)
    wt = torch.from_numpy(wi.reshape(72, 48).T.astype(np.float32).copy()).cuda()
    for s in (1, 2):
        y = nat.conv1x1_f32(torch.from_numpy(xi.astype(np.float32)).cuda(), wt, None, s).cpu().numpy()
        assert np.array_equal(y, oracle.conv2d_int(xi, wi, stride=(s, s)).astype(np.float32))
    xs = rng.integers(-8, 9, (2, 3, 30, 23)).astype(np.int32)
    ws = rng.integers(-8, 9, (64, 3, 7, 7)).astype(np.int32)
    ys = nat.conv_stem_f32(torch.from_numpy(xs.astype(np.float32)).cuda(),
                           nat.pack_stem_weight(torch.from_numpy(ws.astype(np.float32)).cuda()), None, 64, (7, 7), 2, 3).cpu().numpy()
    assert np.array_equal(ys, oracle.conv2d_int(xs, ws, stride=(2, 2), pad=(3, 3)).astype(np.float32))


def test_histogram_with_an_interval_outside_the_fast_quotient_range(nat):
    x, w, b, wt, s = _case((2, 16, 64, 6, 6, 1), 13, integer=False)
    y = nat.conv1x1_f32(x, wt, b, s)
    for ivv in (1e-30, 3e25):                                               # IEEE divide path; everything in the last / first bin
        iv = torch.tensor([ivv], device="cuda")
        hist = torch.zeros(1, 2048, dtype=torch.int64, device="cuda")
        want = torch.zeros(1, 2048, dtype=torch.int64, device="cuda")
        nat.conv1x1_f32(x, wt, b, s, interval_dev=iv, hist_dev=hist, row=0)
        nat.hist2048_seg([y], [0], iv, want)
        assert torch.equal(hist, want)


def test_nan_and_zero_outputs(nat):
    x, w, b, wt, s = _case((2, 16, 64, 6, 6, 1), 14, integer=True)
    b.zero_()
    x[0, :, 2, 3] = 0.0                                                      # an exactly-zero output column: not counted
    x[1, 5, 1, 1] = float("nan")
    r = torch.empty(2, 64, 6, 6, device="cuda")
    mx = torch.zeros(1, device="cuda")
    y = nat.conv1x1_f32(x, wt, b, s, max_dev=mx, row=0, relu_out=r)
    assert bool((y[0, :, 2, 3] == 0).all()) and bool(torch.isnan(y[1, :, 1, 1]).all()) and bool(torch.isnan(r[1, :, 1, 1]).all())
    finite = y[~torch.isnan(y)]
    assert float(mx[0]) == float(finite.abs().max())                        # NaN does not enter the maximum (np.max would; the
    iv = torch.tensor([float(mx[0]) / 2048 + 1e-12], device="cuda")          # reference never sees NaN activations)
    hist = torch.zeros(1, 2048, dtype=torch.int64, device="cuda")
    want = torch.zeros(1, 2048, dtype=torch.int64, device="cuda")
    nat.conv1x1_f32(x, wt, b, s, interval_dev=iv, hist_dev=hist, row=0)
    nat.hist2048_seg([y], [0], iv, want)
    assert torch.equal(hist, want)


def test_streaming_form_beyond_the_infinity_cache(nat):
    shape = (48, 64, 256, 56, 56, 1)                                         # 154 MB out + 154 MB ReLU copy > 256 MB: non-temporal stores
    x, w, b, wt, s = _case(shape, 15, integer=True)
    r = torch.empty(48, 256, 56, 56, device="cuda")
    mx = torch.zeros(1, device="cuda")
    y = nat.conv1x1_f32(x, wt, b, s, max_dev=mx, row=0, relu_out=r)
    ref = torch.nn.functional.conv2d(x, w, b)                                # exact in fp32 as well on this data
    assert torch.equal(y, ref) and torch.equal(r, torch.relu(ref)) and float(mx[0]) == float(ref.abs().max())


def test_tail_split_workspace_is_the_callers(nat):
    """include/fq.h: the float convolutions allocate nothing.  The tail split's scratch is a `workspace` argument
    (fq_conv_f32_workspace_bytes), zero-filled once by the caller; the kernels leave its counters at zero.  NULL, a workspace
    that is too small or FQ_CONV_TAIL_SPLIT=0 run the launch UNSPLIT: on integer-valued data (every partial sum exact) the
    split and the unsplit launch must agree bit for bit, and the unsplit launch is one fma chain per output whatever the
    batch size -- the same images in two halves give the same bits, which the split form does not promise."""
    L = nat.lib()
    shape = (130, 256, 1024, 8, 16, 1)                                       # 1 040 tiles: the last 16 are cut into 4 K slices
    N, cin, cout, H, W, s = shape
    nbytes = int(L.fq_conv_f32_workspace_bytes())
    assert nbytes == 256 * 128 * 128 * 4 + 256 * 4
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")

    def run(x, wt, b, wsp, wsb, n=N):
        y = torch.empty(n, cout, H, W, device="cuda")
        rc = L.fq_conv1x1_f32(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), None, n, cin, H, W, cout, s, None, None, None,
                              wsp, wsb, None)
        assert rc == 0
        torch.cuda.synchronize()
        return y

    x, w, b, wt, _ = _case(shape, 41, integer=True)
    ref = torch.nn.functional.conv2d(x, w, b)
    y_split = run(x, wt, b, ws.data_ptr(), nbytes)
    assert torch.equal(y_split, ref)
    assert int(ws[256 * 128 * 128 * 4:].view(torch.int32).abs().sum()) == 0           # counters back at zero ...
    assert int(ws[:256 * 128 * 128 * 4].view(torch.int32).ne(0).sum()) > 0            # ... and the slices did meet here
    assert torch.equal(run(x, wt, b, ws.data_ptr(), nbytes), ref)                      # the same workspace, launch after launch
    assert torch.equal(run(x, wt, b, None, 0), ref)                                    # no workspace: unsplit
    small = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    assert torch.equal(run(x, wt, b, small.data_ptr(), 4096), ref) and int(small.sum()) == 0   # too small: unsplit, untouched

    xg, wg, bg, wtg, _ = _case(shape, 42, integer=False)
    y_un = run(xg, wtg, bg, None, 0)
    halves = torch.cat([run(xg[:65].contiguous(), wtg, bg, None, 0, 65), run(xg[65:].contiguous(), wtg, bg, None, 0, 65)])
    assert torch.equal(y_un, halves)                                                   # one chain per output, whatever N
    y_sp = run(xg, wtg, bg, ws.data_ptr(), nbytes)
    tol = 1e-5 * (torch.nn.functional.conv2d(xg.abs(), wg.abs()) + bg.abs().view(1, -1, 1, 1))
    assert bool(((y_sp - y_un).abs() <= tol).all())
    diff = (y_sp != y_un).flatten(1).any(1)                                            # only images inside the 16 split tiles differ
    assert int(diff.sum()) <= 17 and not bool(diff[:100].any())
    # the binding: one workspace per (device, stream) from torch's allocator, none when the split is switched off
    p1, n1 = nat.conv_workspace(x)
    assert n1 == nbytes and nat.conv_workspace(x) == (p1, n1)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        p2, _ = nat.conv_workspace(x)
    assert p2 != p1
    old = nat.conv_tail_split
    nat.conv_tail_split = False
    try:
        assert nat.conv_workspace(x) == (None, 0)
        assert torch.equal(nat.conv1x1_f32(xg, wtg, bg, s), y_un)
    finally:
        nat.conv_tail_split = old
    assert torch.equal(nat.conv1x1_f32(xg, wtg, bg, s), y_sp)


def test_argument_errors(nat):
    L = nat.lib()
    x = torch.zeros(1, 4, 2, 2, device="cuda")
    wt = torch.zeros(4, 6, device="cuda")                                    # Cout = 6: not a multiple of 4
    y = torch.zeros(1, 6, 2, 2, device="cuda")
    one = torch.zeros(1, device="cuda")
    h = torch.zeros(2048, dtype=torch.int64, device="cuda")
    call = lambda *a: L.fq_conv1x1_f32(*a)
    assert call(x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), None, 1, 4, 2, 2, 6, 1, None, None, None, None, 0, None) == -4
    wt8 = torch.zeros(4, 8, device="cuda")
    assert call(x.data_ptr(), wt8.data_ptr(), None, y.data_ptr(), None, 1, 4, 2, 2, 8, 1, one.data_ptr(), one.data_ptr(),
                h.data_ptr(), None, 0, None) == -1                           # both statistics at once
    assert call(x.data_ptr(), wt8.data_ptr(), None, y.data_ptr(), None, 1, 4, 2, 2, 8, 1, None, None, h.data_ptr(), None, 0, None) == -1
    assert call(x.data_ptr(), wt8.data_ptr(), None, y.data_ptr(), None, 1, 4, 2, 2, 8, 0, None, None, None, None, 0, None) == -1
    assert call(None, None, None, None, None, 0, 4, 2, 2, 8, 1, None, None, None, None, 0, None) == 0   # no images: nothing to do


def _bottleneck_net():
    """conv3x3 -> [1x1 -> ReLU -> 3x3 -> ReLU -> 1x1] + 1x1 stride-2 downsample -> Eltwise -> ReLU -> pool -> fc."""
    from torch import nn
    from common.quantity import Eltwise, View

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = nn.Conv2d(3, 32, 3, padding=1)
            self.relu0 = nn.ReLU()
            self.c1 = nn.Conv2d(32, 16, 1)
            self.relu1 = nn.ReLU()
            self.c2 = nn.Conv2d(16, 16, 3, stride=2, padding=1)
            self.relu2 = nn.ReLU()
            self.c3 = nn.Conv2d(16, 64, 1)
            self.down = nn.Conv2d(32, 64, 1, stride=2)
            self.add = Eltwise()
            self.relu3 = nn.ReLU()
            self.pool = nn.AvgPool2d(8)
            self.view = View()
            self.fc = nn.Linear(64, 10)

        def forward(self, x):
            x = self.relu0(self.stem(x))
            y = self.c3(self.relu2(self.c2(self.relu1(self.c1(x)))))
            x = self.relu3(self.add(y, self.down(x)))
            return self.fc(self.view(self.pool(x)))
    return Net()


def test_calibration_tables_with_and_without_the_kernel():
    from tools import Quantity
    tables, launches = [], []
    for own in (True, False):
        with product_workdir(input_shape="1,3,16,16", device="gpu", max_cali_img_num=3) as tmp:
            q = Quantity(cases.seed_model(_bottleneck_net(), base_seed=5).eval().cuda())
            q.own_conv1x1 = own
            bits = q.activation_quantize(cases.calib_batches(5, (8, 3, 16, 16), seed=77))
            tables.append((dict(bits), open(tmp + "/test/workdir/feat.table").read()))
            launches.append(q.timings["own_conv1x1_launches"])
    assert tables[0] == tables[1]
    assert launches[0] > 0 and launches[1] == 0          # three 1x1 convolutions from the third batch on / none


def test_resnet50_calibration_is_reproducible_bit_for_bit_and_leaves_the_convolution_library_alone(monkeypatch):
    """Every convolution of the fabu ResNet-50 runs on this library's kernels (stem, 1x1, 3x3), so two calibrations of the
    same batches give the same histograms bit for bit (the convolution library's Winograd kernels are not reproducible from
    call to call), and torch's convolution is never called inside a calibration -- the first one of the process included."""
    from common.quantity import merge_bn
    from model.resnet.ResNet_fabu import ResNet50
    from tools import Quantity
    model = merge_bn(cases.seed_model(ResNet50(input_size=64)).eval()).cuda()
    batches = cases.calib_batches(4, (4, 3, 64, 64), seed=55)
    hists, calls = [], []
    real = torch.nn.functional.conv2d

    def counting(*a, **k):
        calls[-1] += 1
        return real(*a, **k)
    for _run in range(2):
        calls.append(0)
        with product_workdir(input_shape="1,3,64,64", device="gpu", max_cali_img_num=3):
            q = Quantity(model)                                                  # (graph discovery runs torch's forward once)
            monkeypatch.setattr(torch.nn.functional, "conv2d", counting)         # Conv2d._conv_forward goes through F.conv2d
            q.activation_quantize(batches)
            monkeypatch.undo()
            hists.append((q._collector.hist_device.clone(), dict(q._collector.max_vals)))
            assert q.timings["own_conv1x1_launches"] > 0
    assert torch.equal(hists[0][0], hists[1][0]) and hists[0][1] == hists[1][1]
    # no library convolution at all -- not even in the first run, whose once-per-module checks compare with GEMMs (1x1 layers)
    # and im2col + GEMM (3x3 layers, the stem)
    assert calls == [0, 0]


def test_a_module_that_disagrees_keeps_the_library_convolution(monkeypatch):
    from tools import Quantity, pytorch_quantizer
    from common.quantity import _float_conv
    monkeypatch.setattr(_float_conv, "TOL", -1.0)                            # nothing can pass
    with product_workdir(input_shape="1,3,16,16", device="gpu", max_cali_img_num=3) as tmp:
        q = Quantity(cases.seed_model(_bottleneck_net(), base_seed=5).eval().cuda())
        bits = dict(q.activation_quantize(cases.calib_batches(5, (8, 3, 16, 16), seed=77)))
        assert q.timings["own_conv1x1_launches"] == 0
        assert all(_float_conv.is_off(m) for m in (q.model.c1, q.model.c3, q.model.down))
    monkeypatch.undo()
    with product_workdir(input_shape="1,3,16,16", device="gpu", max_cali_img_num=3) as tmp:
        q2 = Quantity(cases.seed_model(_bottleneck_net(), base_seed=5).eval().cuda())
        assert dict(q2.activation_quantize(cases.calib_batches(5, (8, 3, 16, 16), seed=77))) == bits
        import pickle
        # what Reconstruction's torch.save(self.model) would write: nothing of this package rides on the modules
        assert not any(k.startswith("_fq") for m in q2.model.modules() for k in m.__dict__)
        for m in (q2.model.c1, q2.model.c3, q2.model.down):
            assert b"_fq_" not in pickle.dumps(m) and "forward" not in m.__dict__


# ---------------------------------------------------------------------------------------------------------------------
# fq_conv_kxk_f32: R x S taps with zero padding through the same kernel (ResNet's 3x3 layers)
KXK_SHAPES = [  # N, Cin, Cout, H, W, R, S, stride, pad
    (11, 64, 256, 28, 28, 3, 3, 1, 1),   # tail split: 272 narrow tiles, the last 16 cut into 4 K slices (a slice starts inside a tap)
    (256, 512, 512, 7, 7, 3, 3, 1, 1),   # ResNet-50 layer4 conv2 at the bench's batch: 784 tiles, 16 x 16 slices of 18 K steps
    (2, 64, 64, 56, 56, 3, 3, 1, 1),
    (3, 128, 128, 28, 28, 3, 3, 2, 1),
    (4, 256, 256, 14, 14, 3, 3, 1, 1),
    (5, 512, 512, 7, 7, 3, 3, 1, 1),
    (2, 16, 36, 9, 11, 3, 3, 1, 1),      # odd planes, Cout 36 (partial m tile)
    (3, 32, 64, 10, 7, 3, 3, 2, 1),      # stride 2, odd width
    (2, 16, 64, 8, 8, 5, 5, 1, 2),       # 5x5 taps
    (2, 48, 20, 6, 9, 3, 3, 1, 0),       # no padding: Hout = 4, Wout = 7
    (1, 16, 128, 12, 12, 3, 3, 3, 1),    # stride 3
    (2, 16, 16, 5, 5, 1, 3, 1, 1),       # 1x3 taps with padding on both axes
    (7, 32, 32, 1, 1, 3, 3, 1, 1),       # 1x1 planes: eight of the nine taps are padding
]


def _kxk_case(shape, seed, integer):
    N, cin, cout, H, W, R, S, st, pad = shape
    g = torch.Generator(device="cuda").manual_seed(seed)
    if integer:
        x = torch.randint(-8, 9, (N, cin, H, W), device="cuda", generator=g).float()
        w = torch.randint(-8, 9, (cout, cin, R, S), device="cuda", generator=g).float()
        b = torch.randint(-100, 101, (cout,), device="cuda", generator=g).float()
    else:
        x = torch.randn(N, cin, H, W, device="cuda", generator=g)
        w = torch.randn(cout, cin, R, S, device="cuda", generator=g) * (cin * R * S) ** -0.5
        b = torch.randn(cout, device="cuda", generator=g)
    return x, w, b, (R, S), st, pad


@pytest.mark.parametrize("shape", KXK_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_kxk_exact_on_integer_valued_data(nat, shape):
    x, w, b, k, st, pad = _kxk_case(shape, 41, integer=True)       # |sum| <= 4608 * 64 + 100 < 2^24: exact in any order
    y = nat.conv_kxk_f32(x, nat.pack_kxk_weight(w), b, k, st, pad)
    assert torch.equal(y.double(), torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=st, padding=pad))
    y0 = nat.conv_kxk_f32(x, nat.pack_kxk_weight(w), None, k, st, pad)
    assert torch.equal(y0.double(), torch.nn.functional.conv2d(x.double(), w.double(), None, stride=st, padding=pad))


@pytest.mark.parametrize("shape", KXK_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_kxk_gaussian_data_statistics_and_relu(nat, shape):
    x, w, b, k, st, pad = _kxk_case(shape, 42, integer=False)
    wt = nat.pack_kxk_weight(w)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=st, padding=pad)
    bound = torch.nn.functional.conv2d(x.abs().double(), w.abs().double(), b.abs().double(), stride=st, padding=pad)
    y = nat.conv_kxk_f32(x, wt, b, k, st, pad)
    assert bool(((y.double() - ref).abs() <= 1e-5 * bound).all()) and torch.equal(y, nat.conv_kxk_f32(x, wt, b, k, st, pad))
    mx = torch.tensor([0.0, 1e9, 0.0], device="cuda")
    r = torch.empty_like(y)
    y1 = nat.conv_kxk_f32(x, wt, b, k, st, pad, max_dev=mx, row=2, relu_out=r)
    assert torch.equal(y1, y) and torch.equal(r, torch.relu(y)) and mx.tolist() == [0.0, 1e9, float(y.abs().max())]
    iv = torch.tensor([1.0, float(y.abs().max()) / 2048 + 1e-12], device="cuda")
    hist = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
    hist[1, 3] = 11
    want = hist.clone()
    y2 = nat.conv_kxk_f32(x, wt, b, k, st, pad, interval_dev=iv, hist_dev=hist, row=1, relu_out=r)
    nat.hist2048_seg([y], [1], iv, want)
    assert torch.equal(y2, y) and torch.equal(r, torch.relu(y)) and torch.equal(hist, want)


def test_kxk_argument_errors_and_oracle(nat, oracle):
    L = nat.lib()
    x = torch.zeros(1, 8, 4, 4, device="cuda")                               # Cin = 8: not a multiple of 16
    wt = torch.zeros(72, 8, device="cuda")
    y = torch.zeros(1, 8, 4, 4, device="cuda")
    assert L.fq_conv_kxk_f32(x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), None, 1, 8, 4, 4, 8, 3, 3, 1, 1, None, None, None, None, 0, None) == -4
    assert L.fq_conv_kxk_f32(x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), None, 1, 16, 2, 2, 8, 5, 5, 1, 1, None, None, None, None, 0, None) == -1
    rng = np.random.default_rng(6)
    xi = rng.integers(-8, 9, (2, 32, 9, 8)).astype(np.int32)
    wi = rng.integers(-8, 9, (40, 32, 3, 3)).astype(np.int32)
    for st in (1, 2):
        got = nat.conv_kxk_f32(torch.from_numpy(xi.astype(np.float32)).cuda(), nat.pack_kxk_weight(torch.from_numpy(wi.astype(np.float32)).cuda()),
                               None, (3, 3), st, 1).cpu().numpy()
        assert np.array_equal(got, oracle.conv2d_int(xi, wi, stride=(st, st), pad=(1, 1)).astype(np.float32))


# ---------------------------------------------------------------------------------------------------------------------
# fq_conv_stem_f32: the 7x7 stride-2 stem on the same matrix cores, same epilogue contract
STEM_SHAPES = [  # N, H, W, Cout, pad
    (2, 224, 224, 64, 3),         # ResNet-50's conv1
    (3, 64, 64, 64, 3),
    (2, 37, 53, 64, 3),           # odd planes: partial tiles both ways (Hout 19, Wout 27)
    (1, 16, 16, 32, 3),           # Cout 32: the upper MFMA tile is never stored
    (2, 31, 31, 48, 0),           # no padding
    (5, 7, 7, 64, 3),             # one tap row of real data per tile; Hout = Wout = 4
    (2, 40, 24, 20, 1),           # Cout 20, pad 1
]


def _stem_case(shape, seed, integer):
    N, H, W, cout, pad = shape
    g = torch.Generator(device="cuda").manual_seed(seed)
    if integer:
        x = torch.randint(-8, 9, (N, 3, H, W), device="cuda", generator=g).float()
        w = torch.randint(-8, 9, (cout, 3, 7, 7), device="cuda", generator=g).float()
        b = torch.randint(-100, 101, (cout,), device="cuda", generator=g).float()
    else:
        x = torch.randn(N, 3, H, W, device="cuda", generator=g)
        w = torch.randn(cout, 3, 7, 7, device="cuda", generator=g) * 147 ** -0.5
        b = torch.randn(cout, device="cuda", generator=g)
    return x, w, b, pad


def _stem(nat, x, w, b, pad, **kw):
    return nat.conv_stem_f32(x, nat.pack_stem_weight(w), b, w.shape[0], (7, 7), 2, pad, **kw)


@pytest.mark.parametrize("shape", STEM_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_stem_exact_on_integer_valued_data(nat, shape):
    x, w, b, pad = _stem_case(shape, 21, integer=True)            # |sum| <= 147 * 64 + 100: exact in any order
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=2, padding=pad)
    assert torch.equal(_stem(nat, x, w, b, pad).double(), ref)
    ref0 = torch.nn.functional.conv2d(x.double(), w.double(), None, stride=2, padding=pad)
    assert torch.equal(_stem(nat, x, w, None, pad).double(), ref0)


@pytest.mark.parametrize("shape", STEM_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_stem_gaussian_data_statistics_and_relu(nat, shape):
    x, w, b, pad = _stem_case(shape, 22, integer=False)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=2, padding=pad)
    bound = torch.nn.functional.conv2d(x.abs().double(), w.abs().double(), b.abs().double(), stride=2, padding=pad)
    y = _stem(nat, x, w, b, pad)
    assert bool(((y.double() - ref).abs() <= 1e-5 * bound).all())
    assert torch.equal(y, _stem(nat, x, w, b, pad))
    mx = torch.tensor([0.0, 1e9, 0.0], device="cuda")
    r = torch.empty_like(y)
    y1 = _stem(nat, x, w, b, pad, max_dev=mx, row=2, relu_out=r)
    assert torch.equal(y1, y) and torch.equal(r, torch.relu(y)) and mx.tolist() == [0.0, 1e9, float(y.abs().max())]
    iv = torch.tensor([1.0, float(y.abs().max()) / 2048 + 1e-12], device="cuda")
    hist = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
    hist[1, 9] = 3
    want = hist.clone()
    y2 = _stem(nat, x, w, b, pad, interval_dev=iv, hist_dev=hist, row=1, relu_out=r)
    nat.hist2048_seg([y], [1], iv, want)
    assert torch.equal(y2, y) and torch.equal(r, torch.relu(y)) and torch.equal(hist, want)


def test_stem_inf_next_to_a_window_does_not_leak_through_the_padded_tap(nat):
    """The stem pads its 7 taps per row to 8 (zero weights); the patch word behind the 7th tap is a real pixel.  An Inf /
    NaN there must not turn 0 * Inf into NaN in outputs whose own 7x7 window is finite (torch's convolution leaves them
    finite): every output must be NaN exactly where the reference convolution's is."""
    x, w, b, pad = _stem_case((2, 40, 44, 64, 3), 23, integer=True)
    x[0, 1, 10, 17] = float("inf")
    x[1, 2, 25, 30] = float("nan")
    x[1, 0, 0, 43] = float("-inf")
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=2, padding=pad)
    y = _stem(nat, x, w, b, pad).double()
    assert torch.equal(torch.isnan(y), torch.isnan(ref))
    fin = torch.isfinite(ref)
    assert torch.equal(y[fin], ref[fin]) and torch.equal(torch.isinf(y), torch.isinf(ref))


def test_stem_unsupported_shapes_are_refused(nat):
    L = nat.lib()
    assert L.fq_conv_stem_f32_packed_rows(3, 7, 7) == 168
    assert L.fq_conv_stem_f32_packed_rows(3, 3, 3) == -4 and L.fq_conv_stem_f32_packed_rows(1, 7, 7) == -4
    x = torch.zeros(1, 3, 16, 16, device="cuda")
    wp = torch.zeros(168, 64, device="cuda")
    y = torch.zeros(1, 64, 8, 8, device="cuda")
    call = lambda cin, cout, r, s, stride: L.fq_conv_stem_f32(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, 1, cin, 16, 16,
                                                              cout, r, s, stride, 3, None, None, None, None)
    assert call(3, 64, 7, 7, 2) == 0
    assert call(3, 64, 7, 7, 1) == -4 and call(3, 128, 7, 7, 2) == -4 and call(3, 64, 5, 5, 2) == -4 and call(4, 64, 7, 7, 2) == -4


# ---------------------------------------------------------------------------------------------------------------------
# fq_maxpool2d_f32 / fq_avgpool_global_f32: the float forward's pooling layers, torch's bits
@pytest.mark.parametrize("shape,k,s,p", [((3, 64, 112, 112), 3, 2, 1), ((2, 3, 8, 16), 3, 2, 1), ((1, 2, 10, 24), 3, 2, 1), ((2, 5, 17, 23), 3, 2, 1), ((2, 4, 9, 9), 2, 2, 0),
                                         ((1, 3, 8, 11), 3, 1, 1), ((2, 2, 7, 7), (3, 2), (2, 1), (1, 0)), ((1, 1, 3, 3), 3, 3, 1)])
def test_maxpool_equals_torch(nat, shape, k, s, p):
    pair = lambda v: (v, v) if isinstance(v, int) else v
    g = torch.Generator(device="cuda").manual_seed(31)
    x = torch.randn(*shape, device="cuda", generator=g)
    x.view(-1)[::97] = float("-inf")
    want = torch.nn.functional.max_pool2d(x, k, s, p)
    assert torch.equal(nat.maxpool2d_f32(x, pair(k), pair(s), pair(p)), want)
    x.view(-1)[5::131] = float("nan")                                # NaN propagates the way torch's kernel does it
    got, want = nat.maxpool2d_f32(x, pair(k), pair(s), pair(p)), torch.nn.functional.max_pool2d(x, k, s, p)
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(got.nan_to_num(0.0), want.nan_to_num(0.0))


@pytest.mark.parametrize("shape", [(256, 2048, 7, 7), (3, 5, 1, 1), (2, 300, 12, 12), (1, 1, 4, 6), (7, 513, 3, 5)])
def test_global_avgpool_equals_torch(nat, shape):
    g = torch.Generator(device="cuda").manual_seed(32)
    x = torch.randn(*shape, device="cuda", generator=g) * 3 + 0.5
    assert torch.equal(nat.avgpool_global_f32(x), torch.nn.functional.avg_pool2d(x, shape[2:]))


def test_testconv_runs_its_1x1_layers_on_the_own_kernel_and_hooks_still_fire():
    """ReconTest's TestConv (float convolution -> QuanDequan, new_quantity_op.py:283-292): a 1x1 layer goes through
    fq_conv1x1_f32 (checked once against torch), forward hooks on the inner nn.Conv2d still see its raw output, the module
    keeps no patched forward (whole models are pickled), and the result is QuanDequan of that raw output, bit for bit."""
    import pickle
    from torch import nn
    from common.quantity import TestConv, _float_conv, _native
    conv = nn.Conv2d(32, 64, 1).cuda().eval()
    with product_workdir(device="gpu") as tmp:
        layer = TestConv("c", conv, {"weight_bit": 6, "bias_bit": 5, "input_bit": 4, "output_bit": 4}, tmp + "/test/workdir/rt.pth")
        seen = []
        handle = layer.Conv.register_forward_hook(lambda m, i, o: seen.append(o.clone()))
        x = torch.randn(4, 32, 14, 14, device="cuda")
        with torch.no_grad():
            out = layer(x)
            handle.remove()
            assert _float_conv.is_verified(layer.Conv) and "forward" not in layer.Conv.__dict__ and len(seen) == 1
            assert not any(k.startswith("_fq") for k in layer.Conv.__dict__)
            assert torch.equal(out, _native.quandequan(seen[0], 4))
            ref = torch.nn.functional.conv2d(x, layer.Conv.weight, layer.Conv.bias)
            assert float((seen[0] - ref).abs().max()) <= 1e-4
        pickle.dumps(layer.Conv.state_dict())
        assert b"_fq_" not in pickle.dumps(layer)                   # the whole-module pickle carries no packed copy / flag


@pytest.mark.parametrize("bit,bitwidth", [(4, 8), (-1, 8), (9, 16)])
def test_quandequan_epilogue_equals_the_two_pass_form(nat, bit, bitwidth):
    """fq_conv1x1_qd_f32 / fq_conv_kxk_qd_f32 / fq_conv_stem_qd_f32 (TestConv.forward in one kernel) == fq_quandequan_f32 of
    the plain kernel's output, bit for bit, on Gaussian data with saturating, tie and zero cases; ragged shapes (K tail,
    partial tiles, stride 2, Cout not a multiple of the tile)."""
    g = torch.Generator(device="cuda").manual_seed(31 + bit)
    for (N, Cin, H, W, Cout, s) in ((3, 40, 13, 9, 36, 1), (2, 64, 28, 28, 256, 2), (5, 256, 7, 7, 64, 1)):
        x = torch.randn(N, Cin, H, W, device="cuda", generator=g) * 3
        x[0, 0, 0, :3] = torch.tensor([1e6, -1e6, 0.0], device="cuda")
        wt = torch.randn(Cin, Cout, device="cuda", generator=g) * Cin ** -0.5
        b = torch.randn(Cout, device="cuda", generator=g)
        want = nat.quandequan(nat.conv1x1_f32(x, wt, b, s), bit, bitwidth)
        assert torch.equal(nat.conv1x1_f32(x, wt, b, s, qd=(bit, bitwidth)), want)
        assert torch.equal(nat.conv1x1_f32(x, wt, None, s, qd=(bit, bitwidth)), nat.quandequan(nat.conv1x1_f32(x, wt, None, s), bit, bitwidth))
    for (N, Cin, H, W, Cout, k, s, p) in ((2, 32, 14, 11, 48, 3, 1, 1), (3, 64, 15, 15, 128, 3, 2, 1), (1, 16, 9, 9, 20, 5, 1, 2)):
        x = torch.randn(N, Cin, H, W, device="cuda", generator=g) * 2
        w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * (Cin * k * k) ** -0.5
        b = torch.randn(Cout, device="cuda", generator=g)
        wt = nat.pack_kxk_weight(w)
        want = nat.quandequan(nat.conv_kxk_f32(x, wt, b, (k, k), s, p), bit, bitwidth)
        assert torch.equal(nat.conv_kxk_f32(x, wt, b, (k, k), s, p, qd=(bit, bitwidth)), want)
    for shape in STEM_SHAPES[:3]:
        x, w, b, pad = _stem_case(shape, 33, integer=False)
        want = nat.quandequan(_stem(nat, x, w, b, pad), bit, bitwidth)
        assert torch.equal(_stem(nat, x, w, b, pad, qd=(bit, bitwidth)), want)
    with pytest.raises(nat.FqError):
        nat.conv1x1_f32(x[:, :, :4, :4].contiguous(), torch.zeros(3, 4, device="cuda"), None, 1, qd=4, relu_out=torch.zeros(1, device="cuda"))


@pytest.mark.parametrize("shape", [(3, 64, 13, 9, 256, 1), (2, 32, 28, 28, 128, 1), (5, 256, 7, 7, 1024, 1), (2, 64, 15, 15, 128, 2),
                                   (40, 64, 56, 56, 256, 1), (9, 256, 28, 28, 640, 1), (130, 256, 8, 16, 1024, 1)],
                         ids=lambda s: "x".join(map(str, s)))
def test_conv_add_relu_in_one_kernel_equals_the_two_kernels(nat, shape):
    """fq_conv1x1_add_f32 (conv3 + Eltwise + ReLU of a residual block, pass 1) leaves bit for bit what fq_conv1x1_f32 (max
    form) followed by fq_add_absmax_f32 leave -- both maxima, the ReLU output, and each of the two intermediate tensors
    exactly when it is asked for (the others' destinations stay untouched); NaN / Inf / -0.0 included; both tile shapes,
    partial column tiles, stride 2, the streaming (non-temporal) form."""
    N, Cin, H, W, Cout, s = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g) * 2
    wt = torch.randn(Cin, Cout, device="cuda", generator=g) * Cin ** -0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    ho, wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(N, Cout, ho, wo, device="cuda", generator=g) * 3
    res[0, 0, 0, :4] = torch.tensor([float("nan"), float("inf"), -0.0, -1e30], device="cuda")
    x[-1, :, -1, -1] = 0.0
    m2 = torch.zeros(4, device="cuda")
    v = nat.conv1x1_f32(x, wt, b, s, max_dev=m2, row=1)
    r2 = torch.empty_like(v)
    sm = nat.add_absmax(v, res, m2, 3, relu_out=r2)
    for keep_y in (False, True):
        for keep_s in (False, True):
            m1 = torch.zeros(4, device="cuda")
            mark = 12345.0
            y = torch.full_like(v, mark) if keep_y else None
            so = torch.full_like(v, mark) if keep_s else None
            r1 = torch.full_like(v, mark)
            out = nat.conv1x1_add_f32(x, wt, b, s, res, m1, 1, 3, r1, out=y, sum_out=so)
            assert out is r1
            assert torch.equal(m1, m2), (m1, m2)
            assert torch.equal(r1.view(torch.int32), r2.view(torch.int32))          # (bit patterns: NaN == NaN, -0.0 != 0.0)
            if keep_y:
                assert torch.equal(y.view(torch.int32), v.view(torch.int32))
            if keep_s:
                assert torch.equal(so.view(torch.int32), sm.view(torch.int32))
    # maxima fold into what the rows already hold
    m1 = torch.tensor([0.0, 1e9, 0.0, 0.5], device="cuda")
    nat.conv1x1_add_f32(x, wt, b, s, res, m1, 1, 3, torch.empty_like(v))
    assert float(m1[1]) == 1e9 and float(m1[3]) == float(m2[3]) and float(m1[0]) == 0.0 and float(m1[2]) == 0.0


@pytest.mark.parametrize("shape", [(3, 64, 13, 9, 256, 1), (2, 32, 28, 28, 128, 1), (5, 256, 7, 7, 1024, 1), (2, 64, 15, 15, 128, 2),
                                   (40, 64, 56, 56, 256, 1), (9, 256, 28, 28, 640, 1), (130, 256, 8, 16, 1024, 1)],
                         ids=lambda s: "x".join(map(str, s)))
def test_conv_add_relu_histogram_form_equals_the_two_kernels(nat, shape):
    """fq_conv1x1_add_hist_f32 (pass 2) leaves the two histogram rows and the ReLU output that fq_conv1x1_f32 (histogram form)
    followed by fq_add_hist_f32 leave, bit for bit -- on top of counts the rows already hold; intervals inside and outside the
    fast-quotient range; values beyond the last bin, exact zeros, NaN / Inf."""
    N, Cin, H, W, Cout, s = shape
    g = torch.Generator(device="cuda").manual_seed(7 + sum(shape))
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g) * 2
    x[0, :, 0, 0] = 0.0
    wt = torch.randn(Cin, Cout, device="cuda", generator=g) * Cin ** -0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    b[0] = 0.0                                                     # (exact zeros in the convolution's output)
    ho, wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(N, Cout, ho, wo, device="cuda", generator=g) * 3
    res[0, 0, 0, :4] = torch.tensor([float("nan"), float("inf"), -0.0, -1e30], device="cuda")
    for ivs in ((0.004, 0.0061), (3e-39, 0.0061), (0.004, 1e-3)):
        iv = torch.tensor([1.0, ivs[0], 1.0, ivs[1]], device="cuda")
        h2 = torch.arange(4 * 2048, device="cuda", dtype=torch.int64).view(4, 2048) % 7
        h1 = h2.clone()
        v = nat.conv1x1_f32(x, wt, b, s, interval_dev=iv, hist_dev=h2, row=1)
        r2 = torch.empty_like(v)
        nat.add_hist(v, res, iv, h2, 3, relu_out=r2)
        r1 = torch.full_like(v, 777.0)
        assert nat.conv1x1_add_hist_f32(x, wt, b, s, res, iv, h1, 1, 3, r1) is r1
        assert torch.equal(h1, h2)
        assert torch.equal(r1.view(torch.int32), r2.view(torch.int32))


def test_relu_only_form_writes_the_relu_and_not_the_convolution_output(nat):
    """y = NULL with relu_out (out=False): the ReLU copy and the statistic (abs-max or histogram) are those of the ordinary call,
    for the 1x1, the R x S and the stem kernel; without relu_out a missing y is an argument error."""
    g = torch.Generator(device="cuda").manual_seed(404)
    iv = torch.tensor([0.003, 1.0], device="cuda")
    cases_ = []
    x = torch.randn(5, 64, 28, 28, device="cuda", generator=g)
    wt = torch.randn(64, 128, device="cuda", generator=g) * 0.125
    b = torch.randn(128, device="cuda", generator=g)
    cases_.append(lambda **kw: nat.conv1x1_f32(x, wt, b, 1, **kw))
    w3 = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    b3 = torch.randn(64, device="cuda", generator=g)
    wk = nat.pack_kxk_weight(w3)
    cases_.append(lambda **kw: nat.conv_kxk_f32(x, wk, b3, (3, 3), 1, 1, **kw))
    xs, ws, bs, pad = _stem_case(STEM_SHAPES[0], 35, integer=False)
    cases_.append(lambda **kw: _stem(nat, xs, ws, bs, pad, **kw))
    for run in cases_:
        for stat in ("max", "hist"):
            m2, m1 = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
            h2 = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
            h1 = h2.clone()
            kw2 = dict(max_dev=m2, row=0) if stat == "max" else dict(interval_dev=iv, hist_dev=h2, row=0)
            kw1 = dict(max_dev=m1, row=0) if stat == "max" else dict(interval_dev=iv, hist_dev=h1, row=0)
            y = run(**kw2)
            r2 = torch.empty_like(y)
            run(relu_out=r2, **kw2)
            m2.zero_(), h2.zero_()
            run(relu_out=r2, **kw2)
            r1 = torch.full_like(y, 5.5)
            assert run(relu_out=r1, out=False, **kw1) is None
            assert torch.equal(r1, r2) and torch.equal(r1, torch.relu(y))
            assert torch.equal(m1, m2) and torch.equal(h1, h2)
    with pytest.raises((nat.FqError, AssertionError)):
        nat.conv1x1_f32(x, wt, b, 1, out=False)


def test_conv_add_argument_errors(nat):
    x = torch.zeros(1, 16, 4, 4, device="cuda")
    r = torch.zeros(1, 128, 4, 4, device="cuda")
    m = torch.zeros(2, device="cuda")
    with pytest.raises(nat.FqError):            # Cout not a multiple of 128
        nat.conv1x1_add_f32(x, torch.zeros(16, 64, device="cuda"), torch.zeros(64, device="cuda"), 1, r[:, :64].contiguous(), m, 0, 1,
                            torch.zeros(1, 64, 4, 4, device="cuda"))
    with pytest.raises(nat.FqError):            # Cin not a multiple of 16
        nat.conv1x1_add_f32(x[:, :8].contiguous(), torch.zeros(8, 128, device="cuda"), torch.zeros(128, device="cuda"), 1, r, m, 0, 1,
                            torch.empty_like(r))
    assert not nat.conv1x1_add_f32_supported(8, 128) and not nat.conv1x1_add_f32_supported(16, 64)
    assert nat.conv1x1_add_f32_supported(64, 256)


def test_testconv_runs_as_one_kernel_unless_its_convolution_is_hooked():
    """TestConv.forward: no forward hook on the inner nn.Conv2d -> one kernel (QuanDequan in the convolution's epilogue), the
    standalone fq_quandequan_f32 pass is not launched; with a hook on the inner module the two-pass form runs (the hook sees
    the un-quantised output, as the reference's would) -- and both give the same bits."""
    from torch import nn
    from common.quantity import TestConv, _float_conv, _native
    convs = [nn.Conv2d(32, 64, 1), nn.Conv2d(16, 32, 3, padding=1), nn.Conv2d(3, 64, 7, stride=2, padding=3)]
    with product_workdir(device="gpu") as tmp:
        for i, conv in enumerate(convs):
            conv = conv.cuda().eval()
            layer = TestConv("c%d" % i, conv, {"weight_bit": 6, "bias_bit": 5, "input_bit": 4, "output_bit": 4},
                             tmp + "/test/workdir/rt%d.pth" % i)
            x = torch.randn(4, conv.in_channels, 30, 26, device="cuda")
            calls, real = {"n": 0}, _native.quandequan

            def counted(*a, **k):
                calls["n"] += 1
                return real(*a, **k)
            _native.quandequan = counted
            try:
                with torch.no_grad():
                    fused = layer(x)
                    assert calls["n"] == 0 and _float_conv.is_verified(layer.Conv)
                    seen = []
                    h = layer.Conv.register_forward_hook(lambda m, inp, o: seen.append(o.clone()))
                    two_pass = layer(x)
                    h.remove()
                    assert calls["n"] == 1 and len(seen) == 1
            finally:
                _native.quandequan = real
            assert torch.equal(fused, two_pass) and torch.equal(fused, _native.quandequan(seen[0], 4))
            assert not any(k.startswith("_fq") for k in layer.Conv.__dict__)


def test_testlinear_runs_as_one_kernel_unless_its_linear_layer_is_hooked():
    """TestLinear.forward (reference new_quantity_op.py:248-256): one kernel -- the linear layer as the 1x1 convolution of a 1 x 1
    plane with QuanDequan in its epilogue -- when nobody hooks the inner nn.Linear or output_qdp; with a hook the reference's two
    calls run.  The fused result is QuanDequan of the own kernel's plain output bit for bit, and within one quantisation step of
    the two-pass form (torch's GEMM sums in another order)."""
    from torch import nn
    from common.quantity import TestLinear, _float_conv, _native
    with product_workdir(device="gpu") as tmp:
        for i, (cin, cout, n) in enumerate(((2048, 1000, 256), (64, 12, 3), (500, 10, 1), (33, 7, 5))):
            lin = nn.Linear(cin, cout).cuda().eval()
            layer = TestLinear("fc%d" % i, lin, {"weight_bit": 6, "bias_bit": 5, "input_bit": 4, "output_bit": 4},
                               tmp + "/test/workdir/rl%d.pth" % i)
            x = torch.randn(n, cin, device="cuda")
            calls, real = {"n": 0}, _native.quandequan

            def counted(*a, **k):
                calls["n"] += 1
                return real(*a, **k)
            _native.quandequan = counted
            try:
                with torch.no_grad():
                    fused = layer(x)
                    takes = cout % 4 == 0
                    assert calls["n"] == (0 if takes else 1) and _float_conv.is_verified(layer.linear) == takes
                    seen = []
                    h = layer.linear.register_forward_hook(lambda m, inp, o: seen.append(o.clone()))
                    two_pass = layer(x)
                    h.remove()
                    assert len(seen) == 1 and calls["n"] == (1 if takes else 2)
                    h = layer.output_qdp.register_forward_hook(lambda m, inp, o: seen.append(o.clone()))
                    assert torch.equal(layer(x), two_pass) and len(seen) == 2       # a hook on output_qdp fires
                    h.remove()
            finally:
                _native.quandequan = real
            assert fused.shape == (n, cout) and torch.equal(two_pass, _native.quandequan(seen[0], 4))
            if takes:
                wt = layer.linear.weight.detach().t().contiguous()
                plain = _native.conv1x1_f32(x.view(n, cin, 1, 1), wt, layer.linear.bias, 1).view(n, cout)
                assert torch.equal(fused, _native.quandequan(plain, 4))
                assert float((fused - two_pass).abs().max()) <= 2.0 ** -4
            else:
                assert torch.equal(fused, two_pass)
            assert "wt_linear" not in layer.linear.__dict__ and "forward" not in layer.linear.__dict__


def test_pool_modules_are_served_and_tables_do_not_change():
    from torch import nn
    from common.quantity import View
    from tools import Quantity, pytorch_quantizer

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(3, 16, 3, padding=1)
            self.relu = nn.ReLU()
            self.pool = nn.MaxPool2d(3, 2, 1)
            self.conv2 = nn.Conv2d(16, 32, 1)
            self.relu2 = nn.ReLU()
            self.avg = nn.AvgPool2d(8)
            self.view = View()
            self.fc = nn.Linear(32, 10)

        def forward(self, x):
            x = self.pool(self.relu(self.conv(x)))
            return self.fc(self.view(self.avg(self.relu2(self.conv2(x)))))
    tables = []
    for own in (True, False):
        with product_workdir(input_shape="1,3,16,16", device="gpu", max_cali_img_num=3) as tmp:
            q = Quantity(cases.seed_model(Net(), base_seed=9).eval().cuda())
            q.own_pools = own
            bits = dict(q.activation_quantize(cases.calib_batches(5, (8, 3, 16, 16), seed=78)))
            tables.append((bits, open(tmp + "/test/workdir/feat.table").read()))
            served = [pytorch_quantizer._flag(m, pytorch_quantizer._POOL_VERIFIED) for m in (q.model.pool, q.model.avg)]
            assert served == [own, own] and "forward" not in q.model.pool.__dict__      # patched only while calibrating
    assert tables[0] == tables[1]


# ---------------------------------------------------------------------------------------------------------------------
# The split-bf16 form of the 1x1 kernels (fq_conv1x1_sb_f32 and its add / QuanDequan forms; opt-in, FQ_CONV_SPLIT_BF16=1): every fp32
# operand as three bf16 values on the bf16 matrix cores, six of the nine products.  Same contracts as the fp32-MFMA entry points.
SB_SHAPES = [s for s in SHAPES if s[1] % 16 == 0 and s[2] % 4 == 0]


@pytest.mark.parametrize("shape", SB_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_split_bf16_exact_on_integer_valued_data_and_as_accurate_as_the_fp32_chain(nat, shape):
    x, w, b, wt, s = _case(shape, 21, integer=True)       # integers below 2^8 have no mid and lo part: exact, as on the fp32 MFMA
    wsb = nat.pack_sb_weight(w)
    assert torch.equal(nat.conv1x1_f32(x, wsb, b, s).double(), _ref64(x, w, b, s))
    assert torch.equal(nat.conv1x1_f32(x, wsb, None, s).double(), _ref64(x, w, None, s))
    x, w, b, wt, s = _case(shape, 22, integer=False)
    wsb = nat.pack_sb_weight(w)
    ref, bound = _ref64(x, w, b, s), _ref64(x.abs(), w.abs(), b.abs(), s)
    y, yf = nat.conv1x1_f32(x, wsb, b, s), nat.conv1x1_f32(x, wt, b, s)
    err, err_f = float(((y.double() - ref).abs() / bound).max()), float(((yf.double() - ref).abs() / bound).max())
    assert err <= 1e-5 and err <= 1.5 * err_f + 1e-7, (err, err_f)            # the product's bound, and the fp32 chain's class
    assert torch.equal(y, nat.conv1x1_f32(x, wsb, b, s))                      # same bits from run to run
    # the statistic forms, the ReLU copy, the QuanDequan form: on the SAME output
    mx = torch.tensor([0.0, 1e9, 0.0], device="cuda")
    r = torch.empty_like(y)
    assert torch.equal(nat.conv1x1_f32(x, wsb, b, s, max_dev=mx, row=2, relu_out=r), y) and torch.equal(r, torch.relu(y))
    assert mx.tolist() == [0.0, 1e9, float(y.abs().max())]
    iv = torch.tensor([1.0, float(y.abs().max()) / 2048 + 1e-12], device="cuda")
    hist = torch.zeros(2, 2048, dtype=torch.int64, device="cuda")
    hist[1, 5] = 7
    want = hist.clone()
    assert torch.equal(nat.conv1x1_f32(x, wsb, b, s, interval_dev=iv, hist_dev=hist, row=1, relu_out=r), y)
    nat.hist2048_seg([y], [1], iv, want)
    assert torch.equal(hist, want) and torch.equal(r, torch.relu(y))
    assert torch.equal(nat.conv1x1_f32(x, wsb, b, s, qd=(4, 8)), nat.quandequan(y.clone(), 4, 8))
    r2 = torch.full_like(y, 3.0)
    mx2 = torch.zeros(1, device="cuda")
    assert nat.conv1x1_f32(x, wsb, b, s, max_dev=mx2, row=0, relu_out=r2, out=False) is None
    assert torch.equal(r2, torch.relu(y)) and float(mx2[0]) == float(y.abs().max())


@pytest.mark.parametrize("shape", [(3, 64, 13, 9, 256, 1), (5, 256, 7, 7, 1024, 1), (2, 64, 15, 15, 128, 2), (9, 256, 28, 28, 640, 1),
                                   (130, 256, 8, 16, 1024, 1)], ids=lambda s: "x".join(map(str, s)))
def test_split_bf16_conv_add_forms_equal_their_two_kernels(nat, shape):
    """fq_conv1x1_sb_add_f32 / _add_hist_f32 leave bit for bit what fq_conv1x1_sb_f32 followed by the Eltwise kernels leave (the
    contract of the fp32 pair), tail split included (the last two shapes)."""
    N, Cin, H, W, Cout, s = shape
    g = torch.Generator(device="cuda").manual_seed(3 + sum(shape))
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g) * 2
    w = torch.randn(Cout, Cin, device="cuda", generator=g) * Cin ** -0.5
    wsb = nat.pack_sb_weight(w)
    b = torch.randn(Cout, device="cuda", generator=g)
    ho, wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(N, Cout, ho, wo, device="cuda", generator=g) * 3
    m2 = torch.zeros(4, device="cuda")
    v = nat.conv1x1_f32(x, wsb, b, s, max_dev=m2, row=1)
    r2 = torch.empty_like(v)
    sm = nat.add_absmax(v, res, m2, 3, relu_out=r2)
    m1 = torch.zeros(4, device="cuda")
    y, so, r1 = torch.empty_like(v), torch.empty_like(v), torch.empty_like(v)
    nat.conv1x1_add_f32(x, wsb, b, s, res, m1, 1, 3, r1, out=y, sum_out=so)
    assert torch.equal(m1, m2) and torch.equal(y, v) and torch.equal(so, sm) and torch.equal(r1, r2)
    iv = torch.tensor([1.0, 0.004, 1.0, 0.0061], device="cuda")
    h2 = torch.zeros(4, 2048, dtype=torch.int64, device="cuda")
    h1 = h2.clone()
    v = nat.conv1x1_f32(x, wsb, b, s, interval_dev=iv, hist_dev=h2, row=1)
    nat.add_hist(v, res, iv, h2, 3, relu_out=r2)
    nat.conv1x1_add_hist_f32(x, wsb, b, s, res, iv, h1, 1, 3, r1)
    assert torch.equal(h1, h2) and torch.equal(r1, r2)


def test_split_bf16_pack_and_argument_errors(nat):
    L = nat.lib()
    g = torch.Generator(device="cuda").manual_seed(31)
    w = torch.randn(8, 16, device="cuda", generator=g)
    raw = nat.pack_sb_weight(w)
    # the pack: [Cin / 16][plane][k % 16 / 8][Cout][8] bf16 -> [plane][Cout][Cin]
    p = raw.view(torch.bfloat16).reshape(1, 3, 2, 8, 8).permute(1, 3, 0, 2, 4).reshape(3, 8, 16)
    assert torch.equal(p[0].float() + p[1].float() + p[2].float(), w)           # hi + mid + lo IS the weight
    assert torch.equal(p[0], w.bfloat16())
    p = raw
    assert L.fq_conv1x1_sb_supported(16, 8) == 1 and L.fq_conv1x1_sb_supported(24, 8) == 0 and L.fq_conv1x1_sb_supported(16, 6) == 0
    x = torch.zeros(1, 24, 4, 4, device="cuda")
    y = torch.zeros(1, 8, 4, 4, device="cuda")
    assert L.fq_conv1x1_sb_f32(x.data_ptr(), p.data_ptr(), None, y.data_ptr(), None, 1, 24, 4, 4, 8, 1, None, None, None, None, 0, None) == -4
    assert L.fq_conv1x1_sb_f32(x.data_ptr(), None, None, y.data_ptr(), None, 1, 16, 4, 4, 8, 1, None, None, None, None, 0, None) == -1
    assert L.fq_conv1x1_sb_pack(None, p.data_ptr(), 16, 8, None) == -1


def test_resnet50_tables_with_the_split_bf16_kernels(monkeypatch):
    """The product with FQ_CONV_SPLIT_BF16=1: the fabu ResNet-50's feat.table is the one the fp32-MFMA kernels give."""
    from common.quantity import merge_bn, _float_conv
    from model.resnet.ResNet_fabu import ResNet50
    from tools import Quantity
    model = merge_bn(cases.seed_model(ResNet50(input_size=64)).eval()).cuda()
    batches = cases.calib_batches(4, (4, 3, 64, 64), seed=58)
    tables = []
    for env in ("1", "0"):
        monkeypatch.setenv("FQ_CONV_SPLIT_BF16", env)
        _float_conv.forget()
        with product_workdir(input_shape="1,3,64,64", device="gpu", max_cali_img_num=3) as tmp:
            q = Quantity(model)
            bits = q.activation_quantize(batches)
            tables.append((dict(bits), open(tmp + "/test/workdir/feat.table").read()))
    monkeypatch.undo()
    _float_conv.forget()
    assert tables[0] == tables[1]


def test_every_kernel_a_module_runs_on_is_checked_once(monkeypatch):
    """The once-per-process check is keyed by the KERNEL (`_float_conv.kernel_key`), not by the module alone: a 1x1 layer first
    seen on the fp32-MFMA kernel is checked again the first time it runs on the split-bf16 kernel (FQ_CONV_SPLIT_BF16 flipped
    in the process, as bench.py does), a 3x3 layer first seen at a shape the Winograd kernel refuses is checked again when a
    shape it takes arrives -- and each only once."""
    from torch import nn
    from common.quantity import _float_conv
    checks = []
    real = _float_conv.verified

    def counting(m, run, x, k=None):
        before = set(_float_conv.state(m).get("verified", ()))
        out = real(m, run, x, k)
        if set(_float_conv.state(m).get("verified", ())) != before:
            checks.append(_float_conv.kernel_key(m, k))
        return out
    monkeypatch.setattr(_float_conv, "verified", counting)
    c1 = nn.Conv2d(128, 64, 1).cuda().eval()
    c3 = nn.Conv2d(16, 64, 3, padding=1).cuda().eval()
    x1 = torch.randn(2, 128, 6, 6, device="cuda")
    with torch.no_grad():
        monkeypatch.delenv("FQ_CONV_SPLIT_BF16", raising=False)
        a = _float_conv.call(c1, x1)
        _float_conv.call(c1, x1)
        assert checks == [("c1", False)] and _float_conv.is_verified(c1, "c1")
        monkeypatch.setenv("FQ_CONV_SPLIT_BF16", "1")
        assert not _float_conv.is_verified(c1, "c1") and _float_conv.is_verified(c1)
        b = _float_conv.call(c1, x1)
        _float_conv.call(c1, x1)
        assert checks == [("c1", False), ("c1", True)]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
        monkeypatch.delenv("FQ_CONV_SPLIT_BF16")
        _float_conv.call(c1, x1)
        assert len(checks) == 2                                                     # both kernels stay verified
        del checks[:]
        small, big = torch.randn(3, 16, 1, 3, device="cuda"), torch.randn(3, 16, 8, 8, device="cuda")
        assert _float_conv.kind(c3, small) == "kxk" and _float_conv.kind(c3, big) == "wino"
        for x in (small, big, small, big):
            y = _float_conv.call(c3, x)
            ref = torch.nn.functional.conv2d(x.double(), c3.weight.double(), c3.bias.double(), padding=1)
            assert float((y.double() - ref).abs().max()) < 1e-5
        assert checks == [("kxk", False), ("wino", False)]


def test_a_bias_at_an_odd_offset_keeps_a_3x3_layer_on_the_direct_kernel():
    """The Winograd epilogue reads the bias 16 bytes at a time and refuses a bias that is not 16-byte aligned
    (FQ_ERR_UNSUPPORTED); kind() sees that before the call, so a bias that is a view into a flat parameter buffer at an odd
    offset runs on fq_conv_kxk_f32 (scalar bias reads) instead of raising from inside the model's forward."""
    from torch import nn
    from common.quantity import _float_conv, _native
    conv = nn.Conv2d(16, 64, 3, padding=1).cuda().eval()
    flat = torch.randn(65, device="cuda")
    conv.bias = nn.Parameter(flat[1:])                          # 4 bytes past a 16-byte boundary
    assert conv.bias.data_ptr() % 16 == 4
    x = torch.randn(2, 16, 8, 8, device="cuda")
    with pytest.raises(_native.FqError):
        _native.conv_wino_f32(x, _native.pack_wino_weight(conv.weight), conv.bias, 64)
    with torch.no_grad():
        assert _float_conv.kind(conv, x) == "kxk"
        y = _float_conv.call(conv, x)
    ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    assert _float_conv.is_verified(conv, "kxk") and float((y.double() - ref).abs().max()) < 1e-5


def test_testlinear_keeps_the_two_calls_when_the_kernel_refuses_the_layer(monkeypatch):
    """call_linear_qd returns None -- the caller then runs the reference's two calls -- for anything the 1x1 kernel refuses,
    instead of raising FqError from inside TestLinear.forward: the guard knows the weight-matrix limit, and a refusal it does
    not know about (here: forced) turns the module off."""
    from torch import nn
    from common.quantity import _float_conv, _native
    lin = nn.Linear(64, 12).cuda().eval()
    x = torch.randn(3, 64, device="cuda")

    def refuse(*a, **k):
        raise _native.FqError("fq_conv1x1_f32: FQ_ERR_UNSUPPORTED")
    with torch.no_grad():
        monkeypatch.setattr(_native, "conv1x1_f32", refuse)
        assert _float_conv.call_linear_qd(lin, x, 4, 8) is None and _float_conv.is_off(lin)
        monkeypatch.undo()
        assert _float_conv.call_linear_qd(lin, x, 4, 8) is None                    # stays off
        assert _float_conv.call_linear_qd(nn.Linear(64, 12).cuda().eval(), x[:, :32].contiguous(), 4, 8) is None   # wrong width


def test_graph_discovery_on_the_gpu_runs_on_the_own_kernels_and_finds_the_reference_graph(golden_dir, monkeypatch):
    """Quantity(model) traces one forward for the graph (build_net_structure): on the GPU its convolutions run on this library's
    kernels (unchecked: only tensor identities are used), so a fresh process does not enter the convolution library there either;
    the graph is the reference's (golden G7), the same as with the library convolutions, no value fingerprint is computed (no
    host read of a device value), and nothing stays on the modules."""
    import json
    import os
    from common.quantity import merge_bn
    from model.resnet.ResNet_fabu import ResNet50
    from tools import Quantity, pytorch_quantizer
    with open(os.path.join(golden_dir, "g7_netinfo.json")) as fh:
        ref = json.load(fh)["r50"]
    model = merge_bn(cases.seed_model(ResNet50(), gamma_scale=0.5).eval()).cuda()
    real, calls, tids = torch.nn.functional.conv2d, [], []
    real_tid = pytorch_quantizer.tid

    def counting(*a, **k):
        calls[-1] += 1
        return real(*a, **k)
    for own in ("1", "0"):
        monkeypatch.setenv("FQ_OWN_CONV1X1", own)
        monkeypatch.setattr(torch.nn.functional, "conv2d", counting)
        monkeypatch.setattr(pytorch_quantizer, "tid", lambda t: (tids.append(1), real_tid(t))[1])
        calls.append(0)
        with product_workdir(input_shape="1,3,224,224", device="gpu"):
            q = Quantity(model)
        monkeypatch.undo()
        assert dict(q.net_info) == ref["net_info"] and list(q.net_info.keys()) == ref["net_info_order"]
        assert q.cared_op_layer_names == ref["cared_op_layer_names"] and q.layers_num == ref["layers_num"]
        assert not any("forward" in m.__dict__ or m._forward_hooks for m in model.modules())
    assert calls == [0, 53] and tids == []
