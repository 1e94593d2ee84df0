"""The fused calibration forward on random model topologies (scripts/model_fuzz.py) -- residual blocks with and without projection, a
convolution output with two consumers, a sum with two consumers, concatenations, pools in odd places, in-place ReLUs: with and
without the two fusions that rest on a proof about the model's dataflow (deferral, relu-only) the statistics are equal BIT FOR BIT --
a chain wrongly taken hands somebody a tensor nobody wrote -- and against the library convolutions they agree to the
summation-order bound.    pytest -m gpu"""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class InPlaceRelus(torch.nn.Module):
    """The shape scripts/model_fuzz.py found (model 138 of seed 6 at the time): 3x3 stride-2 convolutions followed by in-place ReLUs,
    and one layer the own kernels do not take (3x3 on 8 channels)."""

    def __init__(self):
        super(InPlaceRelus, self).__init__()
        from common.quantity import Eltwise, View
        nn = torch.nn
        self.stem = nn.Conv2d(3, 64, 7, stride=2, padding=3); self.r0 = nn.ReLU()
        self.c1 = nn.Conv2d(64, 256, 3, padding=1); self.r1 = nn.ReLU()
        self.c2 = nn.Conv2d(256, 8, 1); self.r2 = nn.ReLU(inplace=True)
        self.c3 = nn.Conv2d(8, 8, 3, padding=1); self.r3 = nn.ReLU()
        self.c4 = nn.Conv2d(8, 256, 1); self.add = Eltwise(); self.r4 = nn.ReLU()
        self.c5 = nn.Conv2d(256, 128, 3, stride=2, padding=1); self.r5 = nn.ReLU(inplace=True)
        self.c6 = nn.Conv2d(128, 128, 1); self.r6 = nn.ReLU(inplace=True)
        self.c7 = nn.Conv2d(128, 128, 3, padding=1); self.add2 = Eltwise(); self.r7 = nn.ReLU(inplace=True)
        self.view = View(); self.fc = nn.Linear(128 * 8 * 8, 10)

    def forward(self, x):
        x = self.r1(self.c1(self.r0(self.stem(x))))
        y = self.c4(self.r3(self.c3(self.r2(self.c2(x)))))
        x = self.r4(self.add(x, y))
        x = self.r5(self.c5(x))
        z = self.r6(self.c6(x))
        x = self.r7(self.add2(self.c7(z), z))
        return self.fc(self.view(x))


class SharedSum(torch.nn.Module):
    def __init__(self):
        super(SharedSum, self).__init__()
        from common.quantity import Eltwise, Concat, View
        self.stem = torch.nn.Conv2d(3, 16, 3, padding=1); self.r0 = torch.nn.ReLU()
        self.c1 = torch.nn.Conv2d(16, 16, 1); self.add = Eltwise()
        self.raw = torch.nn.Conv2d(16, 8, 1)                    # reads the sum as the Eltwise wrote it
        self.r1 = torch.nn.ReLU(inplace=True)                   # ... which this ReLU then overwrites
        self.act = torch.nn.Conv2d(16, 8, 3, padding=1)         # reads the rectified sum
        self.cat = Concat(); self.r2 = torch.nn.ReLU()
        self.view = View(); self.fc = torch.nn.Linear(16 * 16 * 16, 10)

    def forward(self, x):
        x = self.r0(self.stem(x))
        s = self.add(self.c1(x), x)
        a = self.raw(s)
        b = self.act(self.r1(s))
        return self.fc(self.view(self.r2(self.cat(a, b))))


def test_random_topologies_give_the_unfused_tables():
    spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
    mf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mf)
    found = []
    # evidence size inside the suite (300 models, ~2 minutes; FQ_FUZZ_MODELS shrinks it for a quick local run): round 5's two real
    # dataflow bugs were found by exactly this fuzzer at a few hundred models, not at 48
    bad, seen = mf.run(int(os.environ.get("FQ_FUZZ_MODELS", "300")), 11, log=found.append)
    assert bad == 0, [m for m in found if not m.startswith("  (")]
    # ... and the fused paths really ran on these graphs
    assert seen["conv_add_chains_proven"] > 0 and seen["conv_add_launches"] > 0 and seen["conv_add_hist_launches"] > 0
    assert seen["relu_only_chains_proven"] > 20 and seen["launches_without_own_output"] > 100 and seen["own_conv1x1_launches"] > 500


def test_a_model_with_in_place_relus_calibrates_to_the_same_bits_every_time(monkeypatch):
    """When a later module overwrites hooked tensors in place, pass 2 takes its histograms from inside the hooks -- and its
    convolutions must still run on the own kernels: on the convolution library (whose kernels do not give the same bits from call
    to call) the histograms of such a model differed by a handful of elements from one calibration to the next, and pass 2 binned
    values that were not the ones pass 1 had taken the maxima of (found by scripts/model_fuzz.py)."""
    spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
    mf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mf)
    torch.manual_seed(5)
    model, size = InPlaceRelus().eval().cuda(), 32
    batches = [(torch.randn(8, 3, size, size, device="cuda"), torch.zeros(8, dtype=torch.long)) for _ in range(3)]
    real, calls = torch.nn.functional.conv2d, []

    def counting(x, *a, **k):
        calls.append(tuple(x.shape))
        return real(x, *a, **k)
    first = mf.calibrate(model, size, batches)                 # (the once-per-module checks of the process happen here)
    monkeypatch.setattr(torch.nn.functional, "conv2d", counting)
    runs = [mf.calibrate(model, size, batches) for _ in range(3)]
    monkeypatch.undo()
    assert first[4]["inplace_consumers"] is True
    for r in runs:
        assert torch.equal(r[2], first[2]) and r[1] == first[1] and r[3] == first[3]
    # the one layer the own kernels do not take (3x3 on 8 channels) is the only one that ever reaches the library, in either pass
    assert calls and all(shape[1] == 8 for shape in calls), calls


def test_resident_integer_plans_of_random_topologies_give_the_boundary_logits():
    """scripts/recon_fuzz.py: the same random graphs calibrated, rewritten and rebuilt as ReconModel -- the logits of the resident plan
    (integer hand-offs, fused ReLUs, conv + NewAdd, block tails with and without a projection, pools on integers), eager and as one
    HIP graph, on two inputs, equal the logits with fp32 module boundaries bit for bit."""
    import sys
    spec = importlib.util.spec_from_file_location("recon_fuzz", os.path.join(ROOT, "scripts", "recon_fuzz.py"))
    rf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rf)
    found = []
    bad, seen = rf.run(int(os.environ.get("FQ_FUZZ_MODELS", "300")), 3, log=found.append)
    assert bad == 0, found
    assert seen["resident_convs"] > 300 and seen["fused_conv_adds"] > 40 and seen["fused_relus"] > 200, seen
    assert seen["fused_block_tails"] > 5 and seen["fused_projections"] > 0 and seen["resident_pools"] > 5, seen


def test_an_in_place_relu_between_two_consumers_of_a_sum_invalidates_its_integer_form():
    """A NewAdd whose sum is read RAW by one convolution, then overwritten by an in-place nn.ReLU, then read by another: legal
    PyTorch (the first reader ran before the ReLU).  The producer serves both kinds of consumers with an fp32 tensor that carries
    its integer form; the in-place ReLU writes the fp32 values only, so the integers are stale from then on and the second
    convolution must not read them (resident.carry remembers the tensor's version counter).  Before that, resident.enable()
    refused such a model ("does not reproduce the fp32-boundary forward"; scripts/recon_fuzz.py, 14 of 900 random models)."""
    import sys
    spec = importlib.util.spec_from_file_location("recon_fuzz", os.path.join(ROOT, "scripts", "recon_fuzz.py"))
    rf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rf)
    from common.quantity import resident
    from workdir_util import product_workdir

    def make():
        torch.manual_seed(3)
        return SharedSum().eval().cuda()
    data = [(torch.randn(8, 3, 16, 16, device="cuda"), torch.zeros(8, dtype=torch.long)) for _ in range(2)]
    sys.modules.setdefault("test_gpu_model_fuzz", sys.modules[__name__])
    with product_workdir(input_shape="1,3,16,16", device="gpu", max_cali_img_num=1):
        net = rf.recon_of(make(), make(), data)
        x = data[0][0]
        with torch.no_grad():
            plain = net(x)
            summary = resident.enable(net, x)                 # (verifies the plan on x: raised before the fix)
            assert torch.equal(net(x), plain)
            assert torch.equal(net(torch.flip(x, dims=[0])), torch.flip(plain, dims=[0]))
        assert summary["resident_adds"] == 1 and summary["resident_convs"] >= 3, summary


def test_random_topologies_with_layers_the_library_keeps_and_the_per_channel_rows():
    """The same generator with depthwise and dilated convolutions and nearest-neighbour upsampling here and there (layers that stay
    on the convolution library in the middle of a fused forward; its deterministic mode makes 'bit for bit' meaningful), and the
    per-(tensor, channel) calibration of random models: equal bit for bit when repeated, and consistent with the per-tensor one (a
    tensor's maximum is its largest channel maximum, its histogram holds as many elements as its channels' histograms together)."""
    spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
    mf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mf)
    was = torch.backends.cudnn.deterministic
    try:
        found = []
        bad, _seen = mf.run(30, 12, log=found.append, odd=True)
        assert bad == 0, [m for m in found if not m.startswith("  (")]
        found = []
        assert mf.run_channels(20, 13, log=found.append) == 0, found
        assert mf.run_channels(12, 14, log=found.append, odd=True) == 0, found
        # ... and the activation cache: nothing kept / the deepest tensors of every batch (plan B: pass 2 re-runs a prefix and stops)
        # / whole batches (plan A) / everything -- the statistics of the calibration without a cache, bit for bit
        # (with a cache pass 1 leaves residual sums to pass 2's pair / chain kernels -- Quantity.pair_hist / pair_chain: 100 models)
        bad, plans = mf.run_cache(int(os.environ.get("FQ_FUZZ_CACHE_MODELS", "100")), 15, log=found.append)
        assert bad == 0, found
        assert plans.get(("B", True), 0) > 10 and plans.get(("A", True), 0) > 10 and plans.get("sums_left_to_pairs", 0) > 50, plans
    finally:
        torch.backends.cudnn.deterministic = was


def test_a_calibration_leaves_the_model_as_it_found_it_also_when_it_fails_half_way():
    """The fused forward works by instance-level `forward` attributes and hooks on the user's modules.  After a calibration -- and
    after one that dies in the middle (here: the loader raises on its third batch) -- nothing of that is left: no hook, no patched
    forward, the model computes what it computed before; and the same Quantity object calibrates again to the same table."""
    spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
    mf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mf)
    from tools import Quantity
    from workdir_util import product_workdir
    model, size, bs, _rng = mf.random_net(7, 3, False, "cuda")
    x = torch.randn(bs, 3, size, size, device="cuda")
    with torch.no_grad():
        before = model(x).clone()

    def same(y):                                             # (the library's convolutions need not give the same bits twice)
        return torch.allclose(y, before, rtol=1e-4, atol=1e-5)

    def clean():
        return not any(m._forward_hooks or m._forward_pre_hooks or "forward" in m.__dict__ for m in model.modules())

    class Dies(object):
        def __init__(self, n):
            self.items = [(torch.randn(bs, 3, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(n)]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            if i == 2:
                raise OSError("the loader lost a file")
            return self.items[i]

        def __iter__(self):
            return (self[i] for i in range(len(self)))
    with product_workdir(input_shape="1,3,%d,%d" % (size, size), device="gpu", max_cali_img_num=3) as tmp:
        q = Quantity(model)
        assert clean()
        with pytest.raises(OSError):
            q.activation_quantize(Dies(4))
        assert clean()
        with torch.no_grad():
            assert same(model(x))
        good = [(torch.randn(bs, 3, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(4)]
        q.activation_quantize(good)
        first = open(os.path.join(tmp, "test", "workdir", "feat.table")).read()
        assert clean()
        q.activation_quantize(good)                          # the same object once more
        assert open(os.path.join(tmp, "test", "workdir", "feat.table")).read() == first and clean()
        with torch.no_grad():
            assert same(model(x))
        by_channel = q.activation_quantize_per_channel(good)
        assert clean() and len(by_channel["image"]) == 3
        with torch.no_grad():
            assert same(model(x))
