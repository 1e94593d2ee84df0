"""The fused calibration forward on random model topologies (scripts/model_fuzz.py) -- residual blocks with and without projection, a
convolution output with two consumers, a sum with two consumers, concatenations, pools in odd places, in-place ReLUs: with and
without the two fusions that rest on a proof about the model's dataflow (deferral, relu-only) the statistics are equal BIT FOR BIT --
a chain wrongly taken hands somebody a tensor nobody wrote -- and against the library convolutions they agree to the
summation-order bound.    pytest -m gpu"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_topologies_give_the_unfused_tables():
    spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
    mf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mf)
    found = []
    bad, seen = mf.run(48, 11, log=found.append)
    assert bad == 0, [m for m in found if not m.startswith("  (")]
    # ... and the fused paths really ran on these graphs
    assert seen["conv_add_chains_proven"] > 0 and seen["conv_add_launches"] > 0 and seen["conv_add_hist_launches"] > 0
    assert seen["relu_only_chains_proven"] > 20 and seen["launches_without_own_output"] > 100 and seen["own_conv1x1_launches"] > 500


def test_a_model_with_in_place_relus_calibrates_to_the_same_bits_every_time(monkeypatch):
    """When a later module overwrites hooked tensors in place, pass 2 takes its histograms from inside the hooks -- and its
    convolutions must still run on the own kernels: on the convolution library (whose kernels do not give the same bits from call
    to call) the histograms of such a model differed by a handful of elements from one calibration to the next, and pass 2 binned
    values that were not the ones pass 1 had taken the maxima of (found by scripts/model_fuzz.py, model 138 of seed 6)."""
    import random
    import torch
    spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
    mf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mf)
    i, seed = 138, 6
    rng = random.Random(seed * 100003 + i)
    size = rng.choice([16, 24, 32])
    torch.manual_seed(seed * 7919 + i)
    model = mf.Net(rng, size).eval().cuda()
    assert any(isinstance(m, torch.nn.ReLU) and m.inplace for m in model.modules())
    bs = rng.choice([4, 8])
    batches = [(torch.randn(bs, 3, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(3)]
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
