"""Randomised parity sweeps.  Convolution: the int8 MFMA convolution (every kernel variant the dispatcher picks, all output
forms, the fused residual add) against the CPU oracle: scripts/conv_fuzz.py with a fixed seed.   pytest -m gpu"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shapes_match_the_oracle():
    spec = importlib.util.spec_from_file_location("conv_fuzz", os.path.join(ROOT, "scripts", "conv_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    failures = mod.run(120, 101, verbose=False)
    assert not failures, failures[:5]
    failures = mod.run_stem(40, 102, verbose=False)          # fq_conv2d_i8_stem on random stem-shaped layers
    assert not failures, failures[:5]
    failures = mod.run_halo(50, 103, verbose=False)          # the resident-halo form of the 3x3 / stride 1 / padding 1 layers
    assert not failures, failures[:5]


@pytest.mark.parametrize("groups", ["", "1"])
def test_streaming_1x1_kernel_matches_the_oracle(groups):
    """csrc/fq_conv1x1_i8.hip (opt-in: FQ_CONV_STREAM=1, read once per process, hence the child process): every shape class it
    takes, int8 output and the fused NewAdd with int8 / int16 residuals, against the CPU oracle -- as many streams as the
    device holds, and with FQ_STREAM_GROUPS=1 (8 streams: several pixel tiles per workgroup on small inputs)."""
    import subprocess
    import sys
    env = dict(os.environ, FQ_CONV_STREAM="1")
    if groups:
        env["FQ_STREAM_GROUPS"] = groups
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "conv_fuzz.py"), "40", "303", "stream"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "40 cases, 0 mismatches" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("stages", ["", "3"])
def test_eight_wave_halo_kernel_matches_the_oracle(stages):
    """conv3x3_i8_halo8_kernel (256-pixel tiles on eight waves; by default only launches that fill the chip take it, FQ_HALO8=2
    forces it for every 3x3 / stride 1 / padding 1 layer; read once per process, hence the child process): every plane size
    from 1 x 1, tiles spanning images and lying past the end, 1-4 channel slices, fp32 / int8 / both outputs, ring of two and
    of three, against the CPU oracle."""
    import subprocess
    import sys
    env = dict(os.environ, FQ_HALO8="2")
    if stages:
        env["FQ_HALO_STAGES"] = stages
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "conv_fuzz.py"), "60", "404", "halo"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "60 cases, 0 mismatches" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_random_histograms_kl_sweep_matches_the_oracle_bit_for_bit():
    """scripts/kl_fuzz.py: 96 random histograms of eight families; thresholds and KL curves (same include/fq_log.h on
    both sides) must agree bit for bit."""
    spec = importlib.util.spec_from_file_location("kl_fuzz", os.path.join(ROOT, "scripts", "kl_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    failures = mod.run(96, 2024, verbose=False)
    assert not failures, failures[:5]


def test_random_segment_lists_match_the_oracle():
    """scripts/calib_fuzz.py: ragged / unaligned / multi-segment rows, zeros, denormals, bin-edge values, accumulation,
    and the per-channel kernels, all bit-exact against the oracle."""
    spec = importlib.util.spec_from_file_location("calib_fuzz", os.path.join(ROOT, "scripts", "calib_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    failures = mod.run(25, 77, verbose=False)
    assert not failures, failures[:5]


def test_random_float_convolutions_are_exact_on_integer_data():
    """scripts/float_conv_fuzz.py: fq_conv1x1_f32 / fq_conv_kxk_f32 / fq_conv_stem_f32 on 150 random shapes (K tails, partial
    tiles, strides 1-3, paddings 0-3, 1x1 .. 5x5 taps, 1-pixel planes) with integer-valued data against a float64
    convolution -- exact -- plus the folded abs-max / histogram / ReLU copy on the same output."""
    spec = importlib.util.spec_from_file_location("float_conv_fuzz", os.path.join(ROOT, "scripts", "float_conv_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    failures = mod.run(150, 909, verbose=False)
    assert not failures, failures[:5]
