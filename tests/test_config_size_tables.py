"""The reference's own tables at the sizes BASELINE.json quotes (goldens G12 / G13, captured by tests/golden/make_golden_r50.py in
the two runs whose wall time BASELINE.md records -- `time18`, `time101`):

  G12  config 1 at its stated size: ResNet_18_fabu, 256 synthetic 3x32x32 images (2 batches of 128, MAX_CALI_IMG_NUM 1) through
       the imported reference's Quantity.activation_quantize / weight_quantize (pytorch_quantizer.py:345-489, :592-677):
       feat.table, weight.table, the sha256 of every JSON file, and what its KL search was handed (merged intervals, 2048-bin
       histograms) with the bits it found;
  G13  config 5's shape: fabu ResNet-101 @3x512x512, ONE image: the same statistics and feat.table, 139 rows.

CPU (runs in the build container): the drop-in orchestrator with the oracle-backed statistics doubles on torch-CPU forwards --
the same convolution library the reference ran on, so the statistics are compared EXACTLY (every interval, every one of the
30 x 2048 bins).  GPU (`-m gpu`): the HIP engine end to end; its float forward sums in a different order than the CPU's, so the
statistics carry the tolerance of tests/test_gpu_r50_tables.py::test_r50_end_to_end_feat_table_equals_the_reference (intervals
2e-6 relative, histograms 2e-3 of their element count in L1 distance) and the tables must be the reference's, every row."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import cases
from workdir_util import product_workdir


def _dir_state(d):
    return {f: hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest() for f in sorted(os.listdir(d))}


@pytest.fixture(scope="module")
def g12(golden_dir):
    with open(os.path.join(golden_dir, "g12_r18_config1.json")) as fh:
        return json.load(fh), np.load(os.path.join(golden_dir, "g12_r18_config1_stats.npz"))


def _spy_quantity(base_quantizer, **attrs):
    """A Quantity whose quantizer records what the KL search is handed (the product's counterpart of the capture's spy)."""
    from tools import Quantity
    seen = {}

    class SpyQuantizer(base_quantizer):
        def quantize(self, distributions, distribution_intervals):
            seen["names"] = list(distribution_intervals.keys())
            seen["interval"] = np.array([float(distribution_intervals[k]) for k in seen["names"]])
            if torch.is_tensor(distributions):
                seen["hist"] = distributions.cpu().numpy().copy()
            else:
                seen["hist"] = np.stack([np.asarray(distributions[k], dtype=np.int64) for k in seen["names"]])
            return super().quantize(distributions, distribution_intervals)

    return type("SpyQuantity", (Quantity,), dict(attrs, quantizer_cls=SpyQuantizer)), seen


def _r18():
    from common.quantity import merge_bn
    from model.resnet.ResNet_18_fabu import ResNet18
    return merge_bn(cases.seed_model(ResNet18()).eval())


def test_config1_at_256_images_on_the_cpu_doubles_equals_the_reference_exactly(oracle, g12):
    """BASELINE configs[0] verbatim ("ResNet-18 KL calibration on 256 images ..., CPU reference path"): host logic + oracle on the
    reference's own workload at its own size.  Same torch-CPU forward as the reference's run in this container, so every number
    the KL search sees is the reference's, bit for bit -- not only the table."""
    from engine_doubles import OracleCollector, OracleQuantizer
    tables, stats = g12
    recipe = tables["recipe"]
    Spy, seen = _spy_quantity(OracleQuantizer, collector_cls=OracleCollector)
    with product_workdir(input_shape="1,3,32,32", device="cpu", max_cali_img_num=recipe["max_cali_img_num"]) as tmp:
        q = Spy(_r18())
        q.activation_quantize(cases.calib_batches(recipe["n_batches"], tuple(recipe["shape"])))
        wd = os.path.join(tmp, "test", "workdir")
        feat = open(os.path.join(wd, "feat.table")).read()
        q.weight_quantize()
        weight_table = open(os.path.join(wd, "weight.table")).read()
        files = {d: _dir_state(os.path.join(wd, d)) for d in ("weight", "bias", "new_weight", "new_bias")}
    assert seen["names"] == list(stats["names"])
    assert int(stats["hist"][0].sum()) + 0 <= 256 * 3 * 32 * 32          # the image row: 256 images' worth of non-zero elements
    np.testing.assert_array_equal(seen["interval"], stats["interval"])
    np.testing.assert_array_equal(seen["hist"], stats["hist"])
    assert feat == tables["feat_table"] == str(stats["feat_table"])
    assert weight_table == tables["weight_table"]
    assert files == tables["files"]


@pytest.mark.gpu
def test_config1_at_256_images_on_the_gpu_writes_the_reference_tables(g12):
    """The same workload through the HIP engine (own fp32-MFMA forward, fq_absmax / fq_hist2048 / fq_kl_threshold)."""
    from common.quantity import Quantizer
    tables, stats = g12
    recipe = tables["recipe"]
    Spy, seen = _spy_quantity(Quantizer)
    with product_workdir(input_shape="1,3,32,32", device="gpu", max_cali_img_num=recipe["max_cali_img_num"]) as tmp:
        q = Spy(_r18().cuda())
        assert type(q).collector_cls.__module__ == "common.quantity.distribution_collector"       # the HIP engine
        q.activation_quantize(cases.calib_batches(recipe["n_batches"], tuple(recipe["shape"])))
        wd = os.path.join(tmp, "test", "workdir")
        feat = open(os.path.join(wd, "feat.table")).read()
        q.weight_quantize()
        weight_table = open(os.path.join(wd, "weight.table")).read()
        files = {d: _dir_state(os.path.join(wd, d)) for d in ("weight", "bias", "new_weight", "new_bias")}
    _close_statistics(seen, stats)
    got, want = feat.strip().split("\n"), tables["feat_table"].strip().split("\n")
    assert len(got) == len(want) == 30
    assert [a for a, b in zip(got, want) if a != b] == []
    assert weight_table == tables["weight_table"] and files == tables["files"]     # weights never pass through a convolution


def _close_statistics(seen, stats, rel_interval=2e-6, l1_hist=2e-3):
    assert seen["names"] == list(stats["names"])
    ref_i, ref_h = stats["interval"], stats["hist"]
    assert np.max(np.abs(seen["interval"] - ref_i) / ref_i) <= rel_interval, float(np.max(np.abs(seen["interval"] - ref_i) / ref_i))
    l1 = np.abs(seen["hist"] - ref_h).sum(axis=1) / np.maximum(ref_h.sum(axis=1), 1)
    assert l1.max() <= l1_hist, (float(l1.max()), seen["names"][int(l1.argmax())])
    # the image row never passes through a convolution: exact
    np.testing.assert_array_equal(seen["hist"][0], ref_h[0])
    assert seen["interval"][0] == ref_i[0]


@pytest.mark.gpu
def test_config5_shape_r101_at_512_on_the_gpu_writes_the_reference_table(golden_dir):
    """G13: the reference's calibration of ResNet-101 @3x512x512 on one image (65 s of its Python path; here one forward pair):
    139 rows, 132 M cared elements.  BatchNorm folded with the factors the reference's merge_bn computed in the build container
    (cases.fold_bn_with_scales: torch.sqrt on CPU is machine dependent in the last bit)."""
    from common.quantity import Quantizer
    from model.resnet.ResNet_fabu import ResNet101
    stats = np.load(os.path.join(golden_dir, "g13_r101_512_stats.npz"))
    scales = {k[len("scale__"):]: stats[k] for k in stats.files if k.startswith("scale__")}
    model = cases.fold_bn_with_scales(cases.seed_model(ResNet101(input_size=512), gamma_scale=0.5).eval(), scales)
    Spy, seen = _spy_quantity(Quantizer)
    with product_workdir(input_shape="1,3,512,512", device="gpu", max_cali_img_num=0) as tmp:
        q = Spy(model.cuda())
        q.activation_quantize(cases.calib_batches(1, (1, 3, 512, 512), seed=512))
        feat = open(os.path.join(tmp, "test", "workdir", "feat.table")).read()
    _close_statistics(seen, stats)
    got, want = feat.strip().split("\n"), str(stats["feat_table"]).strip().split("\n")
    assert len(got) == len(want) == 139
    assert [a for a, b in zip(got, want) if a != b] == []
