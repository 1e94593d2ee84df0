"""End-to-end ResNet-18 calibration through the drop-in orchestrator on the CPU, with the statistics
engine replaced by oracle-backed doubles (tests/engine_doubles.py): every table and JSON file must be
byte-identical to what the imported reference produced (golden G3, tests/golden/make_golden_e2e.py).
This pins the host logic (graph discovery, merge groups, bit tying, table / JSON writers, rewriter,
including the reference's second-rewrite quirk) and, transitively, the oracle on real activations."""
import hashlib
import json
import os

import numpy as np
import pytest

import cases
from engine_doubles import OracleCollector, OracleQuantizer
from workdir_util import product_workdir


def _sha(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def _dir_state(d):
    return {f: _sha(os.path.join(d, f)) for f in sorted(os.listdir(d))}


def _unbox(d):
    return {k: list(v.values())[0] for k, v in d.items()}


@pytest.fixture(scope="module")
def g3(golden_dir):
    with open(os.path.join(golden_dir, "g3_r18_e2e.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def run(oracle):
    from common.quantity import merge_bn
    from model.resnet.ResNet_18_fabu import ResNet18
    from tools import Quantity

    class CpuQuantity(Quantity):
        collector_cls = OracleCollector
        quantizer_cls = OracleQuantizer

    out = {}
    with product_workdir(input_shape="1,3,32,32", device="cpu", max_cali_img_num=1) as tmp:
        model = merge_bn(cases.seed_model(ResNet18()).eval())
        q = CpuQuantity(model)
        out["net_info"] = dict(q.net_info)
        out["net_info_order"] = list(q.net_info.keys())
        out["cared"] = q.cared_op_layer_names
        out["groups"] = q.get_merge_groups(q.net_info)
        out["layers_num"] = q.layers_num
        bits = q.activation_quantize(cases.calib_batches(3, (4, 3, 32, 32)))
        out["bits"] = {k: int(v) for k, v in bits.items()}
        out["max_vals"] = {k: float(v) for k, v in q._collector.max_vals.items()}
        out["intervals"] = {k: float(v) for k, v in q._collector._distribution_intervals.items()}
        # the golden was read from the reference collector AFTER the merge step, which writes the
        # pooled histogram back into the collector's own dict (aliasing): compare the pooled sums
        pooled = [g for g in out["groups"] if not q._group_has_eltwise(g)]
        out["hist_sums"] = {k: int(np.asarray(v).sum()) for k, v in q._collector.merged_distributions(pooled).items()}
        out["thr_val"] = {k: float(v) for k, v in q._quantizer.threshold_value.items()}
        wd = os.path.join(tmp, "test", "workdir")
        out["feat_table"] = open(os.path.join(wd, "feat.table")).read()
        q.weight_quantize()
        out["weight_table_1"] = open(os.path.join(wd, "weight.table")).read()
        out["files_1"] = {d: _dir_state(os.path.join(wd, d)) for d in ("weight", "bias", "new_weight", "new_bias")}
        out["verbatim"] = {k: open(os.path.join(wd, k)).read() for k in
                           ("bias/fc.bias.json", "new_bias/fc.bias.json", "new_bias/conv1.0.bias.json",
                            "weight/conv1.0.weight.json")}
        q.rewrite_weight()
        out["weight_table_2"] = open(os.path.join(wd, "weight.table")).read()
        out["files_2"] = {d: _dir_state(os.path.join(wd, d)) for d in ("weight", "bias", "new_weight", "new_bias")}
    return out


def test_graph_matches_reference(run, g3):
    assert run["net_info_order"] == g3["net_info_order"]
    assert run["net_info"] == g3["net_info"]
    assert run["cared"] == g3["cared_op_layer_names"]
    assert run["groups"] == g3["merge_groups"]
    assert run["layers_num"] == g3["layers_num"]


def test_statistics_match_reference(run, g3):
    assert run["max_vals"] == _unbox(g3["max_vals"])
    assert run["intervals"] == _unbox(g3["intervals_final"])
    assert run["hist_sums"] == g3["hist_sums"]
    assert run["thr_val"] == _unbox(g3["threshold_value"])
    assert run["bits"] == g3["bits_final"]


def test_feat_table_byte_identical(run, g3):
    assert run["feat_table"] == g3["feat_table"]


def test_weight_table_and_json_byte_identical(run, g3):
    assert run["weight_table_1"] == g3["weight_table_after_quantize"]
    assert run["files_1"] == g3["files_after_quantize"]
    for k, text in run["verbatim"].items():
        assert text == g3["verbatim"][k], k


def test_second_rewrite_quirk_reproduced(run, g3):
    """The reference script calls rewrite_weight() a second time, which re-reads bias/ at already
    aligned bits and overwrites new_bias/ with un-rescaled values."""
    assert run["weight_table_2"] == g3["weight_table_after_second_rewrite"]
    assert run["files_2"] == g3["files_after_second_rewrite"]
    assert run["files_2"]["new_bias"] != run["files_1"]["new_bias"]


def test_kl_weight_branch_matches_reference(golden_dir, g3, oracle):
    """weight_quantize() with _DKL_weight = True (reference pytorch_quantizer.py:644-648): histogram +
    KL sweep over the parameters themselves; weight.table (including its worker-ordered lines) and
    all JSON files must equal the reference's."""
    from common.quantity import merge_bn
    from model.resnet.ResNet_18_fabu import ResNet18
    from tools import Quantity

    class CpuQuantity(Quantity):
        collector_cls = OracleCollector
        quantizer_cls = OracleQuantizer

    with open(os.path.join(golden_dir, "g3b_r18_dkl_weights.json")) as fh:
        ref = json.load(fh)
    with product_workdir(input_shape="1,3,32,32", device="cpu", max_cali_img_num=1) as tmp:
        q = CpuQuantity(merge_bn(cases.seed_model(ResNet18()).eval()))
        wd = os.path.join(tmp, "test", "workdir")
        with open(os.path.join(wd, "feat.table"), "w") as fh:
            fh.write(g3["feat_table"])
        q._DKL_weight = True
        q.weight_quantize()
        assert open(os.path.join(wd, "weight.table")).read() == ref["weight_table"]
        assert {d: _dir_state(os.path.join(wd, d)) for d in ("weight", "bias", "new_weight", "new_bias")} == ref["files"]
