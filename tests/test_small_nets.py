"""Small end-to-end fixtures captured from the imported reference (golden G9): the LeNet fixture, a
net with a Concat merge group, a VGG-like stack and a depthwise-separable net.  CPU: orchestrator + oracle-backed engine, byte-identical tables and
JSON.  GPU (-m gpu): the same through the HIP engine."""
import hashlib
import json
import os

import pytest

import cases
from workdir_util import product_workdir

SPECS = {
    "lenet": (lambda: __import__("model.lenet.lenet", fromlist=["Cnn"]).Cnn(1, 10), "1,1,28,28", (4, 1, 28, 28)),
    "concat": (cases.tiny_concat_net, "1,3,8,8", (4, 3, 8, 8)),
    # (round 5) models that are not ResNets: a plain 3x3 stack with max-pools and a two-layer classifier -- the direct and the
    # Winograd float kernels without a residual anywhere -- and depthwise-separable blocks, whose grouped convolutions stay on torch
    "vgg": (cases.tiny_vgg_net, "1,3,16,16", (4, 3, 16, 16)),
    "separable": (cases.tiny_separable_net, "1,3,16,16", (4, 3, 16, 16)),
}


def _run(tag, quantity_cls, device):
    ctor, shape_str, bshape = SPECS[tag]
    out = {}
    with product_workdir(input_shape=shape_str, device=device, max_cali_img_num=2) as tmp:
        model = cases.seed_model(ctor(), base_seed=7).eval()
        if device == "gpu":
            model = model.cuda()
        q = quantity_cls(model)
        out["net_info"] = dict(q.net_info)
        out["net_info_order"] = list(q.net_info.keys())
        out["cared_op_layer_names"] = q.cared_op_layer_names
        out["merge_groups"] = q.get_merge_groups(q.net_info)
        q.activation_quantize(cases.calib_batches(4, bshape, seed=4321))
        wd = os.path.join(tmp, "test", "workdir")
        out["feat_table"] = open(os.path.join(wd, "feat.table")).read()
        q.weight_quantize()
        out["weight_table"] = open(os.path.join(wd, "weight.table")).read()
        out["files"] = {d: {f: open(os.path.join(wd, d, f)).read() for f in sorted(os.listdir(os.path.join(wd, d)))}
                        for d in ("bias", "new_bias")}
        out["weight_files_sha"] = {f: hashlib.sha256(open(os.path.join(wd, "weight", f), "rb").read()).hexdigest()
                                   for f in sorted(os.listdir(os.path.join(wd, "weight")))}
    return out


def _check(got, ref):
    for key in ("net_info_order", "net_info", "cared_op_layer_names", "merge_groups", "feat_table", "weight_table",
                "files", "weight_files_sha"):
        assert got[key] == ref[key], key


@pytest.fixture(scope="module")
def g9(golden_dir):
    with open(os.path.join(golden_dir, "g9_small_nets.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("tag", ["lenet", "concat", "vgg", "separable"])
def test_small_net_cpu_matches_reference(g9, oracle, tag):
    from engine_doubles import OracleCollector, OracleQuantizer
    from tools import Quantity

    class CpuQuantity(Quantity):
        collector_cls = OracleCollector
        quantizer_cls = OracleQuantizer

    _check(_run(tag, CpuQuantity, "cpu"), g9[tag])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["lenet", "concat", "vgg", "separable"])
def test_small_net_gpu_matches_reference(g9, tag):
    from tools import Quantity
    _check(_run(tag, Quantity, "gpu"), g9[tag])
