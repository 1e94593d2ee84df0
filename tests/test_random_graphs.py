"""Golden G11: forty-six RANDOM model topologies (cases.random_net -- residual blocks with and without projection, a tensor with two
consumers, concatenations, pools, in-place ReLUs; ten with depthwise / dilated convolutions and upsampling, eight with nn.ReLU modules that serve several places, eight with BatchNorm2d layers that merge_bn folds first) taken through the imported
REFERENCE in the build container (tests/golden/make_golden_e2e.py random): graph discovery, merge groups, feat.table, weight.table.
CPU: the drop-in's orchestrator with the oracle-backed engine gives the reference's graph and byte-identical tables on every graph the
reference accepts; the seven it rejects (its value fingerprints do not survive an in-place ReLU: "Can't find the input tensor") the
drop-in discovers by tensor identity.  GPU (-m gpu): the HIP engine gives the reference's tables up to near ties of the KL search."""
import json
import os

import pytest

import cases
from workdir_util import product_workdir


# (FQ_G11_DIR: another directory holding the two files -- a larger sweep captured in the build container, `make_golden_e2e.py random_sweep`)
G11_DIR = os.environ.get("FQ_G11_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g11():
    with open(os.path.join(G11_DIR, "g11_random_graphs.json")) as fh:
        return json.load(fh)


def _tags():
    with open(os.path.join(G11_DIR, "g11_random_graphs.json")) as fh:
        return sorted(json.load(fh).keys())


def _model(tag):
    """(model with its BatchNorms folded, if it has any -- the flow's first step --, image size, batch size, rng)"""
    from common.quantity import merge_bn
    parts = tag.split("/")
    model, size, bs, rng = cases.random_net(int(parts[0]), int(parts[1]), "odd" in parts[2:], share="share" in parts[2:], bn="bn" in parts[2:])
    return (merge_bn(model) if model.has_bn else model), size, bs, rng


def _run(tag, quantity_cls, device, tables=True):
    model, size, bs, _rng = _model(tag)
    out = {}
    with product_workdir(input_shape="1,%d,%d,%d" % (model.cin, size, size), device=device, max_cali_img_num=2) as tmp:
        if device == "gpu":
            model = model.cuda()
        q = quantity_cls(model)
        out.update({"net_info": dict(q.net_info), "net_info_order": list(q.net_info.keys()), "cared_op_layer_names": q.cared_op_layer_names,
                    "merge_groups": q.get_merge_groups(q.net_info), "layers_num": q.layers_num})
        if tables:
            index = int(tag.split("/")[0])
            q.activation_quantize(cases.calib_batches(3, (bs, model.cin, size, size), seed=9000 + index))
            wd = os.path.join(tmp, "test", "workdir")
            out["feat_table"] = open(os.path.join(wd, "feat.table")).read()
            q.weight_quantize()
            out["weight_table"] = open(os.path.join(wd, "weight.table")).read()
    return out


@pytest.mark.parametrize("tag", _tags())
def test_random_graph_cpu_matches_reference(g11, oracle, tag):
    from engine_doubles import OracleCollector, OracleQuantizer
    from tools import Quantity

    class CpuQuantity(Quantity):
        collector_cls = OracleCollector
        quantizer_cls = OracleQuantizer

    ref = g11[tag]
    if "reference_error" in ref:
        # the reference rejects this graph ("Can't find the input tensor of ReLU_n": a value fingerprint taken before an in-place ReLU
        # does not match the one taken after it); identity tracking finds it: every node has its inputs, the tables get written
        got = _run(tag, CpuQuantity, "cpu")
        assert ref["reference_error"] == "AssertionError" and got["feat_table"].startswith("image ")
        assert all(info["inputs"] for name, info in list(got["net_info"].items())[1:])
        return
    got = _run(tag, CpuQuantity, "cpu")
    for key in ("net_info_order", "net_info", "cared_op_layer_names", "merge_groups", "layers_num", "feat_table", "weight_table"):
        assert got[key] == ref[key], key


def _bits(table):
    return {line.split()[0]: [int(v) for v in line.split()[1:]] for line in table.strip().split("\n")}


@pytest.mark.gpu
def test_random_graphs_gpu_match_reference(g11):
    """The HIP engine on the same graphs and batches: the reference's graph, its weight.table byte for byte wherever the feat.table is, and its feat.table up to
    near ties of the KL search (fp32 sums in another order move a few elements across bin edges: a line may be one bit apart)."""
    from tools import Quantity
    lines = apart = 0
    for tag in _tags():
        ref = g11[tag]
        if "reference_error" in ref:
            continue
        got = _run(tag, Quantity, "gpu")
        for key in ("net_info_order", "net_info", "cared_op_layer_names", "merge_groups", "layers_num"):
            assert got[key] == ref[key], (tag, key)
        # weight.table: the weights' lines never depend on an activation; a bias line follows its layer's feat.table line
        if got["feat_table"] == ref["feat_table"]:
            assert got["weight_table"] == ref["weight_table"], tag
        else:
            wa, wb = _bits(got["weight_table"]), _bits(ref["weight_table"])
            assert wa.keys() == wb.keys() and all(wa[k] == wb[k] for k in wa if k.endswith(".weight")), tag
            assert all(abs(x - y) <= 1 for k in wa for x, y in zip(wa[k], wb[k])), tag
        a, b = _bits(got["feat_table"]), _bits(ref["feat_table"])
        assert a.keys() == b.keys(), tag
        for k in a:
            assert len(a[k]) == len(b[k]) and all(abs(x - y) <= 1 for x, y in zip(a[k], b[k])), (tag, k, a[k], b[k])
            lines += 1
            apart += int(a[k] != b[k])
    assert lines > 300 and apart <= 0.03 * lines, (lines, apart)


@pytest.mark.gpu
def test_random_graphs_reconmodel_logits_equal_the_reference(g11, oracle, golden_dir):
    """G4's check on graphs that are not ResNets: the reference's ReconModel logits (CPU, fixed input) of every graph it accepts,
    against the integer-simulation model on the HIP kernels -- with fp32 module boundaries and with the resident integer plan --
    bit for bit.  The tables the models are built from come from the oracle-backed CPU calibration, which the CPU test above pins to
    the reference's byte for byte."""
    import numpy as np
    import torch
    from engine_doubles import OracleCollector, OracleQuantizer
    from common.quantity import resident
    from tools import Quantity, Reconstruction

    class CpuQuantity(Quantity):
        collector_cls = OracleCollector
        quantizer_cls = OracleQuantizer

    logits = np.load(os.path.join(G11_DIR, "g11_random_recon.npz"))
    checked = planned = 0
    for tag in _tags():
        ref = g11[tag]
        if "reference_error" in ref:
            continue
        index = int(tag.split("/")[0])
        model, size, bs, _rng = _model(tag)
        with product_workdir(input_shape="1,%d,%d,%d" % (model.cin, size, size), device="cpu", max_cali_img_num=2) as tmp:
            q = CpuQuantity(model)
            q.activation_quantize(cases.calib_batches(3, (bs, model.cin, size, size), seed=9000 + index))
            q.weight_quantize()
            q.rewrite_weight()
            wd = os.path.join(tmp, "test", "workdir")
            assert open(os.path.join(wd, "weight.table")).read() == ref["weight_table_rewritten"], tag
            rec = Reconstruction(_model(tag)[0])
            info = rec.get_quantity_information()
            assert sorted(info.keys()) == ref["recon_layers"], tag
            net = rec.ReconModel(info, os.path.join(wd, "recon.pth")).cuda()
            x = cases.fixed_input((4, model.cin, size, size), seed=77 + index).cuda()
            want = logits[tag.replace("/", "_")]
            with torch.no_grad():
                np.testing.assert_array_equal(net(x).cpu().numpy(), want, err_msg=tag)
                summary = resident.enable(net, x)
                np.testing.assert_array_equal(net(x).cpu().numpy(), want, err_msg=tag + " (resident)")
            checked += 1
            planned += summary["resident_convs"]
    assert checked >= 20 and planned > 150, (checked, planned)
