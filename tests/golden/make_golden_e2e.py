#!/usr/bin/env python3
"""Capture end-to-end golden fixtures (G3, G4, G6, G7 of SURVEY.md section 8c) by running the imported
reference in the build container:

    python tests/golden/make_golden_e2e.py [r18] [g6] [g7] [dkl] [small] [random]

Writes tests/golden/g3_r18_e2e.json (+ g4_r18_recon.npz), g6_rewriter.json, g7_netinfo.json.
Fixtures hold inputs' recipes and the reference's outputs only.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import _refenv  # noqa: E402

OURS = os.path.join(ROOT, "pytorch-quantity_amd", "quantity")


def _sha_file(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def _read(path):
    with open(path) as fh:
        return fh.read()


def _dir_state(d):
    return {f: _sha_file(os.path.join(d, f)) for f in sorted(os.listdir(d))}


def _jsonable(v):
    if isinstance(v, (np.floating,)):
        return {"f32" if isinstance(v, np.float32) else "f64": float(v)}
    if isinstance(v, (int, float)):
        return {"py": v}
    raise TypeError(type(v))


def capture_r18(cq, tl):
    import torch
    from model.resnet.ResNet_18_fabu import ResNet18          # the REFERENCE's model file (first on sys.path)
    import tools.pytorch_quantizer as pq
    recorded = {"collectors": [], "quantizers": []}
    orig_c, orig_q = cq.DistributionCollector, cq.Quantizer

    def make_collector(*a, **k):           # plain factories: the instances stay picklable for Pool
        obj = orig_c(*a, **k)
        recorded["collectors"].append(obj)
        return obj

    def make_quantizer(*a, **k):
        obj = orig_q(*a, **k)
        recorded["quantizers"].append(obj)
        return obj

    pq.DistributionCollector = make_collector
    pq.Quantizer = make_quantizer
    out = {}
    with _refenv.reference_workdir(input_shape="1,3,32,32", max_cali_img_num=1) as tmp:
        torch.manual_seed(0)
        model = cases.seed_model(ResNet18()).eval()
        model = cq.merge_bn(model, "cpu")
        batches = cases.calib_batches(3, (4, 3, 32, 32))
        q = tl.Quantity(model)
        out["net_info"] = {k: v for k, v in q.net_info.items()}
        out["net_info_order"] = list(q.net_info.keys())
        out["cared_op_layer_names"] = q.cared_op_layer_names
        out["merge_groups"] = q.get_merge_groups(q.net_info)
        out["layers_num"] = q.layers_num
        q.activation_quantize(batches)
        coll = recorded["collectors"][0]
        quan = recorded["quantizers"][0]
        out["max_vals"] = {k: _jsonable(v) for k, v in coll.max_vals.items()}
        out["intervals_final"] = {k: _jsonable(v) for k, v in coll._distribution_intervals.items()}
        out["hist_sums"] = {k: int(np.asarray(v).sum()) for k, v in coll.distributions.items()}
        out["bits_final"] = {k: int(v) for k, v in quan.bits.items()}
        out["threshold_value"] = {k: _jsonable(v) for k, v in quan.threshold_value.items()}
        wd = os.path.join(tmp, "test", "workdir")
        out["feat_table"] = _read(os.path.join(wd, "feat.table"))
        q.weight_quantize()
        out["weight_table_after_quantize"] = _read(os.path.join(wd, "weight.table"))
        state1 = {d: _dir_state(os.path.join(wd, d)) for d in ("weight", "bias", "new_weight", "new_bias")}
        out["files_after_quantize"] = state1
        out["verbatim"] = {
            "bias/fc.bias.json": _read(os.path.join(wd, "bias", "fc.bias.json")),
            "new_bias/fc.bias.json": _read(os.path.join(wd, "new_bias", "fc.bias.json")),
            "new_bias/conv1.0.bias.json": _read(os.path.join(wd, "new_bias", "conv1.0.bias.json")),
            "weight/conv1.0.weight.json": _read(os.path.join(wd, "weight", "conv1.0.weight.json")),
        }
        q.rewrite_weight()                                     # the script's second call (quirk 2)
        out["weight_table_after_second_rewrite"] = _read(os.path.join(wd, "weight.table"))
        out["files_after_second_rewrite"] = {d: _dir_state(os.path.join(wd, d))
                                             for d in ("weight", "bias", "new_weight", "new_bias")}
        out["verbatim"]["new_bias/fc.bias.json@second"] = _read(os.path.join(wd, "new_bias", "fc.bias.json"))

        # G4: reconstruction on fresh, identically seeded models, tables as left by the run above
        x = cases.fixed_input((4, 3, 32, 32))
        arrays = {"x": x.numpy()}
        m_float = cases.seed_model(ResNet18()).eval()
        rec = tl.Reconstruction(m_float)
        merged = rec.merge_bn().eval()
        with torch.no_grad():
            arrays["logits_merged"] = merged(x).numpy()
        info = rec.get_quantity_information()
        out["quantity_information"] = {k: {kk: vv for kk, vv in v.items() if kk != "layer"} for k, v in info.items()}
        recon = rec.ReconModel(info, os.path.join(wd, "recon.pth"))
        with torch.no_grad():
            arrays["logits_recon"] = recon(x).numpy()
            first = recon.conv1[0]
            arrays["recon_conv1_out"] = first(x).numpy()
            qx = first.Quan(x)
            acc = first.Conv(qx)
            arrays["recon_conv1_acc_absmax"] = np.array(acc.abs().max().item())
        arrays["recon_conv1_qweight"] = recon.conv1[0].Conv.weight.detach().numpy()
        arrays["recon_conv1_qbias"] = recon.conv1[0].quantized_bias.numpy()
        out["recon_state_dict_keys"] = sorted(recon.state_dict().keys())
        # max |accumulator| over all layers (exactness bound of the reference's fp32 conv)
        accmax = []
        hooks = []
        for mod in recon.modules():
            if type(mod).__name__ in ("NewConv2d", "NewLinear"):
                inner = mod.Conv if hasattr(mod, "Conv") else mod.Linear
                hooks.append(inner.register_forward_hook(lambda m, i, o: accmax.append(o.abs().max().item())))
        with torch.no_grad():
            recon(x)
        for h in hooks:
            h.remove()
        out["recon_max_abs_accumulator"] = max(accmax)

        m2 = cases.seed_model(ResNet18()).eval()
        rec2 = tl.Reconstruction(m2)
        rec2.merge_bn()
        info2 = rec2.get_quantity_information()
        tmodel = rec2.ReconTest(info2, os.path.join(wd, "recontest.pth"))
        with torch.no_grad():
            arrays["logits_recontest"] = tmodel(x).numpy()
            arrays["recontest_conv1_out"] = tmodel.conv1[0](x).numpy()
        out["recontest_state_dict_keys"] = sorted(tmodel.state_dict().keys())
    np.savez_compressed(os.path.join(HERE, "g4_r18_recon.npz"), **arrays)
    with open(os.path.join(HERE, "g3_r18_e2e.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("G3/G4 written; feat.table:\n" + out["feat_table"])


def capture_dkl(cq, tl):
    """weight_quantize() with the KL branch enabled (_DKL_weight = True, pytorch_quantizer.py:644-648)."""
    import torch
    from model.resnet.ResNet_18_fabu import ResNet18
    with open(os.path.join(HERE, "g3_r18_e2e.json")) as fh:
        g3 = json.load(fh)
    out = {}
    with _refenv.reference_workdir(input_shape="1,3,32,32", max_cali_img_num=1) as tmp:
        model = cq.merge_bn(cases.seed_model(ResNet18()).eval(), "cpu")
        q = tl.Quantity(model)
        wd = os.path.join(tmp, "test", "workdir")
        with open(os.path.join(wd, "feat.table"), "w") as fh:
            fh.write(g3["feat_table"])
        q._DKL_weight = True
        q.weight_quantize()
        out["weight_table"] = _read(os.path.join(wd, "weight.table"))
        out["files"] = {d: _dir_state(os.path.join(wd, d)) for d in ("weight", "bias", "new_weight", "new_bias")}
    with open(os.path.join(HERE, "g3b_r18_dkl_weights.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("DKL weight table:\n" + out["weight_table"])


def capture_small(cq, tl):
    """End-to-end tables for four small nets: the LeNet fixture (biased convs, MaxPool, Linear chain), a net with a Concat merge
    group, a plain VGG-like stack and a depthwise-separable one (grouped convolutions)."""
    import torch
    from model.lenet.lenet import Cnn                      # the reference's model file
    out = {}
    specs = (("lenet", lambda: Cnn(1, 10), "1,1,28,28", (4, 1, 28, 28)),
             ("concat", cases.tiny_concat_net, "1,3,8,8", (4, 3, 8, 8)),
             ("vgg", cases.tiny_vgg_net, "1,3,16,16", (4, 3, 16, 16)),
             ("separable", cases.tiny_separable_net, "1,3,16,16", (4, 3, 16, 16)))
    for tag, ctor, shape_str, bshape in specs:
        with _refenv.reference_workdir(input_shape=shape_str, max_cali_img_num=2) as tmp:
            model = cases.seed_model(ctor(), base_seed=7).eval()
            q = tl.Quantity(model)
            rec = {"net_info": {k: v for k, v in q.net_info.items()}, "net_info_order": list(q.net_info.keys()),
                   "cared_op_layer_names": q.cared_op_layer_names, "merge_groups": q.get_merge_groups(q.net_info)}
            q.activation_quantize(cases.calib_batches(4, bshape, seed=4321))
            wd = os.path.join(tmp, "test", "workdir")
            rec["feat_table"] = _read(os.path.join(wd, "feat.table"))
            q.weight_quantize()
            rec["weight_table"] = _read(os.path.join(wd, "weight.table"))
            rec["files"] = {d: {f: _read(os.path.join(wd, d, f)) for f in sorted(os.listdir(os.path.join(wd, d)))}
                            for d in ("bias", "new_bias")}
            rec["weight_files_sha"] = _dir_state(os.path.join(wd, "weight"))
            out[tag] = rec
            print("small", tag, "feat.table:", rec["feat_table"].replace("\n", " | "))
    with open(os.path.join(HERE, "g9_small_nets.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


# (index, seed, odd, share, bn): golden G11 -- share: nn.ReLU modules that serve several places of the graph (torchvision's style)
RANDOM_GRAPHS = ([(i, 101, False, False, False) for i in range(20)] + [(i, 102, True, False, False) for i in range(10)] +
                 [(i, 103, False, True, False) for i in range(8)] + [(i, 104, False, False, True) for i in range(8)])      # (.., bn)


def capture_random(cq, tl, graphs=None, out_dir=None):
    """G11: forty-six random topologies (cases.random_net: residual blocks with and without projection, two consumers of one tensor,
    concatenations, pools, in-place ReLUs; ten of them with depthwise / dilated convolutions and upsampling) through the REFERENCE:
    graph discovery, merge groups, the calibration's maxima and feat.table, weight.table.  A graph the reference itself rejects (its
    value fingerprints collide, or it finds an in-place module "useless") is recorded with the exception's type."""
    import torch
    out, logits = {}, {}
    for (index, seed, odd, share, bn) in (graphs or RANDOM_GRAPHS):
        tag = "%d/%d%s%s%s" % (index, seed, "/odd" if odd else "", "/share" if share else "", "/bn" if bn else "")
        model, size, bs, _rng = cases.random_net(index, seed, odd, share=share, bn=bn)
        if bn:
            model = cq.merge_bn(model, "cpu")                # (bn: BatchNorm2d behind half of the convolutions, folded first -- the reference's flow)
        rec = {"size": size, "batch": bs}
        try:
            with _refenv.reference_workdir(input_shape="1,%d,%d,%d" % (model.cin, size, size), max_cali_img_num=2) as tmp:
                torch.manual_seed(0)
                q = tl.Quantity(model)
                rec.update({"net_info": {k: v for k, v in q.net_info.items()}, "net_info_order": list(q.net_info.keys()),
                            "cared_op_layer_names": q.cared_op_layer_names, "merge_groups": q.get_merge_groups(q.net_info),
                            "layers_num": q.layers_num})
                q.activation_quantize(cases.calib_batches(3, (bs, model.cin, size, size), seed=9000 + index))
                wd = os.path.join(tmp, "test", "workdir")
                rec["feat_table"] = _read(os.path.join(wd, "feat.table"))
                q.weight_quantize()
                rec["weight_table"] = _read(os.path.join(wd, "weight.table"))
                # the integer-simulation model of the same graph, as the reference's flow builds it (tables as left above), on a
                # fixed input: logits of ReconModel (G4's check on graphs that are not ResNets)
                q.rewrite_weight()
                rec["weight_table_rewritten"] = _read(os.path.join(wd, "weight.table"))
                twin = cases.random_net(index, seed, odd, share=share, bn=bn)[0]
                r = tl.Reconstruction(twin)
                if bn:
                    r.merge_bn()
                info = r.get_quantity_information()
                recon = r.ReconModel(info, os.path.join(wd, "recon.pth"))
                x = cases.fixed_input((4, model.cin, size, size), seed=77 + index)
                with torch.no_grad():
                    logits[tag] = recon(x).numpy()
                rec["recon_layers"] = sorted(k for k in info.keys())
        except Exception as e:                               # what the reference does with this graph is part of the golden
            rec = {"size": size, "batch": bs, "reference_error": type(e).__name__, "message": str(e)[:60]}
            print("random", tag, "reference raises", type(e).__name__, str(e)[:80])
        out[tag] = rec
        if "feat_table" in rec:
            print("random", tag, "nodes", len(rec["net_info"]), "feat.table:", rec["feat_table"].replace("\n", " | ")[:100])
    with open(os.path.join(out_dir or HERE, "g11_random_graphs.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(out_dir or HERE, "g11_random_recon.npz"), **{k.replace("/", "_"): v for k, v in logits.items()})
    print("G11:", len(out), "graphs,", len(logits), "with ReconModel logits")


def capture_g6(cq, tl):
    """BiasReWriter on a crafted directory: int8 wrap, negative bits, MAX_SHIFT capping."""
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for sub in ("weight", "bias", "new_weight", "new_bias"):
            os.makedirs(os.path.join(d, sub))
        feat = "image 5\nc1 3 5\nc2 6 3\nadd 6 6 3\nfc 1 6\n"
        wt = "c1.weight 12\nc1.bias 9\nc2.weight 7\nc2.bias 2\nfc.weight 9\nfc.bias 4\n"
        with open(os.path.join(d, "feat.table"), "w") as fh:
            fh.write(feat)
        with open(os.path.join(d, "weight.table"), "w") as fh:
            fh.write(wt)
        params = {
            "weight/c1.weight.json": [[[[127, -128], [5, -7]]], [[[1, 0], [-1, 64]]]],
            "weight/c2.weight.json": [[[[3]], [[-3]]], [[[100]], [[-100]]]],
            "weight/fc.weight.json": [[127, -128, 33], [1, 2, 3]],
            "bias/c1.bias.json": [127, -128],
            "bias/c2.bias.json": [8, 13, -9, 127, -128, 100],      # 2 -> 6 bits: x16 wraps
            "bias/fc.bias.json": [127, 5],                           # 4 -> 1 bits: /8 half-even
        }
        for rel, content in params.items():
            with open(os.path.join(d, rel), "w") as fh:
                json.dump(content, fh, indent=4)
        rw = tl.BiasReWriter(os.path.join(d, "weight"), os.path.join(d, "bias"), os.path.join(d, "new_weight"),
                             os.path.join(d, "new_bias"), os.path.join(d, "weight.table"),
                             os.path.join(d, "feat.table"), max_shift_limit=12)
        weight_bits, bias_bits = rw.get_weight_info()
        feat_bits, infeat_bits = rw.get_feat_info()
        rw.rewrite_bias_table(bias_bits, feat_bits)
        rw.rewrite_bias_dir(bias_bits, feat_bits)
        need, new_w = rw.max_shift_limit_weight(feat_bits, infeat_bits, weight_bits)
        if need:
            rw.rewrite_weight_table(weight_bits, new_w)
            rw.rewrite_weight_dir(weight_bits, new_w)
        out["inputs"] = {"feat.table": feat, "weight.table": wt, "params": params}
        out["need_rewrite"] = bool(need)
        out["new_weight_bits"] = new_w
        out["weight.table"] = _read(os.path.join(d, "weight.table"))
        out["files"] = {}
        for sub in ("new_weight", "new_bias"):
            for f in sorted(os.listdir(os.path.join(d, sub))):
                out["files"][sub + "/" + f] = _read(os.path.join(d, sub, f))
    with open(os.path.join(HERE, "g6_rewriter.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("G6 written:", out["weight.table"].replace("\n", " | "), list(out["files"].keys()))


def capture_g7(cq, tl):
    """net_info / merge groups / cared names of OUR fabu ResNet-50 / ResNet-101 definitions as the
    reference's graph discovery sees them."""
    import torch
    sys.path.append(OURS)                      # model.resnet.ResNet_fabu only exists in our tree
    from model.resnet.ResNet_fabu import ResNet50, ResNet101
    out = {}
    for tag, ctor, shape in (("r50", ResNet50, "1,3,224,224"), ("r101", ResNet101, "1,3,224,224")):
        with _refenv.reference_workdir(input_shape=shape):
            torch.manual_seed(0)
            # gamma_scale keeps the 33-block net's activations O(1): the reference's value
            # fingerprints (tid) collide once activations grow large (the topology is weight independent)
            model = cases.seed_model(ctor(), gamma_scale=0.5).eval()
            model = cq.merge_bn(model, "cpu")
            q = tl.Quantity(model)
            out[tag] = {
                "net_info": {k: v for k, v in q.net_info.items()},
                "net_info_order": list(q.net_info.keys()),
                "cared_op_layer_names": q.cared_op_layer_names,
                "merge_groups": q.get_merge_groups(q.net_info),
                "layers_num": q.layers_num,
            }
            print("G7", tag, "nodes", len(q.net_info), "layers_num", q.layers_num)
    with open(os.path.join(HERE, "g7_netinfo.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


def main():
    which = sys.argv[1:] or ["r18", "g6", "g7"]
    cq, tl = _refenv.import_reference()
    if "g6" in which:
        capture_g6(cq, tl)
    if "g7" in which:
        capture_g7(cq, tl)
    if "r18" in which:
        capture_r18(cq, tl)
    if "dkl" in which:
        capture_dkl(cq, tl)
    if "small" in which:
        capture_small(cq, tl)
    if "random" in which:
        capture_random(cq, tl)
    if "random_sweep" in which:
        # not a golden: a larger family written to FQ_G11_SWEEP_DIR for a one-off comparison in the build container
        # (FQ_G11_DIR=<that directory> python -m pytest tests/test_random_graphs.py -m "not gpu")
        n, seed = int(os.environ.get("FQ_G11_SWEEP_N", "100")), int(os.environ.get("FQ_G11_SWEEP_SEED", "201"))
        graphs = [(i, seed, i % 4 == 1, i % 4 == 2, i % 4 == 3) for i in range(n)]
        capture_random(cq, tl, graphs, os.environ["FQ_G11_SWEEP_DIR"])


if __name__ == "__main__":
    main()
