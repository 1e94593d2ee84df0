#!/usr/bin/env python3
"""Capture, by running the imported reference in the build container,

    python tests/golden/make_golden_r50.py [r50] [scales] [h6] [time18]

  r50     G4-R50 (tests/golden/g4_r50_recon.npz + g4_r50_tables.json): the reference's own calibration, weight
          quantisation, ReconModel and ReconTest (quantity/tools/reconstruction.py:175-324,
          quantity/common/quantity/new_quantity_op.py:124-133,280-292) on the build's fabu ResNet-50 @224^2 --
          BASELINE configs 2/3 at their own model size: tables as text, logits for a fixed 2x3x224x224 input,
          a sub-sample + sha256 of the first layer's output, max |accumulator| over all integer layers.
  r50stats  g4_r50_calib_stats.npz: what the reference's calibration of that model handed to its KL search (the merged
          intervals and 2048-bin histograms of all 71 rows, quantity/tools/pytorch_quantizer.py:393-448) and the bits it
          found -- so that a GPU calibration whose float forward differs in the last bits can be compared row by row
          (maxima, histogram L1 distance), not only through the table.
  scales  g4_r50_bn_scales.npz: the BatchNorm fold factors of that model as the reference's merge_bn computes them in the
          build container (torch.sqrt on CPU is machine dependent in the last bit; see cases.fold_bn_with_scales).
  h6      G10 (g10_dilation.npz): Quantity.dilation_to_zero_padding (quantity/tools/pytorch_quantizer.py:679-693)
          on seeded kernels.
  (time18 and time101 also KEEP what those runs computed -- G12 g12_r18_config1.json + g12_r18_config1_stats.npz: feat.table,
          weight.table, the JSON files' sha256, and what the KL search was handed, for config 1 at its stated 256 images; G13
          g13_r101_512_stats.npz: the same statistics + feat.table + the BatchNorm fold factors for ResNet-101 @512^2, one image)
  time18  wall time of the reference's Python path on BASELINE config 1 (ResNet-18, 256 synthetic 3x32x32 images:
          2 batches of 128 with MAX_CALI_IMG_NUM = 1, WORKER_NUM 4 as shipped) -> tests/golden/ref_timing_r18.json
          (recorded in BASELINE.md; a measurement, not a parity fixture).

  time101 wall time of the reference's Python path on BASELINE config 5's shape (ResNet-101 @3x512x512), ONE image
          (MAX_CALI_IMG_NUM 0) -> tests/golden/ref_timing_r101_512.json (recorded in BASELINE.md; a measurement).

Fixtures hold input recipes (seeds) and the reference's outputs only; no reference source enters the repo.
matplotlib's hist() is replaced by a no-op while ReconTest is constructed (the reference's TestConv constructor draws
four 2048-bin PNG histograms per layer; they are not part of any output compared here).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import _refenv  # noqa: E402

OURS = os.path.join(ROOT, "pytorch-quantity_amd", "quantity")
GAMMA = 0.5         # keeps the 16-block net's activations O(1); the reference's tid fingerprints collide on exploding ones


def _read(path):
    with open(path) as fh:
        return fh.read()


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _dir_state(d):
    out = {}
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            out[f] = hashlib.sha256(fh.read()).hexdigest()
    return out


def capture_r50(cq, tl):
    import torch
    import matplotlib.pyplot as plt
    sys.path.append(OURS)                      # model.resnet.ResNet_fabu only exists in our tree
    from model.resnet.ResNet_fabu import ResNet50
    torch.set_num_threads(8)
    out, arrays = {"gamma_scale": GAMMA, "calib": {"n_batches": 2, "shape": [2, 3, 224, 224], "seed": 77},
                   "input": {"shape": [2, 3, 224, 224], "seed": 99}}, {}
    with _refenv.reference_workdir(input_shape="1,3,224,224", max_cali_img_num=1) as tmp:
        t0 = time.time()
        model = cq.merge_bn(cases.seed_model(ResNet50(), gamma_scale=GAMMA).eval(), "cpu")
        q = tl.Quantity(model)
        q.activation_quantize(cases.calib_batches(2, (2, 3, 224, 224), seed=77))
        print("reference activation_quantize: %.1f s" % (time.time() - t0), flush=True)
        wd = os.path.join(tmp, "test", "workdir")
        out["feat_table"] = _read(os.path.join(wd, "feat.table"))
        t0 = time.time()
        q.weight_quantize()
        print("reference weight_quantize: %.1f s" % (time.time() - t0), flush=True)
        out["weight_table"] = _read(os.path.join(wd, "weight.table"))
        out["files"] = {d: _dir_state(os.path.join(wd, d)) for d in ("weight", "bias", "new_weight", "new_bias")}

        x = cases.fixed_input((2, 3, 224, 224))
        rec = tl.Reconstruction(cases.seed_model(ResNet50(), gamma_scale=GAMMA).eval())
        merged = rec.merge_bn().eval()
        with torch.no_grad():
            arrays["logits_merged"] = merged(x).numpy()
        info = rec.get_quantity_information()
        out["quantity_information"] = {k: {kk: vv for kk, vv in v.items() if kk != "layer"} for k, v in info.items()}
        recon = rec.ReconModel(info, os.path.join(wd, "recon.pth"))
        accmax, hooks = [], []
        for mod in recon.modules():
            if type(mod).__name__ in ("NewConv2d", "NewLinear"):
                inner = mod.Conv if hasattr(mod, "Conv") else mod.Linear
                hooks.append(inner.register_forward_hook(lambda m, i, o: accmax.append(o.abs().max().item())))
        with torch.no_grad():
            arrays["logits_recon"] = recon(x).numpy()
            c1 = recon.conv1(x).numpy()
        for h in hooks:
            h.remove()
        out["recon_max_abs_accumulator"] = max(accmax)
        out["recon_conv1_out_sha256"] = _sha(c1)
        arrays["recon_conv1_out_sample"] = c1[:, :8, ::8, ::8].copy()
        # stage outputs as sha256 (+ a small sub-sample) and the classifier's input in full: localises a mismatch
        # without shipping 6 MB tensors
        with torch.no_grad():
            s = recon.maxpool(recon.relu(recon.conv1(x)))
            out["recon_stage_sha256"] = {}
            for stage in ("layer1", "layer2", "layer3", "layer4"):
                s = getattr(recon, stage)(s)
                out["recon_stage_sha256"][stage] = _sha(s.numpy())
                arrays["recon_%s_sample" % stage] = s.numpy()[:, :16, ::4, ::4].copy()
            out["recon_layer1_out_sha256"] = out["recon_stage_sha256"]["layer1"]
            pooled = recon.view(recon.avgpool(s))
            arrays["recon_fc_input"] = pooled.numpy().copy()
            arrays["recon_layer4_out_int8"] = np.round(s.numpy() * 2.0 ** info["layer4.2.Eltwise"]["output_bit"]).astype(np.int16)[:, :64].copy()
        out["recon_state_dict_keys"] = sorted(recon.state_dict().keys())

        rec2 = tl.Reconstruction(cases.seed_model(ResNet50(), gamma_scale=GAMMA).eval())
        rec2.merge_bn()
        info2 = rec2.get_quantity_information()
        real_hist = plt.hist
        plt.hist = lambda *a, **k: None
        try:
            t0 = time.time()
            tmodel = rec2.ReconTest(info2, os.path.join(wd, "recontest.pth"))
            print("reference ReconTest construction: %.1f s" % (time.time() - t0), flush=True)
        finally:
            plt.hist = real_hist
        with torch.no_grad():
            arrays["logits_recontest"] = tmodel(x).numpy()
            t1 = tmodel.conv1(x).numpy()
        out["recontest_conv1_out_sha256"] = _sha(t1)
        arrays["recontest_conv1_out_sample"] = t1[:, :8, ::8, ::8].copy()
    np.savez_compressed(os.path.join(HERE, "g4_r50_recon.npz"), **arrays)
    with open(os.path.join(HERE, "g4_r50_tables.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("G4-R50 written; max |acc| = %s; feat.table head: %s" % (out["recon_max_abs_accumulator"],
                                                                     out["feat_table"].split("\n")[:3]))


def capture_r50_stats(cq, tl):
    import torch
    sys.path.append(OURS)
    from model.resnet.ResNet_fabu import ResNet50
    torch.set_num_threads(8)
    seen = {}
    qmod = sys.modules[tl.Quantity.__module__]
    real = qmod.Quantizer.quantize

    def spy(self, distributions, distribution_intervals):
        seen["names"] = list(distributions.keys())
        seen["hist"] = np.stack([np.asarray(distributions[k], dtype=np.int64) for k in seen["names"]])
        seen["interval"] = np.array([distribution_intervals[k] for k in seen["names"]], dtype=np.float64)
        real(self, distributions, distribution_intervals)
        seen["bits"] = np.array([self.bits[k] for k in seen["names"]], dtype=np.int64)
    qmod.Quantizer.quantize = spy
    try:
        with _refenv.reference_workdir(input_shape="1,3,224,224", max_cali_img_num=1) as tmp:
            model = cq.merge_bn(cases.seed_model(ResNet50(), gamma_scale=GAMMA).eval(), "cpu")
            q = tl.Quantity(model)
            q.activation_quantize(cases.calib_batches(2, (2, 3, 224, 224), seed=77))
            feat = _read(os.path.join(tmp, "test", "workdir", "feat.table"))
    finally:
        qmod.Quantizer.quantize = real
    with open(os.path.join(HERE, "g4_r50_tables.json")) as fh:
        assert json.load(fh)["feat_table"] == feat, "this run's feat.table differs from the committed G4-R50 one"
    np.savez_compressed(os.path.join(HERE, "g4_r50_calib_stats.npz"), names=np.array(seen["names"]), hist=seen["hist"],
                        interval=seen["interval"], bits=seen["bits"])
    print("G4-R50 calibration statistics written:", seen["hist"].shape, "rows x bins; elements per row",
          seen["hist"].sum(axis=1)[:4], "...")


def capture_scales(cq, tl):
    """g4_r50_bn_scales.npz: the BatchNorm fold factors gamma / sqrt(running_var + 1e-5) of the seeded ResNet-50 exactly
    as the reference's merge_bn evaluates them HERE (utils.py:37: torch ops on CPU tensors -- torch.sqrt goes through MKL
    and is machine dependent in the last bit).  Checked on the spot: folding with these factors (cases.fold_bn_with_scales)
    reproduces the reference's merged parameters bit for bit."""
    import torch
    sys.path.append(OURS)
    from model.resnet.ResNet_fabu import ResNet50
    ref_merged = cq.merge_bn(cases.seed_model(ResNet50(), gamma_scale=GAMMA).eval(), "cpu").state_dict()
    model = cases.seed_model(ResNet50(), gamma_scale=GAMMA).eval()
    scales = {}
    for name, layer in model.named_modules():
        if type(layer).__name__ == "BatchNorm2d":
            scales[name] = (layer.weight.data / torch.sqrt(layer.running_var + 1e-5)).numpy().copy()
    # fold with the drop-in's Identity class is not importable here (the reference's `common` is loaded): inline check
    sd = model.state_dict()
    pending = None
    for name, layer in model.named_modules():
        kind = type(layer).__name__
        if kind == "Conv2d":
            pending = name
        elif kind == "BatchNorm2d":
            sc = scales[name]
            w = sd[pending + ".weight"].numpy()
            got_w = sc.reshape(-1, 1, 1, 1) * w
            got_b = sc * (np.zeros_like(sc) - sd[name + ".running_mean"].numpy()) + sd[name + ".bias"].numpy()
            assert np.array_equal(got_w, ref_merged[pending + ".weight"].numpy()), pending
            assert np.array_equal(got_b, ref_merged[pending + ".bias"].numpy()), pending
    np.savez_compressed(os.path.join(HERE, "g4_r50_bn_scales.npz"), **scales)
    print("BN fold factors written:", len(scales), "layers,", sum(v.size for v in scales.values()), "channels; "
          "numpy fold == reference merge_bn bit for bit")


def capture_h6(cq, tl):
    rng = np.random.default_rng(606)
    arrays = {}
    for tag, shape in (("k3", (5, 4, 3, 3)), ("k1", (2, 3, 1, 1)), ("k5", (3, 2, 5, 5)), ("k2", (1, 1, 2, 2))):
        w = rng.standard_normal(shape).astype(np.float32)
        arrays[tag + "_in"] = w
        arrays[tag + "_out"] = tl.Quantity.dilation_to_zero_padding(None, w, (2, 2))
    np.savez_compressed(os.path.join(HERE, "g10_dilation.npz"), **arrays)
    print("G10 written:", {k: v.shape for k, v in arrays.items()})


class _kl_spy(object):
    """What the reference's calibration hands to its KL search (Quantizer.quantize, called at pytorch_quantizer.py:447 with
    the merged intervals and histograms) and the bits it finds, captured in passing."""

    def __init__(self, tl):
        self.qmod, self.seen = sys.modules[tl.Quantity.__module__], {}

    def __enter__(self):
        seen, real = self.seen, self.qmod.Quantizer.quantize
        self.real = real

        def spy(q, distributions, distribution_intervals):
            seen["names"] = list(distributions.keys())
            seen["hist"] = np.stack([np.asarray(distributions[k], dtype=np.int64) for k in seen["names"]])
            seen["interval"] = np.array([distribution_intervals[k] for k in seen["names"]], dtype=np.float64)
            real(q, distributions, distribution_intervals)
            seen["bits"] = np.array([q.bits[k] for k in seen["names"]], dtype=np.int64)
        self.qmod.Quantizer.quantize = spy
        return seen

    def __exit__(self, *exc):
        self.qmod.Quantizer.quantize = self.real
        return False


def _save_stats(path, seen, feat, extra=None):
    np.savez_compressed(path, names=np.array(seen["names"]), hist=seen["hist"], interval=seen["interval"], bits=seen["bits"],
                        feat_table=np.array(feat), **(extra or {}))


def time_r18(cq, tl):
    import torch
    from model.resnet.ResNet_18_fabu import ResNet18          # the REFERENCE's model file
    torch.set_num_threads(8)
    rec = {"config": "BASELINE configs[0]: ResNet_18_fabu, 256 synthetic 3x32x32 images = 2 batches of 128 "
                     "(MAX_CALI_IMG_NUM 1 -> batches 0..1), WORKER_NUM 4, INTERVAL_NUM 2048, CPU",
           "host": {"cpus": os.cpu_count(), "python": sys.version.split()[0], "numpy": np.__version__,
                    "torch": torch.__version__}}
    with _refenv.reference_workdir(input_shape="1,3,32,32", max_cali_img_num=1) as tmp:
        model = cq.merge_bn(cases.seed_model(ResNet18()).eval(), "cpu")
        t0 = time.perf_counter()
        q = tl.Quantity(model)
        rec["graph_discovery_s"] = round(time.perf_counter() - t0, 3)
        batches = cases.calib_batches(2, (128, 3, 32, 32))
        t0 = time.perf_counter()
        with _kl_spy(tl) as seen:                 # G12: the tables of this very run are kept, not only its wall time
            q.activation_quantize(batches)
        rec["activation_quantize_s"] = round(time.perf_counter() - t0, 3)
        feat = _read(os.path.join(tmp, "test", "workdir", "feat.table"))
        t0 = time.perf_counter()
        q.weight_quantize()
        rec["weight_quantize_and_rewrite_s"] = round(time.perf_counter() - t0, 3)
        wd = os.path.join(tmp, "test", "workdir")
        g12 = {"recipe": {"model": "ResNet_18_fabu (cases.seed_model, merge_bn)", "n_batches": 2, "shape": [128, 3, 32, 32], "seed": 0,
                          "max_cali_img_num": 1},
               "feat_table": feat, "weight_table": _read(os.path.join(wd, "weight.table")),
               "files": {d: _dir_state(os.path.join(wd, d)) for d in ("weight", "bias", "new_weight", "new_bias")}}
    rec["calibration_images_per_s"] = round(256 / rec["activation_quantize_s"], 3)
    _save_stats(os.path.join(HERE, "g12_r18_config1_stats.npz"), seen, feat)
    with open(os.path.join(HERE, "g12_r18_config1.json"), "w") as fh:
        json.dump(g12, fh, indent=1, sort_keys=True)
    print("G12 written: config 1 at its stated size,", seen["hist"].shape, "rows x bins,", int(seen["hist"][0].sum()), "image elements")
    with open(os.path.join(HERE, "ref_timing_r18.json"), "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1))


def time_r101_512(cq, tl):
    import torch
    sys.path.append(OURS)
    from model.resnet.ResNet_fabu import ResNet101
    torch.set_num_threads(8)
    rec = {"config": "BASELINE configs[4] shape: fabu ResNet-101 @3x512x512, ONE synthetic image (MAX_CALI_IMG_NUM 0), WORKER_NUM 4, "
                     "INTERVAL_NUM 2048, CPU; 139 histogram rows, 132.25 M cared elements",
           "host": {"cpus": os.cpu_count(), "python": sys.version.split()[0], "numpy": np.__version__, "torch": torch.__version__}}
    with _refenv.reference_workdir(input_shape="1,3,512,512", max_cali_img_num=0) as tmp:
        model = cq.merge_bn(cases.seed_model(ResNet101(input_size=512), gamma_scale=0.5).eval(), "cpu")
        t0 = time.perf_counter()
        q = tl.Quantity(model)
        rec["graph_discovery_s"] = round(time.perf_counter() - t0, 3)
        t0 = time.perf_counter()
        with _kl_spy(tl) as seen:                 # G13: the tables of this very run are kept, not only its wall time
            q.activation_quantize(cases.calib_batches(1, (1, 3, 512, 512), seed=512))
        rec["activation_quantize_s"] = round(time.perf_counter() - t0, 3)
        feat = _read(os.path.join(tmp, "test", "workdir", "feat.table"))
        # the BatchNorm fold factors as THIS machine's torch.sqrt evaluates them (cases.fold_bn_with_scales, see capture_scales)
        raw = cases.seed_model(ResNet101(input_size=512), gamma_scale=0.5).eval()
        scales = {"scale__" + name: (layer.weight.data / torch.sqrt(layer.running_var + 1e-5)).numpy().copy()
                  for name, layer in raw.named_modules() if type(layer).__name__ == "BatchNorm2d"}
    rec["calibration_images_per_s"] = round(1.0 / rec["activation_quantize_s"], 5)
    _save_stats(os.path.join(HERE, "g13_r101_512_stats.npz"), seen, feat, scales)
    print("G13 written: ResNet-101 @512^2, one image,", seen["hist"].shape, "rows x bins")
    with open(os.path.join(HERE, "ref_timing_r101_512.json"), "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1))


def main():
    which = sys.argv[1:] or ["h6", "r50"]
    cq, tl = _refenv.import_reference()
    if "h6" in which:
        capture_h6(cq, tl)
    if "scales" in which:
        capture_scales(cq, tl)
    if "time18" in which:
        time_r18(cq, tl)
    if "r50" in which:
        capture_r50(cq, tl)
    if "time101" in which:
        time_r101_512(cq, tl)
    if "r50stats" in which:
        capture_r50_stats(cq, tl)


if __name__ == "__main__":
    main()
