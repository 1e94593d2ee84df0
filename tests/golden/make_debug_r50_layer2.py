#!/usr/bin/env python3
"""Debugging aid (not a fixture): per-module outputs of layer2.0 of the REFERENCE's ReconModel of the fabu ResNet-50, written
to gpurun_out/dbg_r50_layer2.npz (scratch, git-ignored), for scripts/_dbg/r50_layer2.py to compare with the GPU model module
by module.  This pair localised round 2's only model-level mismatch to one quantised weight of layer2.0.conv3 -- a one-ulp
difference in merge_bn's torch.sqrt between the build container's and the GPU box's CPU."""
import json, os, sys
import numpy as np
sys.path.insert(0, '/root/repo/tests/golden')
import cases, _refenv
cq, tl = _refenv.import_reference()
import torch
sys.path.append('/root/repo/pytorch-quantity_amd/quantity')
from model.resnet.ResNet_fabu import ResNet50
tables = json.load(open('/root/repo/tests/golden/g4_r50_tables.json'))
arrays = {}
with _refenv.reference_workdir(input_shape="1,3,224,224", max_cali_img_num=1) as tmp:
    wd = os.path.join(tmp, "test", "workdir"); os.makedirs(wd, exist_ok=True)
    open(os.path.join(wd, "feat.table"), "w").write(tables["feat_table"])
    open(os.path.join(wd, "weight.table"), "w").write(tables["weight_table"])
    x = cases.fixed_input((2, 3, 224, 224))
    rec = tl.Reconstruction(cases.seed_model(ResNet50(), gamma_scale=tables["gamma_scale"]).eval())
    rec.merge_bn()
    info = rec.get_quantity_information()
    recon = rec.ReconModel(info, os.path.join(wd, "recon.pth"))
    outs = {}
    hooks = []
    for name, m in recon.named_modules():
        if name.startswith("layer2.0.") and type(m).__name__ in ("NewConv2d", "NewAdd", "ReLU"):
            hooks.append(m.register_forward_hook(lambda mod, i, o, name=name: outs.__setitem__(name, o.detach().numpy().copy())))
    with torch.no_grad():
        logits = recon(x).numpy()
    assert np.array_equal(logits, np.load('/root/repo/tests/golden/g4_r50_recon.npz')["logits_recon"])
    for name, o in outs.items():
        key = name
        ob = info[name]["output_bit"] if name in info else None
        print(name, o.shape, ob, float(np.abs(o).max()))
        arrays[name] = o.astype(np.float16) if np.array_equal(o.astype(np.float16).astype(np.float32), o) else o
    # quantised weights / bias of the first two convs
    for n in ("layer2.0.conv1", "layer2.0.conv2", "layer2.0.downsample.0"):
        mod = dict(recon.named_modules())[n]
        arrays[n + ".qweight"] = mod.Conv.weight.detach().numpy().astype(np.int8)
        arrays[n + ".qbias"] = np.asarray(mod.quantized_bias).astype(np.float32)
        print(n, {k: v for k, v in info[n].items() if k not in ("layer",)})
os.makedirs('/root/repo/gpurun_dbg', exist_ok=True)
np.savez_compressed('/root/repo/gpurun_dbg/r50_layer2.npz', **arrays)
print(os.path.getsize('/root/repo/gpurun_dbg/r50_layer2.npz'))
