#!/usr/bin/env python3
"""GPU: how far the fake-quant models (ReconTest) are from the reference's CPU logits (goldens G4-R50, G4-R18)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases
from workdir_util import product_workdir
from tools import Reconstruction
G = os.path.join(ROOT, "tests", "golden")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
tables = json.load(open(os.path.join(G, "g4_r50_tables.json"))); g4 = np.load(os.path.join(G, "g4_r50_recon.npz"))
from model.resnet.ResNet_fabu import ResNet50
scales = np.load(os.path.join(G, "g4_r50_bn_scales.npz"))
x = cases.fixed_input(tuple(tables["input"]["shape"]), seed=tables["input"]["seed"]).cuda()
with product_workdir(input_shape="1,3,224,224", device="gpu") as tmp:
    wd = os.path.join(tmp, "test", "workdir"); os.makedirs(wd, exist_ok=True)
    open(os.path.join(wd, "feat.table"), "w").write(tables["feat_table"]); open(os.path.join(wd, "weight.table"), "w").write(tables["weight_table"])
    rec = Reconstruction(cases.fold_bn_with_scales(cases.seed_model(ResNet50(), gamma_scale=tables["gamma_scale"]).eval(), scales)); rec.merge_bn()
    info = rec.get_quantity_information()
    net = rec.ReconTest(info, os.path.join(wd, "recontest.pth")).cuda()
    with torch.no_grad():
        c1 = net.conv1(x).cpu().numpy(); logits = net(x).cpu().numpy()
sys.stdout = out
s = c1[:, :8, ::8, ::8]; ref = g4["recontest_conv1_out_sample"]
step1 = 2.0 ** -info["conv1"]["output_bit"]; step = 2.0 ** -info["fc"]["output_bit"]
print("R50 conv1 sample: identical fraction %.6f (%d of %d differ), max |diff| / step %.3f" % (np.mean(s == ref), int((s != ref).sum()), s.size, np.max(np.abs(s - ref)) / step1))
d = np.abs(logits - g4["logits_recontest"]) / step
print("R50 logits: max |diff| %.3f steps, mean %.4f steps, fraction identical %.4f, argmax equal %s" % (d.max(), d.mean(), np.mean(d == 0), np.array_equal(logits.argmax(1), g4["logits_recontest"].argmax(1))))
sys.stdout = open(os.devnull, "w")
g3 = json.load(open(os.path.join(G, "g3_r18_e2e.json"))); g418 = np.load(os.path.join(G, "g4_r18_recon.npz"))
from model.resnet.ResNet_18_fabu import ResNet18
with product_workdir(device="gpu") as tmp:
    wd = os.path.join(tmp, "test", "workdir"); os.makedirs(wd, exist_ok=True)
    open(os.path.join(wd, "feat.table"), "w").write(g3["feat_table"]); open(os.path.join(wd, "weight.table"), "w").write(g3["weight_table_after_second_rewrite"])
    rec = Reconstruction(cases.seed_model(ResNet18()).eval()); rec.merge_bn()
    net = rec.ReconTest(rec.get_quantity_information(), os.path.join(wd, "recontest.pth")).cuda()
    xx = torch.from_numpy(g418["x"]).cuda()
    with torch.no_grad():
        first = net.conv1[0](xx).cpu().numpy(); lg = net(xx).cpu().numpy()
sys.stdout = out
print("R18 first layer: identical fraction %.6f (%d of %d differ), max |diff| %.4f (step 0.125)" % (np.mean(first == g418["recontest_conv1_out"]), int((first != g418["recontest_conv1_out"]).sum()), first.size, np.max(np.abs(first - g418["recontest_conv1_out"]))))
print("R18 logits: max |diff| %.3f (step 1.0), identical fraction %.4f" % (np.max(np.abs(lg - g418["logits_recontest"])), np.mean(lg == g418["logits_recontest"])))
