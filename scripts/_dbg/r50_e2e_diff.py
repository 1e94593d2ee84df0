#!/usr/bin/env python3
"""GPU: calibrate the seeded ResNet-50 exactly as tests/golden/make_golden_r50.py did with the reference, print how the
statistics and the table differ from the reference's (g4_r50_tables.json / g4_r50_calib_stats.npz)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases
from workdir_util import product_workdir
G = os.path.join(ROOT, "tests", "golden")
tables = json.load(open(os.path.join(G, "g4_r50_tables.json")))
stats = np.load(os.path.join(G, "g4_r50_calib_stats.npz"))
from model.resnet.ResNet_fabu import ResNet50
scales = np.load(os.path.join(G, "g4_r50_bn_scales.npz"))
model = cases.fold_bn_with_scales(cases.seed_model(ResNet50(), gamma_scale=tables["gamma_scale"]).eval(), scales).cuda()
from tools import Quantity
from common.quantity import Quantizer
seen = {}
class Spy(Quantizer):
    def quantize(self, distributions, distribution_intervals):
        seen["names"] = list(distribution_intervals.keys())
        seen["interval"] = np.array([float(distribution_intervals[k]) for k in seen["names"]])
        seen["dist"] = distributions
        return super().quantize(distributions, distribution_intervals)
class Q(Quantity):
    quantizer_cls = Spy
with product_workdir(input_shape="1,3,224,224", device="gpu", max_cali_img_num=1) as tmp:
    q = Q(model)
    bits = q.activation_quantize(cases.calib_batches(2, (2, 3, 224, 224), seed=77))
    feat = open(os.path.join(tmp, "test", "workdir", "feat.table")).read()
    d = seen["dist"]
    print("type of distributions handed to quantize:", type(d))
    hist = d.cpu().numpy() if torch.is_tensor(d) else np.stack([np.asarray(d[k].cpu() if torch.is_tensor(d[k]) else d[k]) for k in seen["names"]])
ref_lines, got_lines = tables["feat_table"].strip().split("\n"), feat.strip().split("\n")
print("rows", len(ref_lines), len(got_lines), "identical table:", feat == tables["feat_table"])
for a, b in zip(ref_lines, got_lines):
    if a != b: print("  ref:", a, "| got:", b)
print("ref names == ours:", list(stats["names"]) == seen["names"])
ri, gi = stats["interval"], seen["interval"]
print("interval: rows that differ", int((ri != gi).sum()), "max rel diff", float(np.max(np.abs(ri - gi) / ri)))
rh = stats["hist"]
print("hist shapes", rh.shape, hist.shape, "row sums equal:", np.array_equal(rh.sum(1), hist.sum(1)))
l1 = np.abs(rh - hist).sum(1) / np.maximum(rh.sum(1), 1)
print("histogram L1 distance / elements: max %.3e  rows > 0: %d" % (l1.max(), int((l1 > 0).sum())))
order = np.argsort(-l1)[:8]
for i in order: print("   row %2d %-22s L1/n %.3e  interval rel diff %.3e  bits ref %d" % (i, seen["names"][i], l1[i], abs(ri[i] - gi[i]) / ri[i], stats["bits"][i]))
