import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases
from common.quantity import merge_bn, _native
from model.resnet.ResNet_18_fabu import ResNet18
from oracle import fq_oracle as orc
sys.stdout = open(os.devnull, "w")
m = merge_bn(cases.seed_model(ResNet18()).eval())
sys.stdout = sys.__stdout__
for name, p in m.named_parameters():
    w = p.detach().numpy()
    mx = orc.absmax(w); bit = orc.bits_from_absmax(mx)
    ref = orc.quantize_param_i32(w, bit)
    got = _native.quantize_param_i32(p.detach().cuda(), bit).cpu().numpy()
    bad = np.argwhere(ref != got)
    if len(bad):
        i = tuple(bad[0]); print(name, "bit", bit, "nbad", len(bad), "w", repr(w[i]), "w*2^bit", repr(np.float32(w[i]) * np.float32(2.0**bit)), "ref", ref[i], "got", got[i])
print("checked")
