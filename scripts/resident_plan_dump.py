#!/usr/bin/env python3
"""GPU probe: print the resident plan of the fabu ResNet-50 ReconModel (which producers still emit fp32)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
import bench
from tools import Quantity, Reconstruction
from common.quantity import resident
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(1, "1,3,224,224", 0)
data = bench.DeviceBatches(2, 8, 224, 0, 1, dev)
q = Quantity(model); q.activation_quantize(data); q.weight_quantize()
rec = Reconstruction(bench.build_model("r50", 224, dev))
net = rec.ReconModel(rec.get_quantity_information(), "./workdir/recon.pth")
sys.stdout = out
print(resident.enable(net, data[0][0]))
for name, plan in resident.describe(net).items():
    if plan.emit_f32 or name in ("conv1", "maxpool", "layer4.2.Eltwise"):
        print(name, plan)
print("residual adds: name, grid g (exact sum = S * 2^-g), narrow bit, fused ReLU")
for name, plan in resident.describe(net).items():
    if plan.resident_add:
        print("  %-22s g=%s narrow_bit=%s relu=%s want_wide=%s" % (name, plan.grid, plan.narrow_bit, plan.relu, plan.want_wide))
