#!/usr/bin/env python3
"""GPU probe: per-channel KL calibration (extension) of the fabu ResNet-50 at batch 128: images/s end to end."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
import bench
from tools import Quantity
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(K - 1, "1,3,224,224", 0)
data = bench.DeviceBatches(K, B, 224, 0, 1, dev)
q = Quantity(model)
q.activation_quantize_per_channel(data)              # warm-up (MIOpen, code load)
torch.cuda.synchronize(); t0 = time.perf_counter()
bits = q.activation_quantize_per_channel(data)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
sys.stdout = out
rows = q._channel_collector.rows
print("per-channel calibration: %d images, %d rows, %.3f s = %.0f images/s" % (K * B, rows, dt, K * B / dt))
