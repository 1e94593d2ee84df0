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
from common.quantity import channel_collector as cc
phase = {}
def timed(name):
    orig = getattr(cc.ChannelCollector, name)
    def wrapper(self, *a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = orig(self, *a, **k)
        torch.cuda.synchronize(); phase[name] = phase.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(cc.ChannelCollector, name, wrapper)
if "--phases" in sys.argv:                            # synchronising timers around the collector's entry points
    for name in ("refresh_max_val", "add_to_distributions", "intervals", "quantize"):
        timed(name)
torch.cuda.synchronize(); t0 = time.perf_counter()
bits = q.activation_quantize_per_channel(data)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
sys.stdout = out
rows = q._channel_collector.rows
print("per-channel calibration: %d images, %d rows, %.3f s = %.0f images/s" % (K * B, rows, dt, K * B / dt))
if phase:
    print("  of which (synchronised): " + ", ".join("%s %.3f s" % kv for kv in phase.items()))
