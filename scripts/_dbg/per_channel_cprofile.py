"""GPU box: cProfile of a warm per-channel calibration (1 024 images): host time by function."""
import os, sys, time, cProfile, pstats, io, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
import bench
from tools import Quantity
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(3, "1,3,224,224", 0)
data = bench.DeviceBatches(4, 256, 224, 0, 1, dev)
for rep in range(2):
    q = Quantity(model); q.activation_quantize_per_channel(data)
q = Quantity(model)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
q.activation_quantize_per_channel(data)
torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
sys.stdout = out
print("per-channel, 1 024 images: %.4f s" % dt, q.timings)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:7000])
