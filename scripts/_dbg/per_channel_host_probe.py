"""GPU box: cProfile of one warm per-channel calibration (1 024 images): what the host does beside the kernels."""
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
big = torch.empty(int(80e9), dtype=torch.uint8, device=dev); del big          # a warm pool, as in the bench
q = Quantity(model)
for _ in range(2):
    q.activation_quantize_per_channel(data)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
q.activation_quantize_per_channel(data)
torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
sys.stdout = out
print("per-channel calibration %.3f s" % dt, q.timings)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
