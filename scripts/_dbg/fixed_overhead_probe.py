"""GPU box: the fixed part of one activation_quantize call (probes, patching, KL, table) -- seconds against the number of batches."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
import bench
from tools import Quantity
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
res = []
for K in (20, 2, 5, 10, 20, 1):
    bench.make_workdir(K - 1, "1,3,224,224", 0)
    data = bench.DeviceBatches(K, 256, 224, 0, 1, dev)
    for rep in range(2):
        q = Quantity(model); q.profile_phases = True
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q.activation_quantize(data)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res.append((K, dt, q.timings.get("pass1_s"), q.timings.get("pass2_s"), q.timings.get("kl_s")))
sys.stdout = out
for r in res:
    print("K=%2d  total %.4f s  pass1 %.4f  pass2 %.4f  kl %.4f   per batch %.2f ms" % (r[0], r[1], r[2], r[3], r[4], r[1] / r[0] * 1e3))
