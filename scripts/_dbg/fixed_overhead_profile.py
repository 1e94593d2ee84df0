"""GPU box: cProfile of a WARM process's one-batch calibration: where the fixed ~25 ms of a call go."""
import os, sys, time, cProfile, pstats, io, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
import bench
from tools import Quantity
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(0, "1,3,224,224", 0)
data = bench.DeviceBatches(1, 256, 224, 0, 1, dev)
for rep in range(3):
    q = Quantity(model); q.activation_quantize(data)
q = Quantity(model)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
q.activation_quantize(data)
torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
sys.stdout = out
print("one batch: %.4f s" % dt, {k: round(v, 4) for k, v in q.timings.items() if k.endswith("_s")})
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:9000])
