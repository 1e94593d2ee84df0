"""GPU box, fresh process: cProfile of the process's FIRST calibration (5 120 images): where the one-shot user's extra time goes."""
import os, sys, time, cProfile, pstats, io, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
os.environ.setdefault("FQ_ACT_CACHE_GB", "0")
import bench
from tools import Quantity
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(5119, "1,3,224,224", 0)
data = bench.DeviceBatches(20, 256, 224, 0, 1, dev)
q = Quantity(model)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
q.activation_quantize(data)
torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
t1 = time.perf_counter(); q.activation_quantize(data); torch.cuda.synchronize(); dt2 = time.perf_counter() - t1
sys.stdout = out
print("first calibration %.3f s, second %.3f s" % (dt, dt2), {k: v for k, v in q.timings.items() if k.endswith("_s")})
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:8000])
