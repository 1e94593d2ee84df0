"""GPU box, fresh process: what Quantity(model) -- the reference's build_net_structure trace -- costs a one-shot user, beside the first calibration."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
os.environ.setdefault("FQ_ACT_CACHE_GB", "0")
t_imp = time.perf_counter()
import bench
from tools import Quantity
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
t0 = time.perf_counter()
model = bench.build_model("r50", 224, dev)
torch.cuda.synchronize(); t1 = time.perf_counter()
bench.make_workdir(5119, "1,3,224,224", 0)
data = bench.DeviceBatches(20, 256, 224, 0, 1, dev)
torch.cuda.synchronize(); t2 = time.perf_counter()
q = Quantity(model)
torch.cuda.synchronize(); t3 = time.perf_counter()
q.activation_quantize(data)
torch.cuda.synchronize(); t4 = time.perf_counter()
q2 = Quantity(model)
torch.cuda.synchronize(); t5 = time.perf_counter()
sys.stdout = out
print("build_model %.3f s, synthetic data %.3f s, Quantity(model) %.3f s, first calibration %.3f s, a second Quantity(model) %.3f s" %
      (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
