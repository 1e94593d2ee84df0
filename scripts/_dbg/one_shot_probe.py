"""GPU box, fresh process: where the time of a process's FIRST calibration goes (bench.py's one_shot): the once-per-module checks
(synchronised and timed here), the probe forwards, everything else."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
import bench
from common.quantity import _float_conv, _native
from tools import Quantity, pytorch_quantizer as pq

dev = torch.device("cuda", 0)
sys.stdout, out = open(os.devnull, "w"), sys.stdout
_native.lib()
model = bench.build_model("r50", 224, dev)
data = bench.DeviceBatches(20, 256, 224, 0, 1, dev)
bench.make_workdir(19, "1,3,224,224", 0)
acc = {"verified": 0.0, "n_verified": 0, "probe": 0.0, "first_matmul": None}
real_v = _float_conv.verified
def timed_v(m, run, x):
    if _float_conv.is_verified(m):
        return real_v(m, run, x)
    torch.cuda.synchronize(); t = time.perf_counter()
    r = real_v(m, run, x)
    torch.cuda.synchronize(); d = time.perf_counter() - t
    acc["verified"] += d; acc["n_verified"] += 1
    if acc["first_matmul"] is None: acc["first_matmul"] = d
    return r
_float_conv.verified = timed_v
real_p = Quantity._probe_forward
def timed_p(self, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = real_p(self, *a, **k)
    torch.cuda.synchronize(); acc["probe"] += time.perf_counter() - t
    return r
Quantity._probe_forward = timed_p
q = Quantity(model)
torch.cuda.synchronize(); t0 = time.perf_counter()
q.activation_quantize(data)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
sys.stdout = out
print("first calibration %.3f s: module checks %.3f s (%d modules; the first one %.3f s), probe forwards %.3f s, pass1 %.3f pass2 %.3f"
      % (dt, acc["verified"], acc["n_verified"], acc["first_matmul"], acc["probe"], q.timings["pass1_s"], q.timings["pass2_s"]))
bench.make_workdir(19, "1,3,224,224", 0)
q = Quantity(model)
torch.cuda.synchronize(); t0 = time.perf_counter()
q.activation_quantize(data)
torch.cuda.synchronize(); print("second calibration %.3f s" % (time.perf_counter() - t0))
