"""GPU box: wall time of one pass-1 forward (events around the whole forward, no per-kernel events) against the sum of its
kernels' durations (scripts/float_forward_table.py) -- what the launch gaps and the un-instrumented ops cost."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
os.environ.setdefault("FQ_ACT_CACHE_GB", os.environ.get("GAP_CACHE_GB", "0"))
import bench
from tools import Quantity

dev = torch.device("cuda", 0)
sys.stdout, out = open(os.devnull, "w"), sys.stdout
model = bench.build_model("r50", 224, dev)
bench.make_workdir(7, "1,3,224,224", 0)
data = bench.DeviceBatches(8, 256, 224, 0, 1, dev)
q = Quantity(model)
real = q._forward_with_stats
ev = []
def fws(item, fn, feats, extra=None):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = real(item, fn, feats, extra)
    b.record()
    ev.append((a, b))
    return r
q._forward_with_stats = fws
q.activation_quantize(data)
sys.stdout = out
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
print("pass 1 forwards (ms):", " ".join("%.2f" % v for v in ms[:8]))
print("pass 2 forwards (ms):", " ".join("%.2f" % v for v in ms[8:]))
print(q.timings["pass1_s"], q.timings["pass2_s"])
