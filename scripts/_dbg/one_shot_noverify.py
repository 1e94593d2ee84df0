"""GPU box, fresh process: the process's FIRST calibration with the once-per-module checks switched off -- what they cost a one-shot user."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
os.environ.setdefault("FQ_ACT_CACHE_GB", "0")
import bench
from tools import Quantity
from common.quantity import _float_conv
if len(sys.argv) > 1 and sys.argv[1] == "off":
    def fake(m, run, x, k=None):
        _float_conv.state(m).setdefault("verified", set()).add(_float_conv.kernel_key(m, k))
        return None
    _float_conv.verified = fake
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(5119, "1,3,224,224", 0)
data = bench.DeviceBatches(20, 256, 224, 0, 1, dev)
q = Quantity(model)
torch.cuda.synchronize(); t0 = time.perf_counter()
q.activation_quantize(data)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
t1 = time.perf_counter(); q.activation_quantize(data); torch.cuda.synchronize(); dt2 = time.perf_counter() - t1
sys.stdout = out
print("checks %s: first calibration %.3f s, second %.3f s, max reserved %.1f GB" % (sys.argv[1] if len(sys.argv) > 1 else "on", dt, dt2, torch.cuda.max_memory_reserved() / 2 ** 30))
