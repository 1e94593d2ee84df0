"""GPU box: how long the HOST needs to enqueue one calibration forward (pass 1 and pass 2) -- the same loop on 2-image batches, where
the device has almost nothing to do, beside the 256-image batches of the bench."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
import bench
from tools import Quantity
dev = torch.device("cuda", 0)
out = sys.stdout
sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
res = []
for B in (2, 256):
    K = 20
    bench.make_workdir(K * B - 1, "1,3,224,224", 0)
    data = bench.DeviceBatches(K, B, 224, 0, 1, dev)
    q = Quantity(model)
    q.profile_phases = True
    for _ in range(2):
        q.activation_quantize(data)
    torch.cuda.synchronize(); t0 = time.perf_counter(); q.activation_quantize(data); torch.cuda.synchronize()
    res.append((B, time.perf_counter() - t0, q.timings.get("pass1_s"), q.timings.get("pass2_s")))
sys.stdout = out
for B, dt, p1, p2 in res:
    print("batch %3d: %.3f s for 20 steps; pass 1 %.2f ms per step, pass 2 %.2f ms per step" % (B, dt, p1 / 20 * 1e3, p2 / 20 * 1e3))
