#!/usr/bin/env python3
"""GPU probe: ReconModel (int8-sim) / ReconTest forward of the fabu ResNet-50 only, for rocprofv3 kernel stats."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
import bench
from tools import Quantity, Reconstruction
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(1, "1,3,224,224", 0)
data = bench.DeviceBatches(2, B, 224, 0, 1, dev)
q = Quantity(model); q.activation_quantize(data); q.weight_quantize()
rec = Reconstruction(bench.build_model("r50", 224, dev))
net = rec.ReconModel(rec.get_quantity_information(), "./workdir/recon.pth")
sys.stdout = out
x = data[0][0]
if len(sys.argv) > 3 and sys.argv[3] == "resident":
    from common.quantity import resident
    print("resident plan:", resident.enable(net, x))
with torch.no_grad():
    for _ in range(3): net(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(ITERS): net(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / ITERS
print("ReconModel B=%d: %.3f ms/batch = %.0f img/s" % (B, dt * 1e3, B / dt))
if len(sys.argv) > 4 and sys.argv[4] == "graph":
    g = resident.capture(net, x)
    for _ in range(3): g(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(ITERS): g(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / ITERS
    print("ReconModel B=%d HIP graph: %.3f ms/batch = %.0f img/s" % (B, dt * 1e3, B / dt))
if "hostprof" in sys.argv:
    # how long the host needs to ENQUEUE one forward (no sync inside the loop), and where it spends it
    import cProfile, pstats
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(ITERS): net(x)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("host enqueue %.3f ms/forward; drain after the loop %.3f ms" % ((t1 - t0) / ITERS * 1e3, (t2 - t1) * 1e3))
        pr = cProfile.Profile(); pr.enable()
        for _ in range(ITERS): net(x)
        pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
