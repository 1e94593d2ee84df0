#!/usr/bin/env python3
"""GPU probe: per-step time of pass 1 (forward + absmax [+ keep activations]) to locate allocator stalls."""
import os, sys, time
if len(sys.argv) > 2 and sys.argv[2] == "expand":
    import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
import bench
from tools import Quantity
from common.quantity import DistributionCollector
keep_n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = 128
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(1, "1,3,224,224", 0)
q = Quantity(model)
sys.stdout = out
names = ["image"] + list(q.net_info.keys())
coll = DistributionCollector(names)
feats, hooks = q.regist_hook_outfeature(model)
xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(4)]
kept = []
print("alloc conf", os.environ.get("PYTORCH_HIP_ALLOC_CONF"), "keep", keep_n)
for step in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        model(xs[step % 4])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    coll.refresh_max_val(feats)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if len(kept) < keep_n:
        kept.append(dict(feats))
    print("step %2d fwd %7.1f ms absmax %6.2f ms reserved %6.1f GB allocated %6.1f GB" %
          (step, (t1 - t0) * 1e3, (t2 - t1) * 1e3, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30))
