"""GPU box, fresh process: where Quantity(model)'s 0.25 s goes (cProfile, synchronising after every hooked module)."""
import cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
import bench
from tools import Quantity
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(5119, "1,3,224,224", 0)
torch.cuda.synchronize()
stamps = []
def stamp(m, i, o):
    torch.cuda.synchronize()
    stamps.append((time.perf_counter(), type(m).__name__))
hs = [m.register_forward_hook(stamp) for m in model.modules() if not list(m.children())]
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
q = Quantity(model)
torch.cuda.synchronize(); pr.disable(); t1 = time.perf_counter()
for h in hs:
    h.remove()
sys.stdout = out
print("Quantity(model) %.3f s" % (t1 - t0))
prev = t0
for t, name in stamps[:12]:
    print("  %-12s +%.1f ms" % (name, (t - prev) * 1e3)); prev = t
print("  ... the other %d modules %.1f ms; after the forward %.1f ms" % (len(stamps) - 12, (stamps[-1][0] - prev) * 1e3, (t1 - stamps[-1][0]) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
