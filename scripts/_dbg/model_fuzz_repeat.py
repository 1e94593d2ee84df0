"""GPU box: one model of scripts/model_fuzz.py calibrated five times with everything on: which rows differ from run to run?"""
import importlib.util, os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
mf = importlib.util.module_from_spec(spec); spec.loader.exec_module(mf)
i, seed = int(sys.argv[1]), int(sys.argv[2])
odd = False
model, size, bs, rng = mf.random_net(i, seed, odd, "cuda")
batches = [(torch.randn(bs, model.cin, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(3)]
off = tuple(v for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else []) if v)
print("switched off:", off)
for step in model.plan:
    print("  %-4s %-60s %s -> %s" % (step[1], str(getattr(model, step[1]))[:60], step[2], step[3]))
runs = [mf.calibrate(model, size, batches, off=off) for _ in range(5)]
from tools import Quantity
names = None
for r in range(1, 5):
    rows = ["%s (%d of %d moved)" % (runs[0][5][j], int((runs[0][2][j] - runs[r][2][j]).abs().sum()), int(runs[0][2][j].sum()))
            for j in range(runs[0][2].shape[0]) if not torch.equal(runs[0][2][j], runs[r][2][j])]
    print("run %d against run 0: histogram rows that differ %s, maxima equal %s" % (r, rows, runs[0][1] == runs[r][1]))
