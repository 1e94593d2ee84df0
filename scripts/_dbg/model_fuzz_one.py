"""GPU box: one model of scripts/model_fuzz.py in detail.  usage: model_fuzz_one.py <index> <seed>"""
import importlib.util, os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
mf = importlib.util.module_from_spec(spec); spec.loader.exec_module(mf)
i, seed = int(sys.argv[1]), int(sys.argv[2])
odd = "odd" in sys.argv[3:]
torch.backends.cudnn.deterministic = odd
model, size, bs, rng = mf.random_net(i, seed, odd, "cuda")
batches = [(torch.randn(bs, model.cin, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(3)]
print("size", size, "batch", bs)
for step in model.plan:
    m = getattr(model, step[1])
    print("  %-4s %-50s %s -> %s" % (step[1], str(m)[:50], step[2], step[3]))
a = mf.calibrate(model, size, batches)
off = tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else mf.SWITCHES
b = mf.calibrate(model, size, batches, off=off)
print("second calibration without", off)
names = a[5]
for r, k in enumerate(names):
    ha, hb = a[2][r].double(), b[2][r].double()
    flag = ("" if a[0][k] == b[0][k] else "   <-- bits differ") + ("" if torch.equal(ha, hb) and a[1][k] == b[1][k] else "  *")
    print("%-12s bits %3s %3s  max %.8g %.8g  hist total %d %d  moved %d%s" % (k, a[0][k], b[0][k], a[1][k], b[1][k], int(ha.sum()), int(hb.sum()), int((ha - hb).abs().sum()), flag))
print({k: v for k, v in a[4].items() if "proven" in k or "launches" in k or "refused" in k})
print({k: v for k, v in b[4].items() if "proven" in k or "launches" in k or "refused" in k})
