"""GPU box: does the keeper scan's reference-count pre-filter keep the heap pass away in a clean model -- in a process's first
calibration and in later ones?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
import bench
from tools import Quantity, _hook_state
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(0, "1,3,224,224", 0)
data = bench.DeviceBatches(1, 256, 224, 0, 1, dev)
calls = []
real = _hook_state._DeferralProbe.holders
def spy(tensors, ours):
    import sys as _s
    import gc, types
    refs = gc.get_referrers(*tensors)
    desc = []
    for rr in refs:
        if isinstance(rr, types.FrameType): desc.append("frame:" + rr.f_code.co_name); continue
        d = type(rr).__name__
        if isinstance(rr, dict): d += ":" + ",".join(str(k)[:30] for k in list(rr.keys())[:6])
        if isinstance(rr, (list, tuple)): d += ":len%d" % len(rr)
        owners = [type(o).__name__ + (":" + ",".join(str(kk)[:24] for kk in list(o.keys())[:5]) if isinstance(o, dict) else "") for o in gc.get_referrers(rr)[:4]]
        desc.append(d + " <- " + "|".join(owners))
    calls.append(desc)
    t0 = time.perf_counter(); r = real(tensors, ours); dt = time.perf_counter() - t0
    calls.append((len(tensors), [(_s.getrefcount(t), t._use_count(), tuple(t.shape)) for t in tensors[:4]], round(dt * 1e3, 1), dict((k, v) for k, v in r.items())))
    return r
_hook_state._DeferralProbe.holders = staticmethod(spy)
for rep in range(3):
    q = Quantity(model); q.activation_quantize(data)
    calls.append("--- end of calibration %d" % rep)
sys.stdout = out
for c in calls: print(c)
