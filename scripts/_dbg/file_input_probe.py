"""GPU box: the file-input calibration (PRE_PROCESS.IMG = 2) twice in one process beside the tensor-input one -- what the first
run pays for its pinned staging ring (page-locking 6 x 154 MB), and where the main thread waits."""
import os, sys, time, tempfile, shutil
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
import bench
from tools import Quantity

dev = torch.device("cuda", 0)
K, B = 20, 256
out = sys.stdout
sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
wd = bench.make_workdir(K * B - 1, "1,3,224,224", 0)
data = bench.DeviceBatches(K, B, 224, 0, 1, dev)
q = Quantity(model)
for _ in range(2):
    q.activation_quantize(data)
torch.cuda.synchronize(); t0 = time.perf_counter(); q.activation_quantize(data); torch.cuda.synchronize()
t_tensor = time.perf_counter() - t0
fdir = tempfile.mkdtemp(prefix="fq_probe_npy_")
paths = []
for bi, xb in enumerate(data.owned()):
    host = xb.cpu().numpy()
    for j in range(host.shape[0]):
        paths.append(os.path.join(fdir, "img_%05d.npy" % (bi * B + j)))
        np.save(paths[-1], host[j])
ucfg_path = os.path.join(wd, "test", "user_configs.yml")
ucfg = yaml.safe_load(open(ucfg_path)); ucfg["PRE_PROCESS"]["IMG"] = 2
yaml.safe_dump(ucfg, open(ucfg_path, "w"))
t = time.perf_counter(); x = torch.empty(B * 3 * 224 * 224, dtype=torch.float32, pin_memory=True); t_pin = time.perf_counter() - t
del x
res = []
for rep in range(3):
    fq = Quantity(model); fq.file_batch = B; fq.profile_phases = True
    if "--steps" in sys.argv:
        os.environ["FQ_DEBUG_STEP_TIMES"] = "1"        # (synchronises every step: no upload can overlap)
    torch.cuda.synchronize(); t0 = time.perf_counter(); fq.activation_quantize(paths); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res.append((dt, fq.timings.get("pass1_s"), fq.timings.get("pass2_s"), dict(getattr(fq, "input_wait_s", {})), fq.timings.get("pass1_step_ms")))
sys.stdout = out
print("tensor inputs: %.3f s; pinning one 154 MB buffer: %.1f ms" % (t_tensor, t_pin * 1e3))
for dt, p1, p2, w, st in res:
    print("file inputs: %.3f s (%.0f images/s, %.3f x)  pass1 %.3f pass2 %.3f  waits %s" % (dt, K * B / dt, t_tensor / dt, p1, p2, {k: round(v, 3) for k, v in w.items()}))
    print("   pass-1 step ms:", st)
shutil.rmtree(fdir, ignore_errors=True)
