import hashlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases
from model.resnet.ResNet_fabu import ResNet50
from common.quantity import merge_bn
sys.stdout = open(os.devnull, "w")
m = cases.seed_model(ResNet50(), gamma_scale=0.5).eval()
sd = {k: hashlib.sha256(v.numpy().tobytes()).hexdigest()[:16] for k, v in m.state_dict().items() if v.is_floating_point()}
mm = merge_bn(m)
sd2 = {"merged." + k: hashlib.sha256(v.numpy().tobytes()).hexdigest()[:16] for k, v in mm.state_dict().items() if v.is_floating_point()}
w = mm.layer2[0].conv3.weight.detach()
sd2["q.layer2.0.conv3"] = hashlib.sha256(torch.round(w * 256).clamp(-128, 127).numpy().tobytes()).hexdigest()[:16]
sd.update(sd2)
# raw generator probes
g = np.random.default_rng(12345)
sd["probe.normal"] = hashlib.sha256(g.standard_normal(5_000_000, dtype=np.float32).tobytes()).hexdigest()[:16]
sd["probe.random"] = hashlib.sha256(g.random(5_000_000, dtype=np.float32).tobytes()).hexdigest()[:16]
sys.stdout = sys.__stdout__
out = os.path.join(ROOT, "gpurun_dbg", "seed.json")
if len(sys.argv) > 1 and sys.argv[1] == "write":
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(sd, open(out, "w"))
    print("written", len(sd))
else:
    ref = json.load(open(out))
    bad = [k for k in ref if ref[k] != sd.get(k)]
    print("differing tensors:", len(bad), bad[:20])
    import platform
    print(platform.processor(), open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0])
