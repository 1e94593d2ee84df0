import hashlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases
from model.resnet.ResNet_fabu import ResNet50
sys.stdout = open(os.devnull, "w")
m = cases.seed_model(ResNet50(), gamma_scale=0.5).eval()
sys.stdout = sys.__stdout__
h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
out = {}
for name in ("bn1", "layer1.0.bn1", "layer2.0.bn3"):
    bn = dict(m.named_modules())[name]
    conv = dict(m.named_modules())[name.replace("bn", "conv")]
    g, v, mu, be = bn.weight.data, bn.running_var, bn.running_mean, bn.bias.data
    w = conv.weight.data
    t_add = v + 1e-5
    t_sqrt = torch.sqrt(t_add)
    t_scale = g / t_sqrt
    n_add = (v.numpy() + np.float32(1e-5)).astype(np.float32)
    n_sqrt = np.sqrt(n_add)
    n_scale = g.numpy() / n_sqrt
    t_w = t_scale.view(-1, 1, 1, 1) * w
    n_w = n_scale.reshape(-1, 1, 1, 1) * w.numpy()
    t_b = t_scale * (torch.zeros_like(mu) - mu) + be
    n_b = n_scale * (np.zeros_like(mu.numpy()) - mu.numpy()) + be.numpy()
    out[name] = {"t_add": h(t_add.numpy()), "n_add": h(n_add), "t_sqrt": h(t_sqrt.numpy()), "n_sqrt": h(n_sqrt),
                 "t_scale": h(t_scale.numpy()), "n_scale": h(n_scale), "t_w": h(t_w.numpy()), "n_w": h(n_w),
                 "t_b": h(t_b.numpy()), "n_b": h(n_b),
                 "torch==numpy": [bool(np.array_equal(t_add.numpy(), n_add)), bool(np.array_equal(t_sqrt.numpy(), n_sqrt)),
                                  bool(np.array_equal(t_scale.numpy(), n_scale)), bool(np.array_equal(t_w.numpy(), n_w)),
                                  bool(np.array_equal(t_b.numpy(), n_b))]}
    # 1e-5 as python float added to a float32 tensor: torch computes in float32? (v + 1e-5)
    out[name]["add_double_then_round"] = h((v.numpy().astype(np.float64) + 1e-5).astype(np.float32))
path = os.path.join(ROOT, "gpurun_dbg", "merge.json")
if len(sys.argv) > 1 and sys.argv[1] == "write":
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=0)[:1500])
else:
    ref = json.load(open(path))
    for name in out:
        print(name, "differs from the build container in:", [k for k in out[name] if out[name][k] != ref[name][k]], out[name]["torch==numpy"])
    print(torch.__config__.show().split("CPU capability")[1][:30], torch.get_num_threads())
