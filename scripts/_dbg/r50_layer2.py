import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases
from workdir_util import product_workdir
from model.resnet.ResNet_fabu import ResNet50
from tools import Reconstruction
ref = np.load(os.path.join(ROOT, "gpurun_dbg", "r50_layer2.npz"))
tables = json.load(open(os.path.join(ROOT, "tests", "golden", "g4_r50_tables.json")))
x = cases.fixed_input((2, 3, 224, 224)).cuda()
with product_workdir(input_shape="1,3,224,224", device="gpu") as tmp:
    wd = os.path.join(tmp, "test", "workdir"); os.makedirs(wd, exist_ok=True)
    open(os.path.join(wd, "feat.table"), "w").write(tables["feat_table"])
    open(os.path.join(wd, "weight.table"), "w").write(tables["weight_table"])
    sys.stdout = open(os.devnull, "w")
    rec = Reconstruction(cases.seed_model(ResNet50(), gamma_scale=tables["gamma_scale"]).eval())
    rec.merge_bn()
    info = rec.get_quantity_information()
    net = rec.ReconModel(info, os.path.join(wd, "recon.pth")).cuda()
    sys.stdout = sys.__stdout__
    outs, ins = {}, {}
    mods = dict(net.named_modules())
    for name, m in mods.items():
        if name.startswith("layer2.0.") and type(m).__name__ in ("NewConv2d", "NewAdd", "ReLU"):
            def hook(mod, i, o, name=name):
                outs[name] = o.detach().clone()
                ins[name] = [t.detach().clone() for t in i]
            m.register_forward_hook(hook)
    with torch.no_grad():
        net(x)
    for name in ref.files:
        if name.endswith(".qweight") or name.endswith(".qbias"):
            continue
        r = ref[name].astype(np.float32)
        g = outs[name].cpu().numpy()
        bad = g != r
        print("%-24s mismatches %7d / %d  max|d| %.5f" % (name, bad.sum(), r.size, np.abs(g - r).max()))
    for n in ("layer2.0.conv1", "layer2.0.conv2", "layer2.0.downsample.0"):
        m = mods[n]
        print(n, "qweight equal", np.array_equal(m.Conv.weight.detach().cpu().numpy().astype(np.int8), ref[n + ".qweight"]),
              "qbias equal", np.array_equal(m.quantized_bias.cpu().numpy(), ref[n + ".qbias"]))
    # conv1 alone on the reference's input is not available; use module-by-module: feed reference outputs forward
    with torch.no_grad():
        r1 = torch.from_numpy(ref["layer2.0.relu1"].astype(np.float32)).cuda()
        c2 = mods["layer2.0.conv2"](r1).cpu().numpy()
        bad = c2 != ref["layer2.0.conv2"].astype(np.float32)
        print("conv2 on the reference relu1 output: mismatches", bad.sum(), "of", bad.size)
        if bad.any():
            idx = np.argwhere(bad)[:10]
            for i in idx:
                print("  at", tuple(i), "got", c2[tuple(i)], "ref", ref["layer2.0.conv2"][tuple(i)])
            print("  mismatching channels:", np.unique(np.argwhere(bad)[:, 1])[:40], " rows:", np.unique(np.argwhere(bad)[:, 2]), " cols:", np.unique(np.argwhere(bad)[:, 3]))
            # exact integer conv through the fp32 fallback
            m = mods["layer2.0.conv2"]
            m.use_int8_mfma = False
            c2f = m(r1).cpu().numpy()
            m.use_int8_mfma = True
            print("  fp32-conv fallback vs reference:", (c2f != ref["layer2.0.conv2"].astype(np.float32)).sum(), " vs int8 path:", (c2f != c2).sum())
        r2 = torch.from_numpy(ref["layer2.0.relu2"].astype(np.float32)).cuda()
        m = mods["layer2.0.conv3"]
        want = ref["layer2.0.conv3"].astype(np.float32)
        c3 = m(r2).cpu().numpy()
        bad = c3 != want
        print("conv3 on the reference relu2 output: mismatches", bad.sum(), {k: getattr(m, k) for k in ("weight_bit", "input_bit", "output_bit", "rs_bit", "bias_bit")})
        m.use_int8_mfma = False
        c3f = m(r2).cpu().numpy()
        m.use_int8_mfma = True
        print("  fp32 fallback vs ref:", (c3f != want).sum(), " int8 vs fallback:", (c3 != c3f).sum())
        q = m.Quan(r2)
        acc = m.Conv(q)                       # integer-valued fp32 accumulators
        accn = acc.cpu().numpy()
        qb = m.quantized_bias.cpu().numpy()
        idx = np.argwhere(bad)
        print("  channels:", np.unique(idx[:, 1]))
        for i in idx[:12]:
            i = tuple(i)
            a = accn[i]
            print("   at", i, "got", c3[i], "ref", want[i], "acc", a, "acc/2^rs", a / 2.0 ** m.rs_bit, "qbias", qb[i[1]],
                  "chain:", np.clip(np.trunc(a / 2.0 ** m.rs_bit + (0.5 if a > 0 else -0.5)), -128, 127) + qb[i[1]])
        print("  qbias range", qb.min(), qb.max(), " float bias of those channels", m.bias.detach().cpu().numpy()[np.unique(idx[:, 1])][:8],
              " bias*2^bit", (m.bias.detach().cpu().numpy() * 2.0 ** m.bias_bit)[np.unique(idx[:, 1])][:8])
