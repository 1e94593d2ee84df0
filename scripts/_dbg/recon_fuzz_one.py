"""GPU box: one model of scripts/recon_fuzz.py: does the resident plan reproduce the fp32-boundary logits -- with everything, without
the block tail, without the projection in it, with the kernel variants forced (environment, read per call)?  Prints the model.
usage: recon_fuzz_one.py <index> <seed>"""
import importlib.util, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("recon_fuzz", os.path.join(ROOT, "scripts", "recon_fuzz.py"))
rf = importlib.util.module_from_spec(spec); spec.loader.exec_module(rf)
from common.quantity import resident, _native
from tools import Quantity, Reconstruction
from workdir_util import product_workdir
i, seed = int(sys.argv[1]), int(sys.argv[2])
model, size, bs = rf.build(i, seed)
data = [(torch.randn(bs, model.cin, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(2)]
print("size", size, "batch", bs)
for step in model.plan:
    print("  %-4s %-66s %s -> %s" % (step[1], str(getattr(model, step[1]))[:66], step[2], step[3]))
out = sys.stdout
with product_workdir(input_shape="1,%d,%d,%d" % (model.cin, size, size), device="gpu", max_cali_img_num=1):
    sys.stdout = open(os.devnull, "w")
    q = Quantity(model); q.activation_quantize(data); q.weight_quantize(); q.rewrite_weight()
    twin, _s, _r = rf.build(i, seed)
    rec = Reconstruction(twin)
    net = rec.ReconModel(rec.get_quantity_information(), "./workdir/recon.pth")
    sys.stdout = out
    x = data[0][0]
    with torch.no_grad():
        plain = net(x)
    for env in ({}, {"FQ_BLOCK_TAIL_PROJ": "0"}, {"FQ_BLOCK_TAIL": "0"}):
        for k in ("FQ_BLOCK_TAIL_PROJ", "FQ_BLOCK_TAIL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            with torch.no_grad():
                s = resident.enable(net, x, verify=False)
                _native.conv_variant_log = {}
                got = net(x)
                log, _native.conv_variant_log = _native.conv_variant_log, None
            bad = (got != plain)
            print(env or "default", "->", "equal" if not bool(bad.any()) else "DIFFERENT in %d of %d logits (max %.4g)" % (int(bad.sum()), bad.numel(), float((got - plain).abs().max())),
                  {k: v for k, v in s.items() if isinstance(v, int) and v}, log)
            if bool(bad.any()):
                # which module's output differs first: every leaf module's output against its fp32-boundary value
                def record(store):
                    hs = []
                    for name, m in net.named_modules():
                        if list(m.children()) and type(m).__name__ not in ("NewConv2d", "NewAdd", "NewLinear", "NewConcat"):
                            continue
                        def hook(mod, inp, o, name=name):
                            t = o.to_f32() if hasattr(o, "to_f32") else o
                            if torch.is_tensor(t):
                                store.append((name, type(mod).__name__, t.detach().clone()))
                        hs.append(m.register_forward_hook(hook))
                    return hs
                res_out = []
                hs = record(res_out)
                with torch.no_grad():
                    net(x)
                for h in hs: h.remove()
                plans = resident.describe(net)
                resident.disable(net)
                ref_out = []
                hs = record(ref_out)
                with torch.no_grad():
                    net(x)
                for h in hs: h.remove()
                ref = {}
                for name, ty, t in ref_out:
                    ref.setdefault(name, []).append(t)
                shown = 0
                for name, ty, t in res_out:
                    want = ref.get(name, [None]).pop(0) if ref.get(name) else None
                    if want is None or want.shape != t.shape:
                        continue
                    if not torch.equal(want, t):
                        pl = plans.get(name)
                        print("   %-6s %-10s differs in %d of %d values (max %.4g)  plan: %s" % (name, ty, int((want != t).sum()), t.numel(), float((want - t).abs().max()),
                              {k: getattr(pl, k) for k in ("relu", "emit_f32", "emit_int", "narrow_bit", "want_wide", "resident_add", "defer", "fuse_arg", "fuse_proj") if hasattr(pl, k)} if pl is not None else None))
                        shown += 1
                        if shown >= 4: break
                s = resident.enable(net, x, verify=False)
        except Exception as e:
            print(env or "default", "->", type(e).__name__, str(e)[:200])
        resident.disable(net)
