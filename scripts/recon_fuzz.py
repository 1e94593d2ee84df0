#!/usr/bin/env python3
"""GPU fuzz of the integer-simulation model with RESIDENT integer activations on random model topologies (the generator of
scripts/model_fuzz.py): calibrate, rewrite, rebuild as ReconModel, then the logits with fp32 module boundaries (the reference's
form) against the logits of the resident plan (int8 / int16 hand-offs, fused ReLUs, conv + NewAdd in one kernel, block tails, pools
on integers), eagerly and as one HIP graph: bit for bit.  The planner decides per edge who may hand over integers; a wrong decision
on a graph nobody wrote a test for shows here.
usage: recon_fuzz.py [models=40] [seed=1]"""
import importlib.util, os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
mf = importlib.util.module_from_spec(spec)
sys.modules["model_fuzz"] = mf                        # (the reference's flow pickles whole models: the class must be importable)
spec.loader.exec_module(mf)
from common.quantity import resident, _native
from tools import Quantity, Reconstruction
from workdir_util import product_workdir


def build(i, seed, odd=False, share=False, bn=False):
    model, size, bs, _rng = mf.fold(mf.random_net(i, seed, odd, "cuda", share, bn))
    return model, size, bs


def recon_of(model, twin, data):
    """Calibrate `model` on `data`, quantise and rewrite its weights, and rebuild `twin` (the same weights, as the reference's flow
    loads them) as the integer-simulation model.  Call inside product_workdir()."""
    out = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        q = Quantity(model)
        q.activation_quantize(data)
        q.weight_quantize()
        q.rewrite_weight()
        rec = Reconstruction(twin)
        return rec.ReconModel(rec.get_quantity_information(), "./workdir/recon.pth")
    finally:
        sys.stdout = out


def run(n, seed, log=print, odd=False, share=False, bn=False, big=False, variants=None):
    bad, seen = 0, {}
    variants = variants if variants is not None else {}
    for i in range(n):
        model, size, bs, rng = mf.fold(mf.random_net(i, seed, odd, "cuda", share, bn))
        data = [(torch.randn(bs, model.cin, size, size, device="cuda"), torch.zeros(bs, dtype=torch.long)) for _ in range(2)]
        out = sys.stdout
        try:
            with product_workdir(input_shape="1,%d,%d,%d" % (model.cin, size, size), device="gpu", max_cali_img_num=1):
                net = recon_of(model, build(i, seed, odd, share, bn)[0], data)
                x = data[0][0]
                with torch.no_grad():
                    plain = net(x)
                    summary = resident.enable(net, x)
                    got = net(x)
                    x2 = torch.flip(x, dims=[0]) * 0.5
                    got2 = net(x2)
                    graphed = resident.capture(net, x)
                    g1, g2 = graphed(x).clone(), graphed(x2).clone()
                    # two graphs of half the batch on two streams; another batch size than the plan was made on; the plan pickled
                    dual = resident.capture(net, x, streams=2)
                    d1, d2 = dual(x).clone(), dual(x2).clone()
                    got3 = net(x[:3])
                    torch.save(net, "./workdir/recon_resident.pth")
                    again = torch.load("./workdir/recon_resident.pth", weights_only=False)
                    got4 = again(x2)
                    got5 = want5 = None
                    if big:                                   # a batch at which the launches take their many-workgroup forms
                        xb = torch.randn(256, model.cin, size, size, device="cuda")
                        _native.conv_variant_log = {}
                        got5 = net(xb)
                        for k, v in _native.conv_variant_log.items():
                            variants[k] = variants.get(k, 0) + v
                        _native.conv_variant_log = None
                    resident.disable(net)
                    want2 = net(x2)
                    want3 = net(x[:3])
                    if big:
                        want5 = net(xb)
        except Exception as e:
            sys.stdout = out
            bad += 1
            log("model %d (seed %d): %s: %s" % (i, seed, type(e).__name__, str(e)[:300]))
            continue
        for k, v in summary.items():
            if isinstance(v, int):
                seen[k] = seen.get(k, 0) + v
        problems = []
        if not torch.equal(got, plain) or not torch.equal(got2, want2):
            problems.append("resident logits differ (max %.3g)" % float((got - plain).abs().max()))
        if not torch.equal(g1, plain) or not torch.equal(g2, want2):
            problems.append("graphed logits differ")
        if not torch.equal(d1, plain) or not torch.equal(d2, want2):
            problems.append("logits of the two half-batch graphs differ")
        if not torch.equal(got3, want3):
            problems.append("resident logits differ at another batch size")
        if not torch.equal(got4, want2):
            problems.append("logits of the pickled resident model differ")
        if big and not torch.equal(got5, want5):
            problems.append("resident logits differ at 256 images")
        if problems:
            bad += 1
            log("model %d (seed %d, %d modules): %s; plan %s" % (i, seed, model.n, "; ".join(problems), {k: v for k, v in summary.items() if isinstance(v, int)}))
    return bad, seen


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    odd, share, bn, big = "odd" in sys.argv[3:], "share" in sys.argv[3:], "bn" in sys.argv[3:], "big" in sys.argv[3:]
    variants = {}
    bad, seen = run(n, seed, odd=odd, share=share, bn=bn, big=big, variants=variants)
    if big:
        print("integer kernels at 256 images:", dict(sorted(variants.items())))
    print("recon_fuzz%s: %d random models (seed %d), %d with a finding; plans in all: %s" % ((" odd" if odd else "") + (" share" if share else "") + (" bn" if bn else ""), n, seed, bad, seen))
