"""GPU box: one layer of one model of scripts/model_fuzz.py: which kernel takes it, and how its output differs from the library's and
from a float64 reference.  usage: fuzz_layer_probe.py <index> <seed> <module name, e.g. m41> [odd]"""
import importlib.util, os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("model_fuzz", os.path.join(ROOT, "scripts", "model_fuzz.py"))
mf = importlib.util.module_from_spec(spec); spec.loader.exec_module(mf)
from common.quantity import _float_conv
i, seed, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
odd = "odd" in sys.argv[4:]
torch.backends.cudnn.deterministic = odd
odd = False
model, size, bs, rng = mf.random_net(i, seed, odd, "cuda")
x = torch.randn(bs, model.cin, size, size, device="cuda")
m = getattr(model, name)
print(m)
grabbed = {}
h = m.register_forward_hook(lambda mod, inp, out: grabbed.update(x=inp[0].detach().clone(), y=out.detach().clone()))
with torch.no_grad():
    model(x)
h.remove()
xin = grabbed["x"]
with torch.no_grad():
    k = _float_conv.kind(m, xin)
    print("kind:", k, " input", tuple(xin.shape))
    lib = torch.nn.Conv2d.forward(m, xin)
    ref = torch.nn.functional.conv2d(xin.double(), m.weight.double(), m.bias.double(), m.stride, m.padding, m.dilation, m.groups)
    bound = torch.nn.functional.conv2d(xin.abs().double(), m.weight.abs().double(), m.bias.abs().double(), m.stride, m.padding, m.dilation, m.groups)
    print("library vs float64: max |err| / bound %.2e, mean |err| %.2e" % (float(((lib.double() - ref).abs() / bound).max()), float((lib.double() - ref).abs().mean())))
    for kk in ([k] if k else []) + (["kxk"] if k == "wino" else []):
        own = _float_conv.plain(m, kk, xin, check=False)
        print("%-5s vs float64: max |err| / bound %.2e, mean |err| %.2e ; vs library: %d of %d values differ, mean |diff| %.2e" % (
            kk, float(((own.double() - ref).abs() / bound).max()), float((own.double() - ref).abs().mean()), int((own != lib).sum()), own.numel(), float((own - lib).abs().mean())))
    print("output max |y| %.6g, bin width %.3e" % (float(ref.abs().max()), float(ref.abs().max()) / 2048))
