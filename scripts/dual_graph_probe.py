#!/usr/bin/env python3
"""GPU probe: the resident ReconModel forward of 2*B images as ONE HIP graph of 2*B images versus TWO graphs of B images
replayed on two streams (each kernel's ramp-up and drain -- ~9 us of a 20-60 us launch at these sizes -- can then hide
under the other stream's kernels).   usage: dual_graph_probe.py [B per graph] [iters]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
import bench
from tools import Quantity, Reconstruction
from common.quantity import resident
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(1, "1,3,224,224", 0)
data = bench.DeviceBatches(2, 2 * B, 224, 0, 1, dev)
q = Quantity(model); q.activation_quantize(data); q.weight_quantize()
rec = Reconstruction(bench.build_model("r50", 224, dev))
net = rec.ReconModel(rec.get_quantity_information(), "./workdir/recon.pth")
sys.stdout = out
x = data[0][0]
resident.enable(net, x)
with torch.no_grad():
    want = net(x).clone()


def rate(fn, n_img):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(ITERS): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / ITERS
    return dt * 1e3, n_img / dt


with torch.no_grad():
    ms, r = rate(lambda: net(x), 2 * B)
    print("eager, %d images per forward: %.3f ms = %.0f img/s" % (2 * B, ms, r))
g_big = resident.capture(net, x)
ms, r = rate(lambda: g_big(x), 2 * B)
print("one graph of %d images: %.3f ms = %.0f img/s" % (2 * B, ms, r))
xa, xb = x[:B].contiguous(), x[B:].contiguous()
ga, gb = resident.capture(net, xa), resident.capture(net, xb)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()


def dual():
    sa.wait_stream(main); sb.wait_stream(main)
    with torch.cuda.stream(sa):
        oa = ga(xa)
    with torch.cuda.stream(sb):
        ob = gb(xb)
    main.wait_stream(sa); main.wait_stream(sb)
    return oa, ob


oa, ob = dual()
torch.cuda.synchronize()
print("dual-graph logits identical to the eager forward:", bool(torch.equal(torch.cat([oa, ob]), want)))
ms, r = rate(dual, 2 * B)
print("two graphs of %d images on two streams: %.3f ms = %.0f img/s" % (B, ms, r))
ms, r = rate(lambda: (ga(xa), gb(xb)), 2 * B)
print("two graphs of %d images on ONE stream: %.3f ms = %.0f img/s" % (B, ms, r))
# S graphs of 2B / S images on S streams
for S in (3, 4):
    if (2 * B) % S:
        continue
    parts = [p.contiguous() for p in x.chunk(S)]
    graphs = [resident.capture(net, p) for p in parts]
    streams = [torch.cuda.Stream() for _ in range(S)]

    def multi():
        outs = []
        for st, g, p in zip(streams, graphs, parts):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(g(p))
        for st in streams:
            main.wait_stream(st)
        return outs
    outs = multi(); torch.cuda.synchronize()
    ok = bool(torch.equal(torch.cat(outs), want))
    ms, r = rate(multi, 2 * B)
    print("%d graphs of %d images on %d streams: %.3f ms = %.0f img/s (identical: %s)" % (S, 2 * B // S, S, ms, r, ok))
    del graphs
