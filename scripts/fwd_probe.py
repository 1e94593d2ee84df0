#!/usr/bin/env python3
"""GPU probe: float forward time of the fabu ResNet-50 under different layouts / MIOpen settings."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
import bench
sys.stdout = open(os.devnull, "w")
m = bench.build_model("r50", 224, torch.device("cuda"))
sys.stdout = sys.__stdout__
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(NB, 3, 224, 224, device="cuda")
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for bm in (False, True):
    torch.backends.cudnn.benchmark = bm
    with torch.no_grad():
        print("benchmark=%s NCHW: %.2f ms/batch" % (bm, t(lambda: m(x))))
        mc = m.to(memory_format=torch.channels_last); xc = x.to(memory_format=torch.channels_last)
        print("benchmark=%s NHWC: %.2f ms/batch" % (bm, t(lambda: mc(xc))))
        m = m.to(memory_format=torch.contiguous_format)
for B in (32, 128, 256):
    xb = torch.randn(B, 3, 224, 224, device="cuda")
    with torch.no_grad():
        print("NCHW B=%d: %.3f ms/img" % (B, t(lambda: m(xb)) / B))
