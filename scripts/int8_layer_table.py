#!/usr/bin/env python3
"""GPU probe: every integer conv / linear launch of ONE resident ReconModel forward of the fabu ResNet-50, one row per
launch: shape, time (HIP events around the C-ABI call, median over the forwards), TOP/s, algorithmic GB/s, and the
launch's semantic bound max(matrix work at 5 POP/s, operand + result bytes at 8 TB/s).
usage: int8_layer_table.py [batch] [forwards]     (the table the judge asked for; -> profiles/rNN_int8_layer_table_b<batch>.txt)"""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
import bench
from tools import Quantity, Reconstruction
from common.quantity import _native, resident

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
FWD = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = torch.device("cuda")
out = sys.stdout; sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(1, "1,3,224,224", 0)
data = bench.DeviceBatches(2, B, 224, 0, 1, dev)
q = Quantity(model); q.activation_quantize(data); q.weight_quantize()
rec = Reconstruction(bench.build_model("r50", 224, dev))
net = rec.ReconModel(rec.get_quantity_information(), "./workdir/recon.pth")
sys.stdout = out
x = data[0][0]
resident.enable(net, x)

names = ("conv2d_i8_resident", "conv2d_i8_add_resident", "conv2d_i8_stem", "conv2d_i8", "block_tail_i8", "block_tail_proj_i8")
saved = {n: getattr(_native, n) for n in names}
rows, events = [], []


def nbytes(*ts):
    return sum(int(t.numel()) * t.element_size() for t in ts if isinstance(t, torch.Tensor))


def timed(fn, name):
    def wrapper(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        outs = r if isinstance(r, tuple) else (r,)
        res = a[8] if name == "conv2d_i8_add_resident" else (a[5] if name == "block_tail_i8" else None)
        first = next(t for t in outs if isinstance(t, torch.Tensor))
        pixels = first.numel() // first.shape[1 if first.dtype == torch.float32 else -1]
        wq = a[1]
        w_next = (a[12] if len(a) > 12 else k.get("w1q")) if name == "block_tail_i8" else None      # the fused next conv1
        w_proj = None
        if name == "block_tail_proj_i8":                      # the projection shortcut computed in the kernel: its input and weights
            res, w_proj = a[5], a[6]
            w_next = a[16] if len(a) > 16 else k.get("w1q")
        events.append((e0, e1))
        label = (name.replace("conv2d_i8_", "") if not name.startswith("block_tail")
                 else ("proj+" if w_proj is not None else "") + ("tail+conv1" if w_next is not None else "block_tail"))
        wshape = tuple(wq.shape) if w_next is None else tuple(wq.shape[:1]) + (int(w_next.shape[0]),) + tuple(wq.shape[3:])
        rows.append((label, tuple(a[0].shape), wshape, pixels,
                     pixels * (int(wq.numel()) + (int(w_next.numel()) if w_next is not None else 0) + (int(w_proj.numel()) if w_proj is not None else 0)),
                     nbytes(a[0], a[1], res, w_next, w_proj, *outs)))
        return r
    return wrapper


with torch.no_grad():
    for _ in range(3):
        net(x)
    for n in names:
        setattr(_native, n, timed(saved[n], n))
    for _ in range(FWD):
        net(x)
    torch.cuda.synchronize()
for n in names:
    setattr(_native, n, saved[n])
per = [a.elapsed_time(b) * 1e3 for a, b in events]
L = len(rows) // FWD
print("%-3s %-15s %-22s %-20s %9s %8s %8s %8s %8s" % ("#", "call", "x", "w", "us", "TOP/s", "GB/s", "mfma us", "hbm us"))
tot = tb = 0.0
for i in range(L):
    name, xs, ws, pixels, macs, nb = rows[i]
    us = statistics.median(per[i + f * L] for f in range(FWD))
    t_m, t_h = 2.0 * macs / 5.0e15 * 1e6, nb / 8.0e12 * 1e6
    tot += us; tb += max(t_m, t_h)
    print("%-3d %-15s %-22s %-20s %9.1f %8.1f %8.0f %8.1f %8.1f" % (i, name, "x".join(map(str, xs)), "x".join(map(str, ws)), us,
                                                                   2.0 * macs / us / 1e6, nb / us / 1e3, t_m, t_h))
print("one forward of %d images: %d launches, %.3f ms; semantic bound %.3f ms; fraction %.3f" % (B, L, tot / 1e3, tb / 1e3, tb / tot))
