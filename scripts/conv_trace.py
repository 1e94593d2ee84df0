#!/usr/bin/env python3
"""GPU probe (debug build libfq_hip_trace.so, -DFQ_CONV_TRACE): s_memtime stamps inside conv2d_i8_dma_kernel.
usage: conv_trace.py C H K R stride pad"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
nat.LIB_PATH = nat.LIB_PATH.replace("libfq_hip.so", "libfq_hip_trace.so")
C, H, K, R, st, pd = [int(v) for v in sys.argv[1:7]]
B = int(os.environ.get("FQ_CONV_ONE_BATCH", "128"))
x = torch.randn(B, C, H, H, device="cuda") * 2
w = torch.randint(-127, 128, (K, C, R, R), device="cuda").float()
qb = torch.randint(-100, 100, (K,), device="cuda").float()
wq = nat.pack_weight_krsc(w)
xq = nat.quantize_i8_nhwc(x, 4, wq.shape[-1])
for _ in range(3):
    nat.conv2d_i8_resident(xq, wq, qb, (st, st), (pd, pd), (1, 1), 8, 4, False, True, True)
torch.cuda.synchronize()
buf = np.zeros(8192, dtype=np.uint64)
L = nat.lib()
L.fq_debug_read_trace.argtypes = [ctypes.c_void_p]
assert L.fq_debug_read_trace(buf.ctypes.data) == 0
nsteps = R * R * C // 128
for wg in range(8):
    t = buf[wg * 512:(wg + 1) * 512].astype(np.int64)
    if t[0] == 0:
        continue
    print("WG slot %d: prologue %d, first load+barrier %d, loop %d, epilogue %d cycles (s_memtime ticks)" %
          (wg, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3]))
    rows = []
    for s_ in range(nsteps):
        e = t[8 + 8 * s_: 16 + 8 * s_]
        if s_ + 1 < nsteps:
            rows.append("  step %2d: tap offsets %4d  reads+dma+mfma %4d  vmcnt %4d  barrier %4d  | total %5d" %
                        (s_, e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3], (t[8 + 8 * (s_ + 1)] - e[0])))
        else:
            rows.append("  step %2d: tap offsets %4d  reads+dma+mfma %4d" % (s_, e[1] - e[0], e[2] - e[1]))
    print("\n".join(rows[:6] + rows[-2:]))
