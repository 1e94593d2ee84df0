#!/usr/bin/env python3
"""GPU debug: one fused-add stream-kernel case; where wide / narrow differ from the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
from common.quantity import _native as nat
from oracle import fq_oracle as orc
orc.build()
N, C, H, W, K, rs, ob, relu, g_res, ib, r16 = [int(v) for v in sys.argv[1:12]]
rng = np.random.default_rng(5)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x = rng.integers(-128, 128, size=(N, C, H, W)).astype(np.int32)
w = rng.integers(-128, 128, size=(K, C, 1, 1)).astype(np.int32)
qb = rng.integers(-128, 128, size=K).astype(np.float32)
acc = orc.conv2d_int(x, w, (1, 1), (0, 0), (1, 1))
ref = orc.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
xd, wd, bd = dev(x.transpose(0, 2, 3, 1).astype(np.int8)), nat.pack_weight_krsc(dev(w.astype(np.float32))), dev(qb)
res_dtype = np.int16 if r16 else np.int8
g = max(0, ob, g_res)
lim = 128 * 2 ** max(g_res, 0) if r16 else 128
res = rng.integers(-min(lim, 32768), min(lim, 32768), size=(N, H, W, K)).astype(res_dtype)
s = orc.add_sat(ref, orc.dequantity(res.astype(np.float32), g_res).transpose(0, 3, 1, 2))
if relu: s = np.maximum(s, np.float32(0))
e = s.astype(np.float64) * 2.0 ** g
print("exact grid:", bool(np.all(e == np.rint(e))), "g", g)
for (ww, wn) in ((True, True), (True, False), (False, True)):
    wide, narrow = nat.conv2d_i8_add_resident(xd, wd, bd, (1, 1), (0, 0), (1, 1), rs, ob, dev(res), g_res, ww, g, wn, ib, bool(relu))
    for name, got, want in (("wide", wide, e.astype(np.int16)), ("narrow", narrow, orc.quantity(s, ib).astype(np.int8))):
        if got is None: continue
        gt = got.cpu().numpy().reshape(-1, K); wt = want.transpose(0, 2, 3, 1).reshape(-1, K)
        bad = gt != wt
        pb = bad.any(axis=1)
        print("want_wide", ww, "want_narrow", wn, name, "bad elements", int(bad.sum()), "bad pixels", int(pb.sum()), np.nonzero(pb)[0][:16])
        if pb.any():
            cb = bad.any(axis=0)
            print("   bad channels per 16:", [int(cb[i:i + 16].sum()) for i in range(0, K, 16)])
            i = np.nonzero(pb)[0][0]
            print("   pixel", i, "got", gt[i, :24], "want", wt[i, :24])
