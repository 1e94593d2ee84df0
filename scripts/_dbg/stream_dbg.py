#!/usr/bin/env python3
"""GPU debug: one stream-kernel case, where the int8 output differs from the oracle (per pixel block / channel block)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
from common.quantity import _native as nat
from oracle import fq_oracle as orc
orc.build()
N, C, H, W, K, st, rs, ob, relu = [int(v) for v in sys.argv[1:10]]
rng = np.random.default_rng(5)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x = rng.integers(-128, 128, size=(N, C, H, W)).astype(np.int32)
w = rng.integers(-128, 128, size=(K, C, 1, 1)).astype(np.int32)
qb = rng.integers(-128, 128, size=K).astype(np.float32)
acc = orc.conv2d_int(x, w, (st, st), (0, 0), (1, 1))
ref = orc.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
if relu: ref = np.maximum(ref, np.float32(0))
want = orc.quantity(ref, ob).astype(np.int8).transpose(0, 2, 3, 1).reshape(-1, K)
xd, wd, bd = dev(x.transpose(0, 2, 3, 1).astype(np.int8)), nat.pack_weight_krsc(dev(w.astype(np.float32))), dev(qb)
_, q2 = nat.conv2d_i8_resident(xd, wd, bd, (st, st), (0, 0), (1, 1), rs, ob, False, True, bool(relu))
got = q2.cpu().numpy().reshape(-1, K)
bad = got != want
M = want.shape[0]
print("M", M, "bad elements", int(bad.sum()), "of", bad.size)
pb = bad.any(axis=1)
print("bad pixels", int(pb.sum()), "first", np.nonzero(pb)[0][:20], "last", np.nonzero(pb)[0][-5:])
for blk in (256, 32, 8):
    cnt = [int(pb[i:i + blk].sum()) for i in range(0, M, blk)]
    print("per %d px:" % blk, cnt[:64])
cb = bad.any(axis=0)
print("bad channels per 16:", [int(cb[i:i + 16].sum()) for i in range(0, K, 16)])
i = np.nonzero(pb)[0][0] if pb.any() else 0
print("pixel", i, "got", got[i, :32], "want", want[i, :32])
