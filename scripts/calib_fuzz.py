#!/usr/bin/env python3
"""GPU probe: randomised parity sweep of fq_absmax_seg / fq_hist2048_seg (and the per-channel forms) against the
CPU oracle: ragged segment lists, unaligned starts (views into a larger buffer), exact zeros, denormals, values on
bin edges, several segments per row, accumulation over two calls.  usage: calib_fuzz.py [rounds] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
from common.quantity import _native as nat
from oracle import fq_oracle as orc
orc.build()


def values(rng, n):
    kind = rng.integers(0, 6)
    if kind == 0:
        x = rng.standard_normal(n).astype(np.float32) * np.float32(10.0 ** rng.uniform(-6, 4))
    elif kind == 1:
        x = np.maximum(rng.standard_normal(n), 0).astype(np.float32)                    # ReLU-sparse
    elif kind == 2:
        x = (rng.integers(-2048, 2049, n) * np.float32(2.0 ** rng.integers(-10, 3))).astype(np.float32)   # bin edges
    elif kind == 3:
        x = rng.standard_normal(n).astype(np.float32) * np.float32(1e-41)             # denormals
    elif kind == 4:
        x = np.zeros(n, dtype=np.float32)
    else:
        x = rng.laplace(size=n).astype(np.float32)
        if n:
            x[rng.integers(0, n, max(n // 50, 1))] *= 50                                # outliers
    return x


def run(rounds, seed, verbose=True):
    rng = np.random.default_rng(seed)
    failures = []
    for it in range(rounds):
        rows = int(rng.integers(1, 6))
        nseg = int(rng.integers(1, 12))
        segs, seg_rows, hosts = [], [], []
        big = torch.empty(3_000_000, device="cuda")
        cursor = 0
        for s in range(nseg):
            n = int(rng.choice([0, 1, 3, 17, 255, 4096, 4097, 100_003, 400_000]))
            off = cursor + int(rng.integers(0, 4))                                   # 4-byte aligned, often not 16
            h = values(rng, n)
            big[off:off + n] = torch.from_numpy(h).cuda()
            segs.append(big[off:off + n]); seg_rows.append(int(rng.integers(0, rows))); hosts.append(h)
            cursor = off + n
        mx = torch.zeros(rows, device="cuda")
        nat.absmax_seg(segs, seg_rows, mx)
        ref_m = np.zeros(rows, dtype=np.float32)
        for h, r in zip(hosts, seg_rows):
            ref_m[r] = orc.absmax(h, ref_m[r])
        if not np.array_equal(mx.cpu().numpy(), ref_m):
            failures.append("round %d: absmax" % it)
        iv = np.array([orc.interval(m) for m in ref_m], dtype=np.float32)
        hist = torch.zeros(rows, 2048, dtype=torch.int64, device="cuda")
        ivd = torch.from_numpy(iv).cuda()
        nat.hist2048_seg(segs, seg_rows, ivd, hist)
        nat.hist2048_seg(segs[:1], seg_rows[:1], ivd, hist)                          # accumulates
        ref_h = np.zeros((rows, 2048), dtype=np.int64)
        for k, (h, r) in enumerate(zip(hosts, seg_rows)):
            orc.hist2048(h, iv[r], ref_h[r])
            if k == 0:
                orc.hist2048(h, iv[r], ref_h[r])
        if not np.array_equal(hist.cpu().numpy(), ref_h):
            failures.append("round %d: hist2048" % it)
        # per-channel form on a random NCHW tensor
        shape = (int(rng.integers(1, 6)), int(rng.integers(1, 9)), int(rng.choice([1, 5, 16, 33])), int(rng.choice([1, 7, 16, 40])))
        x = values(rng, int(np.prod(shape))).reshape(shape)
        xd = torch.from_numpy(x).cuda()
        C = shape[1]
        mxc = torch.zeros(C, device="cuda")
        nat.absmax_chan([xd], [0], mxc)
        ref_c = np.array([orc.absmax(np.ascontiguousarray(x[:, c]).ravel()) for c in range(C)], dtype=np.float32)
        ivc = np.array([orc.interval(m) for m in ref_c], dtype=np.float32)
        hc = torch.zeros(C, 2048, dtype=torch.int64, device="cuda")
        nat.hist2048_chan([xd], [0], torch.from_numpy(ivc).cuda(), hc)
        ref_hc = np.stack([orc.hist2048(np.ascontiguousarray(x[:, c]).ravel(), ivc[c]) for c in range(C)])
        if not (np.array_equal(mxc.cpu().numpy(), ref_c) and np.array_equal(hc.cpu().numpy(), ref_hc)):
            failures.append("round %d: per-channel %s" % (it, shape))
        if verbose and failures and failures[-1].startswith("round %d" % it):
            print("MISMATCH", failures[-1])
    return failures


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    fails = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("calib_fuzz: %d rounds, %d mismatches" % (n, len(fails)))
    sys.exit(1 if fails else 0)
