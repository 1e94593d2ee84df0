#!/usr/bin/env python3
"""GPU probe: randomised parity sweep of fq_kl_threshold against the CPU oracle run with the same logarithm
(include/fq_log.h): thresholds equal and KL curves equal BIT FOR BIT (NaNs at the same candidates).
usage: kl_fuzz.py [rows] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
from common.quantity import _native as nat
from oracle import fq_oracle as orc
orc.build()


def random_histogram(rng):
    kind = rng.integers(0, 8)
    x = np.arange(2048, dtype=np.float64)
    scale = 10.0 ** rng.uniform(1, 9)
    if kind == 0:      # half-gaussian of random width
        h = np.exp(-0.5 * (x / rng.uniform(20, 900)) ** 2)
    elif kind == 1:    # exponential / laplace tail
        h = np.exp(-x / rng.uniform(5, 600))
    elif kind == 2:    # ReLU-like: spike at 0 plus a tail
        h = np.exp(-x / rng.uniform(30, 400)); h[0] *= rng.uniform(10, 1e4)
    elif kind == 3:    # sparse: most bins empty
        h = np.where(rng.random(2048) < rng.uniform(0.01, 0.3), rng.random(2048), 0.0)
    elif kind == 4:    # uniform with noise
        h = 1.0 + 0.1 * rng.random(2048)
    elif kind == 5:    # bumps
        h = sum(np.exp(-0.5 * ((x - rng.uniform(0, 2047)) / rng.uniform(2, 80)) ** 2) for _ in range(rng.integers(1, 6)))
    elif kind == 6:    # a single outlier bin far out, mass near zero
        h = np.exp(-x / rng.uniform(2, 30)); h[rng.integers(1500, 2048)] += 1.0 / scale * rng.integers(1, 5)
    else:              # tiny counts (0..3 per bin)
        h = rng.integers(0, 4, 2048).astype(np.float64); scale = 1.0
    h = np.rint(h * scale)
    if rng.random() < 0.2:
        h[rng.integers(128, 2048):] = 0          # empty tail
    return h.astype(np.int64)


def run(rows, seed, verbose=True):
    rng = np.random.default_rng(seed)
    hists = np.stack([random_histogram(rng) for _ in range(rows)])
    thr, curve = nat.kl_threshold(torch.from_numpy(hists).cuda(), want_curve=True)
    thr, curve = thr.cpu().numpy(), curve.cpu().numpy()
    failures = []
    for r in range(rows):
        t, c = orc.kl_threshold(orc.normalize(hists[r]), want_curve=True, use_fq_log=True)
        same_curve = np.array_equal(np.isnan(c), np.isnan(curve[r])) and np.array_equal(c[~np.isnan(c)].view(np.uint64),
                                                                                        curve[r][~np.isnan(c)].view(np.uint64))
        if int(thr[r]) != int(t) or not same_curve:
            failures.append("row %d: threshold %d vs %d, curve equal %s" % (r, thr[r], t, same_curve))
            if verbose:
                print("MISMATCH", failures[-1])
    return failures


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    fails = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("kl_fuzz: %d rows, %d mismatches" % (n, len(fails)))
    sys.exit(1 if fails else 0)
