#!/usr/bin/env python3
"""GPU probe: the screened KL search against the exhaustive one on many random histograms (scripts/kl_fuzz_hist.py
families + activation-like rows): thresholds and best KL must agree for every row; prints how many candidates survive
the screen.   usage: kl_screen_check.py [rows] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from common.quantity import _native as nat
import kl_fuzz_hist

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hs = [kl_fuzz_hist.random_histogram(rng) for _ in range(rows // 2)]
# activation-like rows: |N(0, s)| samples binned with a max-based interval (what per-channel calibration produces)
for _ in range(rows - len(hs)):
    n = int(10 ** rng.uniform(3, 6.5))
    x = np.abs(rng.standard_normal(n).astype(np.float32)) * (rng.random() < 0.3 and rng.uniform(0.1, 3) or 1.0)
    if rng.random() < 0.5:
        x = x[rng.random(n) < 0.5]                       # ReLU-like thinning does not change the shape; fewer samples
    iv = np.float32(x.max() / 2048 + 1e-12)
    hs.append(np.bincount(np.minimum((x / iv).astype(np.int64), 2047), minlength=2048).astype(np.int64))
H = torch.from_numpy(np.stack(hs)).cuda()
thr_x, best_x, run_x = (t.cpu().numpy() for t in nat.kl_threshold(H, mode=nat.KL_EXHAUSTIVE, want_evidence=True))
thr_s, cur_s, best_s, run_s = (t.cpu().numpy() for t in nat.kl_threshold(H, want_curve=True, mode=nat.KL_SCREENED, want_evidence=True))
bad = np.flatnonzero((thr_x != thr_s) | (best_x.view(np.uint64) != best_s.view(np.uint64)))
print("rows %d: threshold / best-KL mismatches: %d" % (rows, len(bad)))
mn = np.nanmin(np.where(np.isfinite(cur_s), cur_s, np.inf), axis=1, keepdims=True)
surv = (~(cur_s > mn + 1e-10 + 1e-12 * np.abs(mn))).sum(axis=1)           # upper bound: includes the exact entries
for name, sl in (("fuzz families", slice(0, rows // 2)), ("activation-like", slice(rows // 2, rows))):
    s = surv[sl]
    print("%-16s survivors per row: median %d, mean %.1f, 95%% %d, max %d; rows swept in full (> 64): %.2f %%" %
          (name, np.median(s), s.mean(), np.percentile(s, 95), s.max(), 100.0 * (s > 64).mean()))
margin = run_x - best_x
fin = np.isfinite(margin)
print("argmin margin (runner-up - best), finite rows: min %.3e, median %.3e; rows with margin < 1e-12 * best: %d" %
      (margin[fin].min(), np.median(margin[fin]), int((margin[fin] <= 1e-12 * np.abs(best_x[fin])).sum())))
sys.exit(1 if len(bad) else 0)
