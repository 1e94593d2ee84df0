#!/usr/bin/env python3
"""GPU probe: randomised parity sweep of fq_kl_threshold against the CPU oracle run with the same logarithm
(include/fq_log.h): thresholds equal and KL curves equal BIT FOR BIT (NaNs at the same candidates).
usage: kl_fuzz.py [rows] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from common.quantity import _native as nat
from oracle import fq_oracle as orc
orc.build()


from kl_fuzz_hist import random_histogram  # noqa: E402


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
