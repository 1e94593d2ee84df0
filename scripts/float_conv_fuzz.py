#!/usr/bin/env python3
"""GPU fuzz: fq_conv1x1_f32 / fq_conv_kxk_f32 / fq_conv_stem_f32 / fq_conv3x3_wino_f32 / fq_conv1x1_sb_f32 on random shapes with integer-valued data (every partial sum
exact in fp32, so any indexing mistake is a wrong bit) against a float64 convolution, with the abs-max / histogram / ReLU copy
checked on the same output.  usage: float_conv_fuzz.py [cases=300] [seed=0]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat


def run(cases, seed, verbose=True):
    """Returns the list of failing configurations."""
    rng = np.random.default_rng(seed)
    failures = []
    counts = {}
    for it in range(cases):
        kind = rng.choice(["c1", "kxk", "stem", "wino", "sb"], p=[0.25, 0.25, 0.1, 0.25, 0.15])
        N = int(rng.integers(1, 9))
        if kind == "stem":
            cin, R, S, st = 3, 7, 7, 2
            cout = int(rng.integers(1, 65)); pad = int(rng.integers(0, 4))
            H, W = int(rng.integers(7, 80)), int(rng.integers(7, 80))
        elif kind == "wino":                                  # Winograd F(2x2, 3x3): U = G g Gt has quarters, every sum stays exact
            cin = 8 * int(rng.integers(1, 33)); cout = 64 * int(rng.integers(1, 5)); R = S = 3; st = 1; pad = 1
            H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        elif kind == "sb":                                    # split-bf16 1x1: small integers are their own bf16 "hi" piece
            cin = 16 * int(rng.integers(1, 33)); cout = 4 * int(rng.integers(1, 80)); R = S = 1; pad = 0; st = int(rng.choice([1, 1, 2]))
            H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
            if not nat.conv_sb_supported(cin, cout):
                kind = "c1"
        elif kind == "c1":
            cin = int(rng.integers(1, 200)); cout = 4 * int(rng.integers(1, 80)); R = S = 1; pad = 0
            st = int(rng.choice([1, 1, 2, 3])); H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        else:
            cin = 16 * int(rng.integers(1, 9)); cout = 4 * int(rng.integers(1, 60))
            R, S = int(rng.choice([1, 2, 3, 5])), int(rng.choice([1, 2, 3, 5]))
            if R == 1 and S == 1:
                S = 3
            st = int(rng.choice([1, 1, 2, 3])); pad = int(rng.integers(0, 3))
            H, W = int(rng.integers(max(1, R - 2 * pad), 36)), int(rng.integers(max(1, S - 2 * pad), 36))
        if kind != "stem" and rng.random() < 0.25:
            # larger launches: several hundred tiles, i.e. a last round over the 256 CUs that the tail split cuts along K
            # (fq_conv1x1_f32.hip plan_split); |sum| stays below 2^24 (K <= 1 152 products of magnitude <= 64)
            N = int(rng.integers(16, 80))
            if kind == "c1":
                cin = 16 * int(rng.integers(8, 64))
            if kind == "wino" and not nat.conv_wino_supported(N, cin, H, W, cout):
                N = int(rng.integers(1, 9))
        g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
        x = torch.randint(-8, 9, (N, cin, H, W), device="cuda", generator=g).float()
        w = torch.randint(-8, 9, (cout, cin, R, S), device="cuda", generator=g).float()
        b = torch.randint(-50, 51, (cout,), device="cuda", generator=g).float()
        ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=st, padding=pad)
        mx = torch.zeros(1, device="cuda")
        r = torch.empty(ref.shape, device="cuda")
        if kind == "wino" and not nat.conv_wino_supported(N, cin, H, W, cout):
            if cin % 16:                                         # (the direct kernel takes multiples of 16 input channels: nothing to run)
                continue
            kind = "kxk"
        if kind == "c1":
            run_k = lambda **kw: nat.conv1x1_f32(x, w.view(cout, cin).t().contiguous(), b, st, **kw)
        elif kind == "sb":
            wsb = nat.pack_sb_weight(w)
            run_k = lambda **kw: nat.conv1x1_f32(x, wsb, b, st, **kw)
        elif kind == "wino":
            wu = nat.pack_wino_weight(w)
            run_k = lambda **kw: nat.conv_wino_f32(x, wu, b, cout, **kw)
        elif kind == "kxk":
            run_k = lambda **kw: nat.conv_kxk_f32(x, nat.pack_kxk_weight(w), b, (R, S), st, pad, **kw)
        else:
            run_k = lambda **kw: nat.conv_stem_f32(x, nat.pack_stem_weight(w), b, cout, (7, 7), 2, pad, **kw)
        counts[kind] = counts.get(kind, 0) + 1
        y = run_k(max_dev=mx, row=0, relu_out=r)
        iv = torch.tensor([float(mx[0]) / 2048 + 1e-12], device="cuda")
        hist = torch.zeros(1, 2048, dtype=torch.int64, device="cuda")
        want = torch.zeros_like(hist)
        y2 = run_k(interval_dev=iv, hist_dev=hist, row=0)
        nat.hist2048_seg([y], [0], iv, want)
        r3 = torch.full_like(r, 7.0)                          # the ReLU-only form: y itself is not written
        mx3 = torch.zeros(1, device="cuda")
        run_k(max_dev=mx3, row=0, relu_out=r3, out=False)
        ok = (torch.equal(y.double(), ref) and torch.equal(y2, y) and float(mx[0]) == float(y.abs().max())
              and torch.equal(r, torch.relu(y)) and torch.equal(hist, want) and torch.equal(r3, r) and torch.equal(mx3, mx))
        if not ok:
            cfg = (kind, dict(N=N, cin=cin, cout=cout, H=H, W=W, R=R, S=S, stride=st, pad=pad))
            failures.append(cfg)
            if verbose:
                print("MISMATCH", cfg, flush=True)
    if verbose:
        print("kernels exercised:", ", ".join("%s %d" % kv for kv in sorted(counts.items())), flush=True)
    return failures


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("%d cases, %d mismatches" % (n, len(bad)))
    sys.exit(1 if bad else 0)
