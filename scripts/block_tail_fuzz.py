#!/usr/bin/env python3
"""GPU fuzz: fq_block_tail_i8 (conv3 + NewAdd + ReLU + the next block's conv1 in one kernel) against the two launches it replaces
(fq_conv2d_i8_add_resident, itself pinned to the CPU oracle in tests/test_gpu_resident.py, then fq_conv2d_i8_resident) on random
shapes, shifts, grids and output subsets; every output bit for bit.  A third of the 64-channel draws run the projection form
(fq_block_tail_proj_i8: the shortcut is a 1x1 convolution, stride 1 or 2, computed in the kernel) against fq_conv2d_i8_resident for
the projection followed by the two launches.  usage: block_tail_fuzz.py [cases=300] [seed=0]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat


def run(cases, seed):
    rng = np.random.default_rng(seed)
    bad, taken, refused, projs = [], 0, 0, 0
    for it in range(cases):
        C = int(rng.choice([64, 128, 256]))
        K3 = 128 * int(rng.integers(1, 9))
        C2 = int(rng.choice([0, 0, 64, 128]))
        N = int(rng.integers(1, 6)); H = int(rng.integers(1, 30)); W = int(rng.integers(1, 30))
        if rng.random() < 0.1:
            N, H, W = int(rng.integers(8, 30)), 56, 56 if C == 64 else 28          # several tiles per CU
            H = W
        res_dtype = torch.int16 if rng.random() < 0.7 else torch.int8
        ob3, g_res, ib = int(rng.integers(2, 7)), int(rng.integers(2, 7)), int(rng.integers(2, 6))
        rs3, rs1 = int(rng.integers(6, 13)), int(rng.integers(6, 13))
        relu, relu1 = bool(rng.random() < 0.8), bool(rng.random() < 0.8)
        want_wide, want_narrow = bool(rng.random() < 0.7), bool(rng.random() < 0.5)
        if not (C2 or want_wide or want_narrow):
            want_wide = True
        rb = 2 if res_dtype == torch.int16 else 1
        g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
        x = torch.randint(-128, 128, (N, H, W, C), dtype=torch.int8, device="cuda", generator=g)
        w3 = nat.pack_weight_krsc(torch.randint(-127, 128, (K3, C, 1, 1), device="cuda", generator=g).float())
        b3 = torch.randint(-100, 101, (K3,), device="cuda", generator=g).float()
        lim = 3000 if res_dtype == torch.int16 else 128
        res = torch.randint(-lim, lim, (N, H, W, K3), dtype=res_dtype, device="cuda", generator=g)
        w1 = b1 = None
        if C2:
            w1 = nat.pack_weight_krsc(torch.randint(-127, 128, (C2, K3, 1, 1), device="cuda", generator=g).float())
            b1 = torch.randint(-100, 101, (C2,), device="cuda", generator=g).float()
        g_wide = max(0, ob3, g_res)
        if C == 64 and C2 in (0, 64) and rng.random() < 0.33:
            # the projection form: the shortcut = Sp(RightShift(conv1x1(xp, wp)) + bias) on the grid g_res, int8
            sp, rsp = int(rng.choice([1, 2])), int(rng.integers(6, 13))
            Hp, Wp = (H - 1) * sp + 1 + int(rng.integers(0, sp)), (W - 1) * sp + 1 + int(rng.integers(0, sp))
            xp = torch.randint(-128, 128, (N, Hp, Wp, 64), dtype=torch.int8, device="cuda", generator=g)
            wp = nat.pack_weight_krsc(torch.randint(-127, 128, (K3, 64, 1, 1), device="cuda", generator=g).float())
            bp = torch.randint(-100, 101, (K3,), device="cuda", generator=g).float()
            if not nat.block_tail_proj_supported(C, K3, C2, 64, rs3, rs1, rsp, sp):
                refused += 1
                continue
            taken += 1
            projs += 1
            _, pres = nat.conv2d_i8_resident(xp, wp, bp, (sp, sp), (0, 0), (1, 1), rsp, g_res, False, True, False)
            wide, narrow = nat.conv2d_i8_add_resident(x, w3, b3, (1, 1), (0, 0), (1, 1), rs3, ob3, pres, g_res, want_wide, g_wide, True, ib, relu)
            q1 = None
            if C2:
                _, q1 = nat.conv2d_i8_resident(narrow, w1, b1, (1, 1), (0, 0), (1, 1), rs1, ib, False, True, relu1)
            ref = (wide, narrow if want_narrow else None, q1)
            got = nat.block_tail_proj_i8(x, w3, b3, rs3, ob3, xp, wp, bp, rsp, g_res, sp, want_wide, g_wide, want_narrow, ib, relu, w1, b1,
                                         rs1, relu1)
            if not all((a is None) == (b is None) and (a is None or torch.equal(a, b)) for a, b in zip(got, ref)):
                cfg = dict(form="proj", N=N, H=H, W=W, K3=K3, C2=C2, sp=sp, Hp=Hp, Wp=Wp, ob3=ob3, obp=g_res, ib=ib, rs3=rs3, rsp=rsp, rs1=rs1,
                           relu=relu, relu1=relu1, wide=want_wide, narrow=want_narrow)
                bad.append(cfg)
                print("MISMATCH", cfg, flush=True)
            continue
        if not nat.block_tail_supported(C, K3, C2, rs3, rs1, ob3, g_res, rb, ib):
            refused += 1
            continue
        taken += 1
        wide, narrow = nat.conv2d_i8_add_resident(x, w3, b3, (1, 1), (0, 0), (1, 1), rs3, ob3, res, g_res, want_wide, g_wide, True, ib, relu)
        q1 = None
        if C2:
            _, q1 = nat.conv2d_i8_resident(narrow, w1, b1, (1, 1), (0, 0), (1, 1), rs1, ib, False, True, relu1)
        ref = (wide, narrow if want_narrow else None, q1)
        got = nat.block_tail_i8(x, w3, b3, rs3, ob3, res, g_res, want_wide, g_wide, want_narrow, ib, relu, w1, b1, rs1, relu1)
        ok = all((a is None) == (b is None) and (a is None or torch.equal(a, b)) for a, b in zip(got, ref))
        if not ok:
            cfg = dict(N=N, H=H, W=W, C=C, K3=K3, C2=C2, res=str(res_dtype), ob3=ob3, g_res=g_res, ib=ib, rs3=rs3, rs1=rs1, relu=relu,
                       relu1=relu1, wide=want_wide, narrow=want_narrow)
            bad.append(cfg)
            print("MISMATCH", cfg, flush=True)
    print("%d cases: %d taken by the kernels (%d of them the projection form), %d refused (grids outside their forms), %d mismatches" %
          (cases, taken, projs, refused, len(bad)))
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    sys.exit(1 if run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
