#!/usr/bin/env python3
"""GPU probe: randomised parity sweep of fq_conv2d_i8 / _resident / _add_resident against the CPU oracle
(oracle/fq_oracle.c: exact integer conv + the reference's fp32 tail).  usage: conv_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity")); sys.path.insert(0, ROOT)
from common.quantity import _native as nat
from oracle import fq_oracle as orc
orc.build()


def run(cases, seed, verbose=True):
    """Returns the list of mismatching cases (empty = parity)."""
    rng = np.random.default_rng(seed)
    failures = []
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for it in range(cases):
        kind = rng.integers(0, 5)
        if kind == 0:      # LDS-DMA shapes: C % 128 == 0, K % 64 == 0, deep reduction
            C = int(rng.choice([128, 256, 512])); K = int(rng.choice([64, 128, 192, 256])); R = int(rng.choice([1, 3]))
            if R == 1: C = int(rng.choice([1024, 2048]))
        elif kind == 1:    # register-staged fast path
            C = int(rng.choice([128, 256])); K = int(rng.choice([64, 128, 256, 320])); R = int(rng.choice([1, 1, 3]))
        elif kind == 2:    # C == 64 path (two taps per K-step; odd and even tap counts)
            C = 64; K = int(rng.choice([64, 100, 128, 256])); R = int(rng.choice([1, 2, 3]))
        else:              # general path: ragged channels
            C = int(rng.integers(1, 80)); K = int(rng.integers(1, 150)); R = int(rng.choice([1, 2, 3, 5]))
        S = R if rng.random() < 0.8 else int(rng.choice([1, 2, 3]))
        H = int(rng.integers(R, 13)); W = int(rng.integers(S, 13)); N = int(rng.integers(1, 5))
        st = int(rng.choice([1, 1, 2])); pd = int(rng.integers(0, (min(R, S) + 1) // 2 + 1)); dl = int(rng.choice([1, 1, 1, 2]))
        if (H + 2 * pd - dl * (R - 1) - 1) // st + 1 <= 0 or (W + 2 * pd - dl * (S - 1) - 1) // st + 1 <= 0:
            continue
        x = rng.integers(-128, 128, size=(N, C, H, W)).astype(np.int32)
        w = rng.integers(-128, 128, size=(K, C, R, S)).astype(np.int32)
        qb = rng.integers(-128, 128, size=K).astype(np.float32)
        rs = int(rng.integers(0, 19)); ob = int(rng.integers(-2, 7)); relu = bool(rng.integers(0, 2))
        acc = orc.conv2d_int(x, w, (st, st), (pd, pd), (dl, dl))
        ref = orc.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
        cpad = (C + 15) // 16 * 16; kpad = (K + 15) // 16 * 16
        xn = np.zeros((N, H, W, cpad), dtype=np.int8); xn[..., :C] = x.transpose(0, 2, 3, 1)
        wd = nat.pack_weight_krsc(dev(w.astype(np.float32)))
        xd, bd = dev(xn), dev(qb)
        msg = "case %d: N%d C%d H%d W%d K%d R%d S%d st%d pd%d dl%d rs%d ob%d relu%d" % (it, N, C, H, W, K, R, S, st, pd, dl, rs, ob, relu)
        try:
            y = nat.conv2d_i8(xd, wd, bd, (st, st), (pd, pd), (dl, dl), rs, ob).cpu().numpy()
            assert np.array_equal(y, ref), "fp32 output"
            refr = np.maximum(ref, np.float32(0)) if relu else ref
            y2, q2 = nat.conv2d_i8_resident(xd, wd, bd, (st, st), (pd, pd), (dl, dl), rs, ob, True, True, relu)
            assert np.array_equal(y2.cpu().numpy(), refr), "resident fp32"
            qn = q2.cpu().numpy()
            assert np.array_equal(qn[..., :K].transpose(0, 3, 1, 2), orc.quantity(refr, ob).astype(np.int8)) and not qn[..., K:].any(), "resident int8"
            # fused residual add
            g_res = int(rng.integers(-1, 9)); res_dtype = np.int16 if rng.random() < 0.6 else np.int8
            g = max(0, ob, g_res)
            if g <= 8 and -16 <= ob <= 16:
                P, Q = ref.shape[2], ref.shape[3]
                lim = 128 * 2 ** max(g_res, 0) if res_dtype == np.int16 else 128
                res = np.zeros((N, P, Q, kpad), dtype=res_dtype)
                res[..., :K] = rng.integers(-min(lim, 32768), min(lim, 32768), size=(N, P, Q, K))
                ib = int(rng.integers(-1, 7))
                s = orc.add_sat(ref, orc.dequantity(res[..., :K].astype(np.float32), g_res).transpose(0, 3, 1, 2))
                if relu: s = np.maximum(s, np.float32(0))
                e = s.astype(np.float64) * 2.0 ** g
                if np.all(e == np.rint(e)):
                    wide, narrow = nat.conv2d_i8_add_resident(xd, wd, bd, (st, st), (pd, pd), (dl, dl), rs, ob, dev(res), g_res, True, g, True, ib, relu)
                    assert np.array_equal(wide.cpu().numpy()[..., :K].transpose(0, 3, 1, 2), e.astype(np.int16)), "fused add wide"
                    assert np.array_equal(narrow.cpu().numpy()[..., :K].transpose(0, 3, 1, 2), orc.quantity(s, ib).astype(np.int8)), "fused add narrow"
        except AssertionError as ex:
            failures.append("%s -> %s" % (msg, ex))
            if verbose:
                print("MISMATCH", msg, "->", ex)

    return failures


def run_stream(cases, seed, verbose=True):
    """The streaming 1x1 kernel (csrc/fq_conv1x1_i8.hip): every shape class it takes -- C = 64 or a multiple of 128, K a
    multiple of 64, stride 1 / 2, int8 output alone or the fused NewAdd with an int8 / int16 residual and any subset of
    {wide, narrow} -- on pixel counts that end inside a tile, inside a wave and inside a DMA row group.  With
    FQ_STREAM_GROUPS=1 in the environment (read once, at the first launch) every launch runs as 8 streams, so that small
    inputs walk several pixel tiles per workgroup."""
    rng = np.random.default_rng(seed)
    failures = []
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for it in range(cases):
        C = int(rng.choice([64, 64, 128, 128, 256, 384, 512, 1024])); K = int(rng.choice([64, 128, 192, 256, 512]))
        st = int(rng.choice([1, 1, 1, 2]))
        N = int(rng.integers(1, 7)); H = int(rng.integers(1, 40)); W = int(rng.integers(1, 40))
        if rng.random() < 0.3:
            N, H, W = int(rng.integers(1, 4)), int(rng.integers(30, 64)), int(rng.integers(30, 64))
        x = rng.integers(-128, 128, size=(N, C, H, W)).astype(np.int32)
        w = rng.integers(-128, 128, size=(K, C, 1, 1)).astype(np.int32)
        qb = rng.integers(-128, 128, size=K).astype(np.float32)
        rs = int(rng.integers(1, 17)); ob = int(rng.integers(-2, 7)); relu = bool(rng.integers(0, 2))
        acc = orc.conv2d_int(x, w, (st, st), (0, 0), (1, 1))
        ref = orc.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
        xd, wd, bd = dev(x.transpose(0, 2, 3, 1).astype(np.int8)), nat.pack_weight_krsc(dev(w.astype(np.float32))), dev(qb)
        msg = "stream %d: N%d C%d H%d W%d K%d st%d rs%d ob%d relu%d" % (it, N, C, H, W, K, st, rs, ob, relu)
        try:
            refr = np.maximum(ref, np.float32(0)) if relu else ref
            _, q2 = nat.conv2d_i8_resident(xd, wd, bd, (st, st), (0, 0), (1, 1), rs, ob, False, True, relu)
            assert np.array_equal(q2.cpu().numpy().transpose(0, 3, 1, 2), orc.quantity(refr, ob).astype(np.int8)), "int8 output"
            if st != 1:
                continue
            g_res = int(rng.integers(-1, 9)); res_dtype = np.int16 if rng.random() < 0.6 else np.int8
            g = max(0, ob, g_res)
            if g > 8:
                continue
            P, Q = ref.shape[2], ref.shape[3]
            lim = 128 * 2 ** max(g_res, 0) if res_dtype == np.int16 else 128
            res = rng.integers(-min(lim, 32768), min(lim, 32768), size=(N, P, Q, K)).astype(res_dtype)
            ib = int(rng.integers(-1, 7))
            s = orc.add_sat(ref, orc.dequantity(res.astype(np.float32), g_res).transpose(0, 3, 1, 2))
            if relu: s = np.maximum(s, np.float32(0))
            e = s.astype(np.float64) * 2.0 ** g
            if not np.all(e == np.rint(e)):
                continue
            want_w, want_n = [(True, True), (True, False), (False, True)][int(rng.integers(0, 3))]
            wide, narrow = nat.conv2d_i8_add_resident(xd, wd, bd, (st, st), (0, 0), (1, 1), rs, ob, dev(res), g_res, want_w, g, want_n, ib, relu)
            if want_w:
                assert np.array_equal(wide.cpu().numpy().transpose(0, 3, 1, 2), e.astype(np.int16)), "fused add wide (%s residual)" % res_dtype.__name__
            if want_n:
                assert np.array_equal(narrow.cpu().numpy().transpose(0, 3, 1, 2), orc.quantity(s, ib).astype(np.int8)), "fused add narrow (%s residual)" % res_dtype.__name__
        except AssertionError as ex:
            failures.append("%s -> %s" % (msg, ex))
            if verbose:
                print("MISMATCH", msg, "->", ex)
    return failures


def run_halo(cases, seed, verbose=True):
    """The resident-halo form of the 3 x 3 / stride 1 / padding 1 layers (conv3x3_i8_halo_kernel in csrc/fq_conv_i8.hip): C a
    multiple of 128, every plane size from 1 x 1 up (tiles that start and end inside an image row, span several images, or
    lie past the end), one and several channel slices, fp32 / int8 / both outputs."""
    rng = np.random.default_rng(seed)
    failures = []
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for it in range(cases):
        C = int(rng.choice([64, 64, 128, 128, 256, 384, 512])); K = int(rng.choice([64, 128, 192, 256]))
        if C == 64:                                          # the stationary-weight form of the 64-channel 3x3 layers (K <= 64)
            K = int(rng.choice([64, 64, 48, 16, 33]))
        N = int(rng.integers(1, 6)); H = int(rng.integers(1, 31)); W = int(rng.integers(1, 31))
        if C == 64 and rng.random() < 0.3:
            N, H, W = int(rng.integers(2, 9)), int(rng.integers(20, 60)), int(rng.integers(20, 60))   # several tiles per workgroup
        if rng.random() < 0.25:
            H, W = int(rng.choice([7, 14, 28])), int(rng.choice([7, 14, 28]))
        x = rng.integers(-128, 128, size=(N, C, H, W)).astype(np.int32)
        w = rng.integers(-128, 128, size=(K, C, 3, 3)).astype(np.int32)
        qb = rng.integers(-128, 128, size=K).astype(np.float32)
        rs = int(rng.integers(0, 19)); ob = int(rng.integers(-2, 7)); relu = bool(rng.integers(0, 2))
        acc = orc.conv2d_int(x, w, (1, 1), (1, 1), (1, 1))
        ref = orc.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
        xd, wd, bd = dev(x.transpose(0, 2, 3, 1).astype(np.int8)), nat.pack_weight_krsc(dev(w.astype(np.float32))), dev(qb)
        msg = "halo %d: N%d C%d H%d W%d K%d rs%d ob%d relu%d" % (it, N, C, H, W, K, rs, ob, relu)
        try:
            y = nat.conv2d_i8(xd, wd, bd, (1, 1), (1, 1), (1, 1), rs, ob).cpu().numpy()
            assert np.array_equal(y, ref), "fp32 output"
            refr = np.maximum(ref, np.float32(0)) if relu else ref
            y2, q2 = nat.conv2d_i8_resident(xd, wd, bd, (1, 1), (1, 1), (1, 1), rs, ob, True, True, relu)
            assert np.array_equal(y2.cpu().numpy(), refr), "resident fp32"
            qn = q2.cpu().numpy()                             # [N, H, W, Kpad]: channels [K, Kpad) are padding and must be zero
            assert np.array_equal(qn[..., :K].transpose(0, 3, 1, 2), orc.quantity(refr, ob).astype(np.int8)) and not qn[..., K:].any(), "resident int8"
            _, q3 = nat.conv2d_i8_resident(xd, wd, bd, (1, 1), (1, 1), (1, 1), rs, ob, False, True, relu)
            assert np.array_equal(q3.cpu().numpy(), q2.cpu().numpy()), "int8-only output"
        except AssertionError as ex:
            failures.append("%s -> %s" % (msg, ex))
            if verbose:
                print("MISMATCH", msg, "->", ex)
    return failures


def run_stem(cases, seed, verbose=True):
    """fq_conv2d_i8_stem against the oracle chain (Quantity -> integer conv -> tail -> ReLU -> next Quantity) on random
    stem-shaped layers: 1-4 input channels, kernels up to 8x8, strides 1-3, ragged images, K <= 64."""
    import torch
    rng = np.random.default_rng(seed)
    failures = []
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    done = 0
    while done < cases:
        C = int(rng.integers(1, 5)); R = int(rng.integers(1, 9)); S = int(rng.integers(1, 9))
        st = int(rng.integers(1, 4)); K = int(rng.integers(1, 65))
        ph, pw = int(rng.integers(0, R // 2 + 1)), int(rng.integers(0, S // 2 + 1))
        H, W = int(rng.integers(R, 60)), int(rng.integers(S, 75))
        N = int(rng.integers(1, 4))
        ib, rs, ob = int(rng.integers(3, 8)), int(rng.integers(1, 17)), int(rng.integers(-1, 7))
        relu = bool(rng.integers(0, 2))
        if not nat.stem_supported(C, K, R, S, (st, st), (1, 1), rs):
            continue
        done += 1
        msg = "stem N%d C%d H%d W%d K%d R%d S%d st%d pad(%d,%d) ib%d rs%d ob%d relu%d" % (N, C, H, W, K, R, S, st, ph, pw, ib, rs, ob, relu)
        try:
            x = (rng.standard_normal((N, C, H, W)) * rng.choice([0.3, 1.5, 20.0])).astype(np.float32)
            wq = rng.integers(-128, 128, size=(K, C, R, S)).astype(np.int32)
            qb = rng.integers(-128, 128, size=K).astype(np.float32)
            xq = orc.quantity(x, ib).astype(np.int32)
            # the accumulator test of the general kernels holds here too: R*S*C <= 256 taps of 2^14
            acc = orc.conv2d_int(xq, wq, (st, st), (ph, pw), (1, 1))
            ref = orc.recon_epilogue(acc.astype(np.float32), qb, rs, ob)
            if relu:
                ref = np.maximum(ref, np.float32(0))
            kpad = (K + 15) // 16 * 16
            want = np.zeros((N,) + ref.shape[2:] + (kpad,), dtype=np.int8)
            want[..., :K] = orc.quantity(ref, ob).astype(np.int8).transpose(0, 2, 3, 1)
            got = nat.conv2d_i8_stem(dev(x), nat.pack_weight_stem(dev(wq.astype(np.float32))), dev(qb), K, S, (st, st), (ph, pw),
                                     ib, rs, ob, relu).cpu().numpy()
            assert got.shape == want.shape and np.array_equal(got, want), "%d differing bytes" % int((got != want).sum())
        except AssertionError as ex:
            failures.append("%s -> %s" % (msg, ex))
            if verbose:
                print("MISMATCH", msg, "->", ex)
    return failures


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    if len(sys.argv) > 3 and sys.argv[3] == "halo":            # only the 3x3 halo kernels (FQ_HALO8=2: the eight-wave form everywhere)
        halo_fails = run_halo(n, seed)
        print("halo_fuzz: %d cases, %d mismatches (FQ_HALO8=%s FQ_HALO_STAGES=%s)"
              % (n, len(halo_fails), os.environ.get("FQ_HALO8", ""), os.environ.get("FQ_HALO_STAGES", "")))
        sys.exit(1 if halo_fails else 0)
    if len(sys.argv) > 3 and sys.argv[3] == "stream":          # only the streaming 1x1 kernel (run with FQ_CONV_STREAM=1)
        stream_fails = run_stream(n, seed)
        print("stream_fuzz: %d cases, %d mismatches (FQ_CONV_STREAM=%s FQ_STREAM_GROUPS=%s)"
              % (n, len(stream_fails), os.environ.get("FQ_CONV_STREAM", ""), os.environ.get("FQ_STREAM_GROUPS", "")))
        sys.exit(1 if stream_fails else 0)
    fails = run(n, seed)
    print("conv_fuzz: %d cases, %d mismatches" % (n, len(fails)))
    n_stem = max(20, n // 4)
    stem_fails = run_stem(n_stem, seed + 1)
    print("stem_fuzz: %d cases, %d mismatches" % (n_stem, len(stem_fails)))
    halo_fails = run_halo(max(20, n // 3), seed + 3)
    print("halo_fuzz: %d cases, %d mismatches" % (max(20, n // 3), len(halo_fails)))
    fails = fails + halo_fails
    n_stream = max(30, n // 2)
    stream_fails = run_stream(n_stream, seed + 2)
    print("stream_fuzz: %d cases, %d mismatches (FQ_STREAM_GROUPS=%s)" % (n_stream, len(stream_fails), os.environ.get("FQ_STREAM_GROUPS", "")))
    sys.exit(1 if fails or stem_fails or stream_fails else 0)
