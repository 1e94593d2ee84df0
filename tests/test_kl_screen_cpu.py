"""The closed form behind FQ_KL_SCREENED (csrc/fq_kl.hip: kl_screen_kernel), restated in NumPy and pinned against the
exact CPU oracle: for every golden (G2) and fuzz histogram and every candidate threshold t,

    |S(t) - KL(t)| < 2e-13        (the kernel keeps candidates within 1e-10 of min S: 500 x that),
    KL(t) is NaN  =>  S(t) is NaN (a NaN is always kept, so the exact pass reproduces it),

hence the exact minimum is always among the survivors and the screened search returns the exhaustive search's
threshold.  The GPU kernel itself is compared with the exhaustive kernel in tests/test_gpu_kernels.py."""
import os
import sys

import numpy as np
import pytest

import cases

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

MARGIN = 1e-10
BOUND = 2e-13


def screen_curve(P):
    """S(t), t = 128..2047, the way kl_screen_kernel evaluates it (prefix sums in extended precision standing in for the
    kernel's double-double)."""
    P = np.asarray(P, dtype=np.float64)
    nzb = P != 0
    ext = np.longdouble
    SP = np.concatenate([[ext(0)], np.cumsum(P.astype(ext))])
    NZ = np.concatenate([[0], np.cumsum(nzb)])
    plogp = np.where(nzb, P * np.log(np.where(nzb, P, 1.0)), 0.0)
    AL = np.concatenate([[ext(0)], np.cumsum(plogp.astype(ext))])
    tail = np.zeros(2049)
    ts = np.sum(P[128:])                            # quantizer.py:100 (pairwise), then the subtraction chain of :108
    tail[128] = ts
    for t in range(128, 2047):
        ts = ts - P[t]
        tail[t + 1] = ts
    out = np.empty(1920)
    i = np.arange(128)

    def mass(lo, hi):
        return (SP[hi] - SP[lo]).astype(np.float64)

    for c in range(1920):
        t = 128 + c
        npb = t / 128.0
        start = i * npb
        end = start + npb
        lu = np.ceil(start).astype(int)
        rl = np.floor(end).astype(int)
        has_l, has_r = lu > start, rl < end
        ls = np.where(has_l, lu - start, 0.0)
        rs = np.where(has_r, end - rl, 0.0)
        pl = np.where(has_l, P[np.maximum(lu - 1, 0)], 0.0)
        pr = np.where(has_r, P[np.minimum(rl, 2047)], 0.0)
        q = ls * pl + rs * pr + mass(lu, rl)
        count = 1e-12 + (NZ[rl] - NZ[lu]) + np.where(pl != 0, ls, 0.0) + np.where(pr != 0, rs, 0.0)
        ev = q / count
        hi = rl.copy()
        hi[127] = t - 1                               # the folded last bin is handled apart
        nint = NZ[hi] - NZ[lu]
        with np.errstate(all="ignore"):
            eint = (1e-9 + ev) + 1e-12
            part = np.where(nint > 0, mass(lu, hi) * np.log(eint), 0.0).sum()
            corr = np.where(nint > 0, nint * eint, 0.0).sum()
            evn = np.concatenate([ev[1:], [0.0]])
            redge = has_r & (pr != 0)
            redge[127] = False
            eedge = ((1e-9 + ev * rs) + evn * (1.0 - rs)) + 1e-12
            part += np.where(redge, pr * np.log(eedge), 0.0).sum()
            corr += np.where(redge, eedge, 0.0).sum()
            a_last = P[t - 1] + tail[t]
            s = float(AL[t - 1]) - part
            if a_last != 0.0:
                e_last = ((1e-9 + ev[127]) if P[t - 1] != 0 else 1e-9) + 1e-12
                s += (a_last * np.log(a_last) - a_last * np.log(e_last)) if a_last > 0 else float("nan")
                corr += e_last
            out[c] = s + 1e-12 * corr
    return out


def _check(oracle, h):
    p = oracle.normalize(np.asarray(h))
    thr, kl = oracle.kl_threshold(p, want_curve=True, use_fq_log=True)
    s = screen_curve(p)
    assert np.all(np.isnan(s)[np.isnan(kl)])
    both = np.isfinite(kl) & np.isfinite(s)
    err = float(np.max(np.abs(kl[both] - s[both]))) if both.any() else 0.0
    assert err < BOUND, err
    finite = np.isfinite(s)
    mn = s[finite].min() if finite.any() else np.inf
    keep = ~(s > mn + MARGIN + 1e-12 * abs(mn))
    # the exhaustive search's winner (first strict minimum below 66666, NaN never wins) is a survivor
    valid = np.where(np.isnan(kl), np.inf, kl)
    if valid.min() < 66666.0:
        assert keep[int(np.argmin(valid))] and 128 + int(np.argmin(valid)) == thr
    return err, int(keep.sum())


@pytest.mark.parametrize("name", list(cases.g2_cases().keys()))
def test_closed_form_tracks_the_exact_kl_on_the_goldens(oracle, name):
    _check(oracle, cases.g2_cases()[name])


def test_closed_form_tracks_the_exact_kl_on_fuzz_histograms(oracle):
    import kl_fuzz_hist
    rng = np.random.default_rng(3)
    worst = 0.0
    for _ in range(24):
        err, _kept = _check(oracle, kl_fuzz_hist.random_histogram(rng))
        worst = max(worst, err)
    assert worst < BOUND
