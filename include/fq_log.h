/* fq_log.h -- the natural logarithm used inside fq_kl_threshold (include/fq.h).
 *
 * Why a private log: the KL sweep (reference quantity/common/quantity/quantizer.py:169-174) calls
 * np.log on float64.  NumPy's log, glibc's log and the GPU's ocml log are each faithful (< 1 ulp)
 * but not bit-identical to one another, so no device log can match "the" reference log bit for bit.
 * fq_log is written with nothing but IEEE-754 binary64 add / multiply / fma and integer operations,
 * in one fixed order, so that the SAME source gives the SAME bits when compiled by gcc for the host
 * (oracle/, KL curve checks) and by hipcc for gfx950 (the product kernel).  It evaluates log(x) in
 * double-double (~100 bits) and rounds once, i.e. it is correctly rounded except in astronomically
 * rare hard cases, which also makes it agree with glibc/NumPy in all but the ~1e-4 of arguments
 * where those libraries themselves are not correctly rounded.
 *
 * Contract: x finite and > 0 (normal or subnormal).  The KL sweep only ever calls it with
 * a/(b+1e-12)+1e-12 where a > 0, b > 0, so zero / negative / inf / nan never occur; for
 * completeness, as np.log: x < 0 or nan returns nan, +-0 returns -inf, +inf returns +inf.
 *
 * Compile with -ffp-contract=off (both compilers): every fma below is explicit.
 */
#ifndef FQ_LOG_H
#define FQ_LOG_H

#include <stdint.h>
#include <string.h>

#include "fq_log_table.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define FQ_LOG_FN __device__ static inline
#define FQ_LOG_TABLE __device__ static const
#else
#define FQ_LOG_FN static inline
#define FQ_LOG_TABLE static const
#endif

#define FQ_FMA(a, b, c) __builtin_fma((a), (b), (c))

typedef struct { double hi, lo; } fq_dd;

FQ_LOG_FN fq_dd fq_two_sum(double a, double b) {
    fq_dd r;
    double s = a + b;
    double bb = s - a;
    r.hi = s;
    r.lo = (a - (s - bb)) + (b - bb);
    return r;
}

FQ_LOG_FN fq_dd fq_fast_two_sum(double a, double b) { /* |a| >= |b| */
    fq_dd r;
    double s = a + b;
    r.hi = s;
    r.lo = b - (s - a);
    return r;
}

FQ_LOG_FN fq_dd fq_two_prod(double a, double b) {
    fq_dd r;
    r.hi = a * b;
    r.lo = FQ_FMA(a, b, -r.hi);
    return r;
}

FQ_LOG_FN fq_dd fq_dd_add(fq_dd a, fq_dd b) {
    fq_dd s = fq_two_sum(a.hi, b.hi);
    fq_dd t = fq_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = fq_fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return fq_fast_two_sum(s.hi, s.lo);
}

FQ_LOG_FN fq_dd fq_dd_add_d(fq_dd a, double b) {
    fq_dd s = fq_two_sum(a.hi, b);
    s.lo += a.lo;
    return fq_fast_two_sum(s.hi, s.lo);
}

FQ_LOG_FN fq_dd fq_dd_mul(fq_dd a, fq_dd b) {
    fq_dd p = fq_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo;
    p.lo += a.lo * b.hi;
    return fq_fast_two_sum(p.hi, p.lo);
}

FQ_LOG_FN fq_dd fq_dd_mul_d(fq_dd a, double b) {
    fq_dd p = fq_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return fq_fast_two_sum(p.hi, p.lo);
}

/* a / b for doubles a, b -> double-double quotient (b != 0) */
FQ_LOG_FN fq_dd fq_dd_div_dd(fq_dd a, fq_dd b) {
    double q1 = a.hi / b.hi;
    /* r = a - q1*b */
    fq_dd p = fq_dd_mul_d(b, q1);
    fq_dd r;
    r.hi = a.hi - p.hi;          /* exact-ish leading cancellation */
    r.lo = ((a.hi - p.hi) - r.hi) + (a.lo - p.lo);
    double q2 = (r.hi + r.lo) / b.hi;
    p = fq_dd_mul_d(b, q2);
    double r2 = ((r.hi - p.hi) + (r.lo - p.lo));
    double q3 = r2 / b.hi;
    fq_dd q = fq_fast_two_sum(q1, q2);
    q.lo += q3;
    return fq_fast_two_sum(q.hi, q.lo);
}

/* The double-double evaluation: correctly rounded, ~450 flops.  fq_log below reaches the same bits ~10x cheaper
 * and calls this only when its own error bound cannot decide the rounding. */
FQ_LOG_FN double fq_log_dd(double x) {
    uint64_t ux;
    memcpy(&ux, &x, 8);
    if (ux == 0x7ff0000000000000ULL) return x;                       /* +inf */
    if ((ux << 1) == 0) {                                            /* log(+-0) = -inf, as np.log */
        uint64_t ni = 0xfff0000000000000ULL;
        double r;
        memcpy(&r, &ni, 8);
        return r;
    }
    if ((ux >> 63) || (ux & 0x7fffffffffffffffULL) > 0x7ff0000000000000ULL) {
        uint64_t qn = 0x7ff8000000000000ULL;
        double r;
        memcpy(&r, &qn, 8);
        return r;
    }
    int e = 0;
    if ((ux >> 52) == 0) {                                           /* subnormal: scale by 2^54 */
        x = x * 18014398509481984.0;
        memcpy(&ux, &x, 8);
        e = -54;
    }
    e += (int)(ux >> 52) - 1023;
    uint64_t um = (ux & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &um, 8);                                              /* m in [1, 2) */
    if (m > 1.4142135623730951) {                                    /* m in [sqrt2/2, sqrt2) */
        m = m * 0.5;
        e += 1;
    }
    /* log(m) = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716.  m-1 is exact; m+1 as double-double. */
    fq_dd num; num.hi = m - 1.0; num.lo = 0.0;
    fq_dd den = fq_two_sum(m, 1.0);
    fq_dd s = fq_dd_div_dd(num, den);
    fq_dd s2 = fq_dd_mul(s, s);
    /* series sum_{k>=0} s^(2k)/(2k+1); |s2| <= 0.02944 -> 2^-5.09 per term.
     * Terms k >= 10 are evaluated in plain double (their value is < 2^-55 of the sum, their own
     * rounding error is < 2^-107 relative to the sum); terms k < 10 in double-double.
     * Truncation after k = 33: s2^34/69 < 2^-179. */
    double z = s2.hi;
    double tail = 1.0 / 67.0;
    tail = FQ_FMA(tail, z, 1.0 / 65.0);
    tail = FQ_FMA(tail, z, 1.0 / 63.0);
    tail = FQ_FMA(tail, z, 1.0 / 61.0);
    tail = FQ_FMA(tail, z, 1.0 / 59.0);
    tail = FQ_FMA(tail, z, 1.0 / 57.0);
    tail = FQ_FMA(tail, z, 1.0 / 55.0);
    tail = FQ_FMA(tail, z, 1.0 / 53.0);
    tail = FQ_FMA(tail, z, 1.0 / 51.0);
    tail = FQ_FMA(tail, z, 1.0 / 49.0);
    tail = FQ_FMA(tail, z, 1.0 / 47.0);
    tail = FQ_FMA(tail, z, 1.0 / 45.0);
    tail = FQ_FMA(tail, z, 1.0 / 43.0);
    tail = FQ_FMA(tail, z, 1.0 / 41.0);
    tail = FQ_FMA(tail, z, 1.0 / 39.0);
    tail = FQ_FMA(tail, z, 1.0 / 37.0);
    tail = FQ_FMA(tail, z, 1.0 / 35.0);
    tail = FQ_FMA(tail, z, 1.0 / 33.0);
    tail = FQ_FMA(tail, z, 1.0 / 31.0);
    tail = FQ_FMA(tail, z, 1.0 / 29.0);
    tail = FQ_FMA(tail, z, 1.0 / 27.0);
    tail = FQ_FMA(tail, z, 1.0 / 25.0);
    tail = FQ_FMA(tail, z, 1.0 / 23.0);
    tail = FQ_FMA(tail, z, 1.0 / 21.0);
    /* double-double Horner for k = 9 .. 0 with coefficients 1/(2k+1) as double-double constants */
    fq_dd acc; acc.hi = tail; acc.lo = 0.0;
    /* 1/(2k+1) as double-double: hi = RN(1/d), lo = RN(1/d - hi) (generated with exact rationals;
     * tests/test_fq_log.py re-derives them). */
#define FQ_LOG_STEP(HI, LO) do { fq_dd c_; c_.hi = (HI); c_.lo = (LO); \
        acc = fq_dd_mul(acc, s2); acc = fq_dd_add(acc, c_); } while (0)
    FQ_LOG_STEP(0x1.af286bca1af28p-5, 0x1.af286bca1af28p-59);    /* 1/19 */
    FQ_LOG_STEP(0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61);    /* 1/17 */
    FQ_LOG_STEP(0x1.1111111111111p-4, 0x1.1111111111111p-60);    /* 1/15 */
    FQ_LOG_STEP(0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58);   /* 1/13 */
    FQ_LOG_STEP(0x1.745d1745d1746p-4, -0x1.745d1745d1746p-59);   /* 1/11 */
    FQ_LOG_STEP(0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58);    /* 1/9 */
    FQ_LOG_STEP(0x1.2492492492492p-3, 0x1.2492492492492p-57);    /* 1/7 */
    FQ_LOG_STEP(0x1.999999999999ap-3, -0x1.999999999999ap-57);   /* 1/5 */
    FQ_LOG_STEP(0x1.5555555555555p-2, 0x1.5555555555555p-56);    /* 1/3 */
    FQ_LOG_STEP(1.0, 0.0);                                       /* 1/1 */
#undef FQ_LOG_STEP
    fq_dd lm = fq_dd_mul(acc, s);
    lm.hi *= 2.0; lm.lo *= 2.0;                                      /* log(m), exact scaling */
    /* e * ln2 in double-double: ln2 = LN2_HI + LN2_LO (+ 2^-110), LN2_HI has 11 trailing zero bits
     * so e*LN2_HI is exact for |e| < 2^11. */
    const double LN2_HI = 0x1.62e42fefa38p-1;      /* 0.693147180559890330187045037746429443359375 */
    const double LN2_MD = 0x1.ef35793c76p-45;      /* next 42 bits */
    const double LN2_LO = 0x1.cc01f97b57a08p-87;   /* remainder, rounded */
    double de = (double)e;
    fq_dd t; t.hi = de * LN2_HI; t.lo = 0.0;                         /* exact */
    fq_dd t2; t2.hi = de * LN2_MD; t2.lo = 0.0;                      /* exact (42-bit * 11-bit) */
    fq_dd r = fq_dd_add(t, t2);
    r = fq_dd_add_d(r, de * LN2_LO);
    r = fq_dd_add(r, lm);
    return r.hi + r.lo;
}

/* ---- fast path ----------------------------------------------------------------------------------
 * x = 2^k * z with z in [0.6875, 1.375) (k = 0 around 1, so nothing cancels there); the top 7 bits of z's
 * offset select a table slice with invc ~ 1/centre and log(1/invc) as a double-double (fq_log_table.h):
 *     log(x) = k ln2 + log(1/invc) + log1p(r),   r = z * invc - 1   (|r| < 2^-7, held as r_hi + r_lo, exact)
 *     log1p(r) = r - r^2/2 + r^3 (1/3 - r/4 + ... - r^7/10)
 * r^2 is a two-product, the cubic part is plain double (< 2^-22, its rounding error < 2^-75), every other sum
 * keeps its low word: the value hi + lo carries a relative error below 2^-65.6 (step-by-step bound in the
 * comments of tests/test_fq_log.py).  Ziv's test: if hi + (lo +- hi * 2^-64) round to the same double, that
 * double is the correctly rounded log -- the same bits fq_log_dd gives -- and it is returned; otherwise
 * (about one argument in two thousand) fq_log_dd decides.  tests/test_fq_log.py compares the two on millions of
 * arguments, including the neighbourhood of 1, subnormals and the table boundaries. */
#ifdef FQ_LOG_STATS
static long fq_log_fallbacks = 0;
#endif
typedef struct { double invc, logc_hi, logc_lo; } fq_log_row;
FQ_LOG_TABLE fq_log_row fq_log_table[128] = { FQ_LOG_TABLE_ROWS };

FQ_LOG_FN double fq_log(double x) {
    uint64_t ix;
    memcpy(&ix, &x, 8);
    /* zero, negative, inf, nan and subnormals: the reference path knows them all */
    if (ix - 0x0010000000000000ULL >= 0x7ff0000000000000ULL - 0x0010000000000000ULL) return fq_log_dd(x);
    const uint64_t tmp = ix - 0x3fe6000000000000ULL;
    const int i = (int)((tmp >> 45) & 127);
    const int64_t k = (int64_t)tmp >> 52;                             /* arithmetic shift */
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ULL);
    double z;
    memcpy(&z, &iz, 8);
    const double invc = fq_log_table[i].invc, lch = fq_log_table[i].logc_hi, lcl = fq_log_table[i].logc_lo;
    /* r = z * invc - 1 exactly, as r_hi + r_lo */
    const fq_dd p = fq_two_prod(z, invc);
    const double t = p.hi - 1.0;                                      /* exact: p.hi is within [0.98, 1.02] */
    const fq_dd r = fq_two_sum(t, p.lo);
    const double rh = r.hi;
    /* log1p(r) - r */
    const fq_dd r2 = fq_two_prod(rh, rh);
    double c = -1.0 / 10.0;
    c = FQ_FMA(c, rh, 1.0 / 9.0);
    c = FQ_FMA(c, rh, -1.0 / 8.0);
    c = FQ_FMA(c, rh, 1.0 / 7.0);
    c = FQ_FMA(c, rh, -1.0 / 6.0);
    c = FQ_FMA(c, rh, 1.0 / 5.0);
    c = FQ_FMA(c, rh, -1.0 / 4.0);
    c = FQ_FMA(c, rh, 1.0 / 3.0);
    const double cubic = (r2.hi * rh) * c;                            /* r^3 (1/3 - r/4 + ...) */
    /* r_lo enters to first order only: d/dr log1p(r) = 1/(1 + r) ~ 1 - r */
    const double small = ((-0.5 * r2.lo) + cubic) + (r.lo - rh * r.lo);
    /* k ln2 + logc */
    const double LN2_HI = 0x1.62e42fefa38p-1, LN2_MD = 0x1.ef35793c76p-45, LN2_LO = 0x1.cc01f97b57a08p-87;
    const double dk = (double)k;
    const fq_dd w = fq_two_sum(dk * LN2_HI, lch);                     /* dk * LN2_HI is exact */
    const double wlo = w.lo + (dk * LN2_MD + (lcl + dk * LN2_LO));
    /* hi + lo = w + r_hi - r2_hi / 2 + small terms */
    const fq_dd s1 = fq_two_sum(w.hi, rh);
    const fq_dd s2 = fq_two_sum(s1.hi, -0.5 * r2.hi);
    const double lo = s2.lo + (s1.lo + (wlo + small));
    const fq_dd res = fq_fast_two_sum(s2.hi, lo);                     /* |lo| << |s2.hi| */
    const double err = (res.hi < 0.0 ? -res.hi : res.hi) * 0x1p-64;
    const double up = res.hi + (res.lo + err), dn = res.hi + (res.lo - err);
    if (up == dn) return up;
#ifdef FQ_LOG_STATS                                   /* test harness only: how often the bound cannot decide */
    ++fq_log_fallbacks;
#endif
    return fq_log_dd(x);
}

#endif /* FQ_LOG_H */
