/* Test harness (tests/test_fq_log.py): fq_log (fast path + Ziv test) against fq_log_dd (double-double, correctly
 * rounded) on many arguments.  Prints: checked <n> mismatches <n> fallbacks <n>. */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#define FQ_LOG_STATS
#include "../include/fq_log.h"

static uint64_t s[2] = {0x9e3779b97f4a7c15ULL, 0xbf58476d1ce4e5b9ULL};
static uint64_t rnd(void) {                       /* xorshift128+ */
    uint64_t a = s[0], b = s[1];
    s[0] = b; a ^= a << 23; a ^= a >> 17; a ^= b ^ (b >> 26); s[1] = a;
    return a + b;
}
static double from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 1000000;
    long bad = 0, checked = 0;
    /* 1. random bit patterns over all positive finite doubles */
    for (long i = 0; i < n; ++i) {
        uint64_t u = rnd() & 0x7fffffffffffffffULL;
        if ((u >> 52) == 0x7ff) continue;
        double x = from_bits(u);
        if (x == 0.0) continue;
        double a = fq_log(x), b = fq_log_dd(x);
        ++checked;
        if (memcmp(&a, &b, 8)) { if (bad < 10) printf("MISMATCH x=%a fast=%a dd=%a\n", x, a, b); ++bad; }
    }
    /* 2. the range the KL sweep lives in: a/(b+1e-12)+1e-12 with a, b in (1e-9, 1] -> around 1, log-uniform */
    for (long i = 0; i < n; ++i) {
        double e = (double)(int64_t)(rnd() % 4000001) / 100000.0 - 20.0;          /* exponent in [-20, 20] */
        double x = exp2(e) * (1.0 + (double)(rnd() >> 11) * 0x1p-53);
        double a = fq_log(x), b = fq_log_dd(x);
        ++checked;
        if (memcmp(&a, &b, 8)) { if (bad < 10) printf("MISMATCH x=%a fast=%a dd=%a\n", x, a, b); ++bad; }
    }
    /* 3. neighbourhood of 1 (where log cancels) and of every table boundary */
    for (long i = 0; i < n; ++i) {
        int sh = (int)(rnd() % 52);
        double d = ldexp((double)(rnd() >> 11) * 0x1p-53, -sh);                   /* (0, 2^-sh) */
        double x = (rnd() & 1) ? 1.0 + d : 1.0 - d * 0.5;
        double a = fq_log(x), b = fq_log_dd(x);
        ++checked;
        if (memcmp(&a, &b, 8)) { if (bad < 10) printf("MISMATCH x=%a fast=%a dd=%a\n", x, a, b); ++bad; }
    }
    for (int i = 0; i <= 128; ++i) {
        uint64_t edge = 0x3fe6000000000000ULL + ((uint64_t)i << 45);
        for (int d = -2000; d <= 2000; ++d) {
            double x = from_bits(edge + (uint64_t)(int64_t)d);
            for (int k = -3; k <= 3; ++k) {
                double xs = ldexp(x, k * 17);
                double a = fq_log(xs), b = fq_log_dd(xs);
                ++checked;
                if (memcmp(&a, &b, 8)) { if (bad < 10) printf("MISMATCH x=%a fast=%a dd=%a\n", xs, a, b); ++bad; }
            }
        }
    }
    /* 4. specials */
    const double sp[] = {1.0, 0x1p-1074, 0x1p-1022, 0x1.fffffffffffffp+1023, 2.0, 0.5, 0x1.0000000000001p+0, 0x1.fffffffffffffp-1};
    for (unsigned i = 0; i < sizeof sp / sizeof sp[0]; ++i) {
        double a = fq_log(sp[i]), b = fq_log_dd(sp[i]);
        ++checked;
        if (memcmp(&a, &b, 8)) { printf("MISMATCH special x=%a fast=%a dd=%a\n", sp[i], a, b); ++bad; }
    }
    double z = fq_log(0.0), ng = fq_log(-1.0), inf = fq_log(INFINITY);
    if (!(isinf(z) && z < 0) || !isnan(ng) || !(isinf(inf) && inf > 0)) { printf("MISMATCH specials\n"); ++bad; }
    printf("checked %ld mismatches %ld fallbacks %ld\n", checked, bad, fq_log_fallbacks);
    return bad ? 1 : 0;
}
