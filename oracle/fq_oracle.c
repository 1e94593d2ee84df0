/* fq_oracle.c -- CPU restatement of the pytorch-quantity hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for libfq_hip.so.  It is NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker
 * (or as the timed CPU baseline).  The product path has no CPU fallback and never links this.
 *
 * What it restates: the reference's Python/NumPy arithmetic (lswzjuer/pytorch-quantity), one C
 * function per reference function, each citing the reference file:line it follows (paths relative
 * to the reference root, quantity/...).  The reference is pure Python, so nothing of it is compiled;
 * the third-party arithmetic it leans on (NumPy 2.2 float32/float64 promotion rules, ndarray.sum
 * pairwise order, np.around = rint, torch.round = rint, C casts) is restated here from those
 * libraries' published semantics.
 *
 * Pinning: tests/test_oracle_golden.py checks every function below against the golden vectors in
 * tests/golden/ (npz files), which were captured by IMPORTING the reference in the build container
 * (tests/golden/make_golden_*.py, committed).  Histograms, intervals, op outputs, thresholds and
 * bits must match the goldens exactly; KL values must match to <= 4 ulp (the reference's np.log is
 * SVML/AVX-512 here and is itself not correctly rounded, see include/fq_log.h).
 *
 * Plain C11, no dependencies beyond libm.  Build: make -C oracle  ->  oracle/libfq_oracle.so
 * Compile with -ffp-contract=off: every operation below must round exactly once.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fq_log.h"

#define BINS 2048
#define BINS_MAX 4096               /* INTERVAL_NUM of the _n entry points: any value in (128, 4096] */
#define TARGET 128

/* ------------------------------------------------------------------------------------------
 * A1  DistributionCollector.refresh_max_val   quantity/common/quantity/distribution_collector.py:70-78
 *     max_val = max(abs(np.max(t)), abs(np.min(t)));  running = max(running, max_val)
 *     (fp32 compares only; the running max starts at 0)
 * ------------------------------------------------------------------------------------------ */
float orc_absmax(const float* x, uint64_t n, float running) {
    if (n == 0) return running;
    float mx = x[0], mn = x[0];
    for (uint64_t i = 1; i < n; ++i) {
        if (x[i] > mx) mx = x[i];
        if (x[i] < mn) mn = x[i];
    }
    float a = fabsf(mx), b = fabsf(mn);
    float m = a > b ? a : b;
    return m > running ? m : running;
}

/* ------------------------------------------------------------------------------------------
 * A2  distribution_intervals   distribution_collector.py:52-63
 *     interval = statistic * max / interval_num + 1e-12
 *     Under NumPy 2 promotion max is np.float32 and Python scalars are weak, so every step is
 *     fp32: fl32( fl32( fl32(stat*max) / 2048 ) + fl32(1e-12) ).  For an all-zero tensor max stays
 *     the Python int 0 and the reference yields the Python float 1e-12; its fp32 image is what any
 *     later fp32-array divide would use, and is what this returns.
 * ------------------------------------------------------------------------------------------ */
float orc_interval_n(float max_val, int statistic, int bins) {       /* interval_num = bins (configs.yml:24; dc.py:9-14) */
    volatile float a = (float)statistic * max_val;
    volatile float b = a / (float)bins;
    volatile float c = b + (float)1e-12;
    return c;
}
float orc_interval(float max_val, int statistic) { return orc_interval_n(max_val, statistic, BINS); }

/* ------------------------------------------------------------------------------------------
 * A3  _add_to_distribution   distribution_collector.py:127-135, accumulated as :115-118
 *     indexes = np.minimum((abs(data[data != 0]) / interval).astype(np.int32), 2047)
 *     fp32 array / fp32 scalar = correctly rounded fp32 divide; astype(int32) truncates.
 *     hist is int64 here (the reference's int32 wraps past 2^31-1; documented deviation).
 *     Quotients >= 2048, inf and nan go to bin 2047 (the product's documented behaviour; the
 *     reference raises for the last two).
 * ------------------------------------------------------------------------------------------ */
void orc_hist_n(const float* x, uint64_t n, float interval, int64_t* hist, int bins) {   /* INTERVAL_NUM = bins (dc.py:131: interval_num - 1) */
    const float limit = (float)bins;
    for (uint64_t i = 0; i < n; ++i) {
        float v = x[i];
        if (v != 0.0f) {
            volatile float q = fabsf(v) / interval;
            int idx = (q < limit) ? (int)q : (bins - 1);
            hist[idx] += 1;
        }
    }
}
void orc_hist2048(const float* x, uint64_t n, float interval, int64_t* hist) { orc_hist_n(x, n, interval, hist, BINS); }

/* ------------------------------------------------------------------------------------------
 * NumPy pairwise summation of a contiguous float64 vector (numpy/_core/src/umath/loops_utils.h.src,
 * DOUBLE_pairwise_sum, PW_BLOCKSIZE 128): what ndarray.sum() / np.sum() do for 1-D float64.
 * ------------------------------------------------------------------------------------------ */
static double np_pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

double orc_np_sum(const double* a, int64_t n) { return np_pairwise_sum(a, n); }

/* ------------------------------------------------------------------------------------------
 * A5  Quantizer.normalize_distribution   quantizer.py:95-96
 *     distribution.astype(np.float32) / (distribution.sum() + 1e-12)
 *     int32 hist: sum is an exact int64; + 1e-12 in float64.  float64 (merged) hist: pairwise
 *     float64 sum of integers (exact below 2^53, so the order is immaterial).  fp32 array divided
 *     by a float64 scalar gives float64: (double)(float)h / denom.
 * ------------------------------------------------------------------------------------------ */
void orc_normalize_i64_n(const int64_t* hist, double* p, int bins) {
    int64_t total = 0;
    for (int j = 0; j < bins; ++j) total += hist[j];
    double denom = (double)total + 1e-12;
    for (int j = 0; j < bins; ++j) p[j] = (double)(float)hist[j] / denom;
}
void orc_normalize_i64(const int64_t* hist, double* p) { orc_normalize_i64_n(hist, p, BINS); }

void orc_normalize_f64_n(const double* hist, double* p, int bins) {
    double denom = np_pairwise_sum(hist, bins) + 1e-12;
    for (int j = 0; j < bins; ++j) p[j] = (double)(float)hist[j] / denom;
}
void orc_normalize_f64(const double* hist, double* p) { orc_normalize_f64_n(hist, p, BINS); }

/* ------------------------------------------------------------------------------------------
 * A7  compute_kl_divergence   quantizer.py:169-174
 *     nz = a != 0;  np.sum(a[nz] * np.log(a[nz] / (b[nz] + 1e-12) + 1e-12))
 *     use_fq_log = 0: libm log (closest to NumPy);  1: fq_log (bit-identical to the HIP kernel)
 * ------------------------------------------------------------------------------------------ */
static double kl_divergence(const double* a, const double* b, int n, int use_fq_log, double* scratch) {
    int m = 0;
    for (int j = 0; j < n; ++j) {
        if (a[j] != 0.0) {
            double arg = a[j] / (b[j] + 1e-12) + 1e-12;
            double lg = use_fq_log ? fq_log(arg) : log(arg);
            scratch[m++] = a[j] * lg;
        }
    }
    return np_pairwise_sum(scratch, m);
}

/* ------------------------------------------------------------------------------------------
 * A6  Quantizer.threshold_distribution   quantizer.py:98-167
 *     p: float64[2048].  Returns the threshold in [128, 2047]; kl_curve (nullable) gets the 1920
 *     divergences.  Order of every floating-point operation follows the Python source.
 * ------------------------------------------------------------------------------------------ */
/* bins = distribution.size (quantizer.py:101,:103: the sweep runs to the histogram's length, whatever INTERVAL_NUM is); bins <= BINS_MAX */
int orc_kl_threshold_n(const double* p, int bins, double* kl_curve, int use_fq_log) {
    if (bins <= TARGET || bins > BINS_MAX) return -1;
    double min_kl = 66666.0;                                 /* :99 */
    double threshold_sum = np_pairwise_sum(p + TARGET, bins - TARGET);   /* :100 */
    int target_threshold = bins - 1;                         /* :101 */
    double t_dist[BINS_MAX], q[TARGET], expand[BINS_MAX], scratch[BINS_MAX];

    for (int threshold = TARGET; threshold < bins; ++threshold) {        /* :103 */
        memcpy(t_dist, p, sizeof(double) * threshold);       /* :104 */
        t_dist[threshold - 1] += threshold_sum;              /* :105 */
        threshold_sum = threshold_sum - p[threshold];        /* :108 */

        for (int j = 0; j < threshold; ++j) expand[j] = 1e-9;            /* :111 */
        double num_per_bin = (double)threshold / (double)TARGET;         /* :112 (exact) */

        for (int i = 0; i < TARGET; ++i) {                   /* :114-126 */
            double start = (double)i * num_per_bin;
            double end = start + num_per_bin;
            int left_upper = (int)ceil(start);
            double qi = 0.0;
            if ((double)left_upper > start) {
                double left_scale = (double)left_upper - start;
                qi += left_scale * p[left_upper - 1];
            }
            int right_lower = (int)floor(end);
            if ((double)right_lower < end) {
                double right_scale = end - (double)right_lower;
                qi += right_scale * p[right_lower];
            }
            qi += np_pairwise_sum(p + left_upper, right_lower > left_upper ? right_lower - left_upper : 0);
            q[i] = qi;
        }

        for (int i = 0; i < TARGET; ++i) {                   /* :128-160 */
            double start = (double)i * num_per_bin;
            double end = start + num_per_bin;
            double count = 1e-12;
            int left_upper = (int)ceil(start);
            double left_scale = 0.0;
            if ((double)left_upper > start) {
                left_scale = (double)left_upper - start;
                if (p[left_upper - 1] != 0.0) count += left_scale;
            }
            int right_lower = (int)floor(end);
            double right_scale = 0.0;
            if ((double)right_lower < end) {
                right_scale = end - (double)right_lower;
                if (p[right_lower] != 0.0) count += right_scale;
            }
            for (int j = left_upper; j < right_lower; ++j)
                if (p[j] != 0.0) count = count + 1.0;
            double expand_value = q[i] / count;
            if ((double)left_upper > start)
                if (p[left_upper - 1] != 0.0) expand[left_upper - 1] += expand_value * left_scale;
            if ((double)right_lower < end)
                if (p[right_lower] != 0.0) expand[right_lower] += expand_value * right_scale;
            for (int j = left_upper; j < right_lower; ++j)
                if (p[j] != 0.0) expand[j] += expand_value;
        }

        double kl = kl_divergence(t_dist, expand, threshold, use_fq_log, scratch);   /* :162 */
        if (kl_curve) kl_curve[threshold - TARGET] = kl;
        if (kl < min_kl) {                                   /* :163-165 */
            min_kl = kl;
            target_threshold = threshold;
        }
    }
    return target_threshold;
}
int orc_kl_threshold(const double* p, double* kl_curve, int use_fq_log) { return orc_kl_threshold_n(p, BINS, kl_curve, use_fq_log); }

/* ------------------------------------------------------------------------------------------
 * A8  quantize_worker bits   quantizer.py:86-90
 *     threshold_bias = (threshold_bin + 0.5) * interval      -> fp32 (interval is np.float32)
 *     bit = int(8 - 1 - math.ceil(math.log(threshold_bias, 2)))
 *     CPython math.log(x, 2) = log(x) / log(2) in float64 with the C library's log.
 * ------------------------------------------------------------------------------------------ */
int orc_bits_from_threshold(int thr, float interval, float* thr_val_out) {
    volatile float tv = ((float)thr + 0.5f) * interval;
    if (thr_val_out) *thr_val_out = tv;
    double l = log((double)tv) / log(2.0);
    return (int)(8 - 1 - ceil(l));
}

/* pytorch_quantizer.py:651-653  bit = int(8 - 1 - math.ceil(math.log(max_val, 2))) */
int orc_bits_from_absmax(float absmax) {
    double l = log((double)absmax) / log(2.0);
    return (int)(8 - 1 - ceil(l));
}

/* ------------------------------------------------------------------------------------------
 * Element-wise ops   quantity/common/quantity/new_quantity_op.py
 * torch.mul / torch.div by pow(2, k): the scalar becomes fp32 2^k, the op is one fp32 rounding.
 * torch.round = round half to even (rintf under the default rounding mode).
 * torch.clamp propagates NaN.
 * ------------------------------------------------------------------------------------------ */
static float pow2f(int k) { return ldexpf(1.0f, k); }

static float clampf(float v, float lo, float hi) {
    if (v != v) return v;
    return v < lo ? lo : (v > hi ? hi : v);
}

static void range_of(int bitwidth, float* lo, float* hi) {
    if (bitwidth == 8) { *lo = -128.0f; *hi = 127.0f; } else { *lo = -32768.0f; *hi = 32767.0f; }
}

/* Quantity.forward  new_quantity_op.py:52-58 */
void orc_quantity(const float* x, float* y, uint64_t n, int ib, int bitwidth) {
    float lo, hi; range_of(bitwidth, &lo, &hi);
    float s = pow2f(ib);
    for (uint64_t i = 0; i < n; ++i) { volatile float m = x[i] * s; y[i] = clampf(rintf(m), lo, hi); }
}

/* DeQuantity.forward  :66-68 */
void orc_dequantity(const float* x, float* y, uint64_t n, int ob) {
    float s = pow2f(ob);
    for (uint64_t i = 0; i < n; ++i) { volatile float d = x[i] / s; y[i] = d; }
}

/* Sp.forward  :76-91 */
void orc_sp(const float* x, float* y, uint64_t n, int bitwidth) {
    float lo, hi; range_of(bitwidth, &lo, &hi);
    for (uint64_t i = 0; i < n; ++i) y[i] = clampf(x[i], lo, hi);
}

/* RightShift.forward  :17-44: v = x / 2^rs; r = (int32)(v + (v > 0 ? 0.5 : -0.5)); clamp; float.
 * The fp32 add rounds (matters only beyond 2^23, i.e. far outside the clamp range).  The int32
 * cast saturates here; torch's CPU cast yields INT_MIN for |v| >= 2^31, unreachable for int8 data. */
void orc_rightshift(const float* x, float* y, uint64_t n, int rs, int bitwidth) {
    float lo, hi; range_of(bitwidth, &lo, &hi);
    float s = pow2f(rs);
    for (uint64_t i = 0; i < n; ++i) {
        volatile float v = x[i] / s;
        volatile float w = v + (v > 0.0f ? 0.5f : -0.5f);
        float t = truncf(w);
        if (t < -2147483648.0f) t = -2147483648.0f;
        if (t > 2147483520.0f) t = 2147483520.0f;
        int32_t r = (int32_t)t;
        int32_t c = r < (int32_t)lo ? (int32_t)lo : (r > (int32_t)hi ? (int32_t)hi : r);
        y[i] = (float)c;
    }
}

/* NewAdd.forward  :171-174 */
void orc_add_sat(const float* a, const float* b, float* y, uint64_t n, int bitwidth) {
    float lo, hi; range_of(bitwidth, &lo, &hi);
    for (uint64_t i = 0; i < n; ++i) { volatile float s = a[i] + b[i]; y[i] = clampf(s, lo, hi); }
}

/* QuanDequan.forward  :246-257 */
void orc_quandequan(const float* x, float* y, uint64_t n, int bit, int bitwidth) {
    float lo, hi; range_of(bitwidth, &lo, &hi);
    float s = pow2f(bit);
    for (uint64_t i = 0; i < n; ++i) {
        volatile float m = x[i] * s;
        float c = clampf(rintf(m), lo, hi);
        volatile float d = c / s;
        y[i] = d;
    }
}

/* NewConv2d.forward tail  :127-132  RightShift -> BiasAdd -> Sp -> DeQuantity on acc[outer][C][inner] */
void orc_recon_epilogue(const float* acc, const float* qbias, float* y, uint64_t outer, uint64_t C,
                        uint64_t inner, int rs, int ob, int bitwidth) {
    float lo, hi; range_of(bitwidth, &lo, &hi);
    float so = pow2f(ob);
    for (uint64_t o = 0; o < outer; ++o)
        for (uint64_t c = 0; c < C; ++c) {
            const float* src = acc + (o * C + c) * inner;
            float* dst = y + (o * C + c) * inner;
            orc_rightshift(src, dst, inner, rs, bitwidth);
            for (uint64_t i = 0; i < inner; ++i) {
                volatile float s = dst[i] + qbias[c];
                float cl = clampf(s, lo, hi);
                volatile float d = cl / so;
                dst[i] = d;
            }
        }
}

/* weight quantiser  quantity/tools/pytorch_quantizer.py:656-657,663
 *   np.clip(np.around(w * math.pow(2, bit)), -128, 127).astype(np.int32)
 *   (fp32 array * Python float -> fp32; np.around = rint) */
void orc_quantize_param_i32(const float* w, int32_t* q, uint64_t n, int bit) {
    float s = pow2f(bit);
    for (uint64_t i = 0; i < n; ++i) {
        volatile float m = w[i] * s;
        float r = rintf(m);
        if (r < -128.0f) r = -128.0f;
        if (r > 127.0f) r = 127.0f;
        q[i] = (int32_t)r;
    }
}

/* ------------------------------------------------------------------------------------------
 * NewConv2d / NewLinear integer contraction, new_quantity_op.py:124-133 with :147-163.
 * The reference runs an fp32 conv over integer-valued tensors; below 2^24 per partial sum that is
 * exact integer arithmetic, which is what this computes (int64 accumulator, no rounding at all).
 * x: int8-range values as int32 [N][C][H][W]; w: [K][C/groups][R][S]; out acc int64 [N][K][P][Q].
 * ------------------------------------------------------------------------------------------ */
void orc_conv2d_int(const int32_t* x, const int32_t* w, int64_t* acc, int N, int C, int H, int W,
                    int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                    int dil_h, int dil_w, int groups) {
    int P = (H + 2 * pad_h - dil_h * (R - 1) - 1) / stride_h + 1;
    int Q = (W + 2 * pad_w - dil_w * (S - 1) - 1) / stride_w + 1;
    int Cg = C / groups, Kg = K / groups;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            int g = k / Kg;
            for (int p = 0; p < P; ++p)
                for (int qq = 0; qq < Q; ++qq) {
                    int64_t s = 0;
                    for (int c = 0; c < Cg; ++c)
                        for (int r = 0; r < R; ++r) {
                            int ih = p * stride_h - pad_h + r * dil_h;
                            if (ih < 0 || ih >= H) continue;
                            for (int ss = 0; ss < S; ++ss) {
                                int iw = qq * stride_w - pad_w + ss * dil_w;
                                if (iw < 0 || iw >= W) continue;
                                s += (int64_t)x[((n * C + g * Cg + c) * H + ih) * W + iw] *
                                     (int64_t)w[((k * Cg + c) * R + r) * S + ss];
                            }
                        }
                    acc[((int64_t)(n * K + k) * P + p) * Q + qq] = s;
                }
        }
}

int orc_version(void) { return 100; }
