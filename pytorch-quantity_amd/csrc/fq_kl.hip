// fq_kl.hip -- the KL-divergence threshold sweep on gfx950.
//
// Replaces Quantizer.normalize_distribution / threshold_distribution / compute_kl_divergence
// (quantity/common/quantity/quantizer.py:95-174): for every histogram row and every candidate
// threshold t in [128, 2047] build the 128-level quantised distribution, expand it back over the
// non-empty bins, and take KL(P[:t] || expanded).  The reference spends ~2.3 s per tensor in
// interpreted loops; here the 1920 candidates of every row are independent workgroups.
//
// Bit-exactness contract: all arithmetic is IEEE binary64 in the reference's operation order
// (compiled with -ffp-contract=off), sums follow NumPy's pairwise order, the incremental tail
// "threshold_sum" is the same sequential subtraction chain, and log is include/fq_log.h (the same
// source the CPU oracle can run).  Latency/LDS bound, tiny data (rows x 16 KB): reported in ms.
//
//   kl_prepare_kernel  grid = rows          P = fp32(hist)/(sum+1e-12); tail[t] chain
//   kl_sweep_kernel    grid = (1920, rows)  KL(t)
//   kl_argmin_kernel   grid = rows          first strict minimum below 66666, else 2047
//
// Many rows (the per-channel extension: 42 667 rows for ResNet-50) make the sweep throughput bound: 82 M candidates x
// ~1 100 logarithm + divide pairs in correctly rounded float64 is ~7e12 fp64 operations, 0.67 s.  The argmin does not
// need all of them exactly: the SCREENED path first evaluates every candidate through a closed form that costs 128
// bins instead of t elements (kl_screen_kernel, below: plain fp64, |S(t) - KL(t)| < FQ_KL_SCREEN_BOUND = 2e-13), keeps
// only the candidates within FQ_KL_SCREEN_MARGIN = 1e-10 of the smallest S(t) -- the true minimum is provably among them -- and runs the
// exact, bit-for-bit evaluation on those alone (kl_exact_list_kernel, typically 1-3 per row).  Same thresholds by
// construction; tests/test_gpu_kernels.py checks the bound on every golden and fuzz histogram.
#include <cstdlib>

#include "fq_common.h"
#include "../../include/fq_log.h"

namespace fq {

constexpr int kKlBlock = 256;
constexpr int kTarget = FQ_KL_TARGET_BINS;     // 128
constexpr int kCand = FQ_KL_CANDIDATES;        // 1920
constexpr int kMaxLeaves = 40;

// ---- NumPy pairwise summation (numpy/_core/src/umath/loops_utils.h.src, PW_BLOCKSIZE 128) ------
__device__ __forceinline__ int pw_half(int n) { int n2 = n >> 1; return n2 - (n2 & 7); }

__device__ double pw_leaf(const double* a, int n) {          // n <= 128
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    const int lim = n - (n & 7);
    int i;
    for (i = 8; i < lim; i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}

// pw_leaf for a slice of at most 16 elements that also reports which of them are non-zero (bit i = a[i] != 0):
// the quantised-bin phase needs the sum, the count and, later, the positions -- one pass over LDS instead of
// three (adjacent lanes read slices 128 bytes apart: every pass is a 32-way bank conflict).
__device__ __forceinline__ double pw_leaf_nz(const double* a, int n, unsigned* nz) {          // n <= 16
    unsigned m = 0;
    double res;
    if (n < 8) {
        res = 0.0;
        for (int i = 0; i < n; ++i) { const double v = a[i]; res += v; m |= (v != 0.0 ? 1u : 0u) << i; }
    } else {
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { r[k] = a[k]; m |= (r[k] != 0.0 ? 1u : 0u) << k; }
        const int lim = n - (n & 7);                      // 8 or 16
        if (lim == 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const double v = a[8 + k]; r[k] += v; m |= (v != 0.0 ? 1u : 0u) << (8 + k); }
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = lim; i < n; ++i) { const double v = a[i]; res += v; m |= (v != 0.0 ? 1u : 0u) << i; }
    }
    *nz = m;
    return res;
}

// ... and for slices of up to 32 elements (INTERVAL_NUM 4096: 4096 / 128 bins per quantised bin)
__device__ __forceinline__ double pw_leaf_nz32(const double* a, int n, unsigned* nz) {          // n <= 32
    unsigned m = 0;
    double res;
    if (n < 8) {
        res = 0.0;
        for (int i = 0; i < n; ++i) { const double v = a[i]; res += v; m |= (v != 0.0 ? 1u : 0u) << i; }
    } else {
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { r[k] = a[k]; m |= (r[k] != 0.0 ? 1u : 0u) << k; }
        const int lim = n - (n & 7);                      // 8, 16, 24 or 32
        for (int i = 8; i < lim; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const double v = a[i + k]; r[k] += v; m |= (v != 0.0 ? 1u : 0u) << (i + k); }
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = lim; i < n; ++i) { const double v = a[i]; res += v; m |= (v != 0.0 ? 1u : 0u) << i; }
    }
    *nz = m;
    return res;
}

// Walks the recursion tree of pairwise_sum(n) in order; leaf(off, len) supplies each leaf's value.
template <typename LeafFn>
__device__ double pw_walk(int n, LeafFn&& leaf) {
    int so[8], sn[8];
    double sv[8];
    unsigned char st[8];
    int sp = 0;
    so[0] = 0; sn[0] = n; st[0] = 0;
    double result = 0.0;
    while (true) {
        if (sn[sp] <= 128) {
            result = leaf(so[sp], sn[sp]);
            while (true) {
                if (sp == 0) return result;
                --sp;
                if (st[sp] == 0) {                       // back from the left child: descend right
                    sv[sp] = result;
                    st[sp] = 1;
                    const int n2 = pw_half(sn[sp]);
                    so[sp + 1] = so[sp] + n2; sn[sp + 1] = sn[sp] - n2; st[sp + 1] = 0;
                    ++sp;
                    break;
                }
                result = sv[sp] + result;                // back from the right child
            }
        } else {
            const int n2 = pw_half(sn[sp]);
            so[sp + 1] = so[sp]; sn[sp + 1] = n2; st[sp + 1] = 0;
            ++sp;
        }
    }
}

__device__ double pw_sum_serial(const double* a, int n) {
    return pw_walk(n, [&](int off, int len) { return pw_leaf(a + off, len); });
}

// ---- prepare: normalise, tail chain -------------------------------------------------------------
// B = INTERVAL_NUM = the histogram's length (quantizer.py:98-167 is generic in distribution.size): 2048 as shipped; 512 / 1024 / 4096
// through fq_kl_threshold_n
template <int B = FQ_BINS>
__global__ __launch_bounds__(kKlBlock) void kl_prepare_kernel(const long long* __restrict__ hist, double* __restrict__ Pw,
                                                              double* __restrict__ tailw) {
    __shared__ double sP[B];
    __shared__ long long s_part[kKlBlock / kWave];
    const int row = blockIdx.x, tid = threadIdx.x;
    const long long* h = hist + (size_t)row * B;
    long long part = 0;
    for (int j = tid; j < B; j += kKlBlock) part += h[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, kWave);
    if ((tid & (kWave - 1)) == 0) s_part[tid / kWave] = part;
    __syncthreads();
    long long total = 0;
#pragma unroll
    for (int w = 0; w < kKlBlock / kWave; ++w) total += s_part[w];
    const double denom = (double)total + 1e-12;                  // quantizer.py:96
    for (int j = tid; j < B; j += kKlBlock) {
        const double p = (double)(float)h[j] / denom;            // astype(float32) / float64 scalar
        sP[j] = p;
        Pw[(size_t)row * B + j] = p;
    }
    __syncthreads();
    if (tid == 0) {
        double* tail = tailw + (size_t)row * B;
        double ts = pw_sum_serial(sP + kTarget, B - kTarget);      // :100
        tail[kTarget] = ts;
#pragma unroll 8
        for (int t = kTarget; t < B - 1; ++t) {                    // :108
            ts = ts - sP[t];
            tail[t + 1] = ts;
        }
    }
}

// ---- sweep: one workgroup per (threshold, row) --------------------------------------------------
template <int B = FQ_BINS>
struct SweepSmemT {
    double sP[B];
    double sE[B];            // expand_distribution, later reused for the compacted KL terms
    double s_leaf_val[kMaxLeaves];
    int s_leaf_off[kMaxLeaves];
    int s_leaf_len[kMaxLeaves];
    int s_wave_cnt[kKlBlock / kWave];
    int s_nleaf;
};
typedef SweepSmemT<FQ_BINS> SweepSmem;

// KL(t) of one row, by the whole workgroup; lane 0 of wave 0 stores it to *out.  Exact: the reference's operations in
// the reference's order.  (Every thread must call it; it ends without a barrier -- callers that reuse `sm` add one.)
template <int B = FQ_BINS>
__device__ __forceinline__ void kl_exact_candidate(SweepSmemT<B>& sm, const double* __restrict__ P, const double tail, const int t,
                                                   double* __restrict__ out) {
    constexpr int KPL = B > 2048 ? 16 : 8;            // consecutive bins per lane in the KL-term phase (256 lanes cover B)
    static_assert(KPL * kKlBlock >= B && B / kTarget <= 32, "INTERVAL_NUM beyond the kernel's lane mapping");
    double* sP = sm.sP;
    double* sE = sm.sE;
    double* s_leaf_val = sm.s_leaf_val;
    int* s_leaf_off = sm.s_leaf_off;
    int* s_leaf_len = sm.s_leaf_len;
    int* s_wave_cnt = sm.s_wave_cnt;
    int& s_nleaf = sm.s_nleaf;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;

    for (int j = tid; j < t; j += kKlBlock) {
        sP[j] = P[j];
        sE[j] = 1e-9;                                                     // :111
    }
    __syncthreads();

    // one lane per quantised bin i (quantizer.py:114-160)
    double ev = 0.0, ls = 0.0, rs = 0.0;
    int lu = 0, rl = 0;
    bool left_live = false;
    if (tid < kTarget) {
        const double npb = (double)t / (double)kTarget;                   // exact dyadic
        const double start = (double)tid * npb;
        const double end = start + npb;
        lu = (int)ceil(start);
        rl = (int)floor(end);
        const bool has_l = (double)lu > start;
        const bool has_r = (double)rl < end;
        double q = 0.0;
        if (has_l) { ls = (double)lu - start; q += ls * sP[lu - 1]; }
        if (has_r) { rs = end - (double)rl;   q += rs * sP[rl]; }
        unsigned nz = 0;
        if constexpr (B > 2048) q += pw_leaf_nz32(sP + lu, rl - lu, &nz);  // slice .sum(), length <= B / 128
        else q += pw_leaf_nz(sP + lu, rl - lu, &nz);                      // length <= 16
        double count = 1e-12;
        left_live = has_l && sP[lu - 1] != 0.0;
        const bool right_live = has_r && sP[rl] != 0.0;
        if (left_live) count += ls;
        if (right_live) count += rs;
        for (int k = __popc(nz); k > 0; --k) count = count + 1.0;         // one rounding per non-zero bin, as the reference
        ev = q / count;
        // this bin's right edge and interior first; its left edge is shared with bin i-1's right
        // edge, which the reference adds earlier (loop order), so left edges go in a second phase.
        if (right_live) sE[rl] = 1e-9 + ev * rs;
        const double inner = 1e-9 + ev;
        for (unsigned mm = nz; mm; mm &= mm - 1) sE[lu + __ffs(mm) - 1] = inner;
    }
    __syncthreads();
    if (left_live) sE[lu - 1] += ev * ls;
    __syncthreads();

    // KL terms (quantizer.py:169-174) for 8 (16) consecutive bins per lane, compacted over a != 0
    double term[KPL];
    unsigned nzmask = 0;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int j = tid * KPL + k;
        term[k] = 0.0;
        if (j < t) {
            double a = sP[j];
            if (j == t - 1) a += tail;                                    // :105
            if (a != 0.0) {
                const double arg = a / (sE[j] + 1e-12) + 1e-12;
                term[k] = a * fq_log(arg);
                nzmask |= 1u << k;
            }
        }
    }
    const int cnt = __popc(nzmask);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int v = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += v;
    }
    if (lane == kWave - 1) s_wave_cnt[wave] = incl;
    __syncthreads();                                  // also: every read of sE above is done
    int base = 0, m = 0;
#pragma unroll
    for (int w = 0; w < kKlBlock / kWave; ++w) {
        const int c = s_wave_cnt[w];
        if (w < wave) base += c;
        m += c;
    }
    double* sT = sE;
    int pos = base + incl - cnt;
#pragma unroll
    for (int k = 0; k < KPL; ++k)
        if (nzmask & (1u << k)) sT[pos++] = term[k];
    // Leaves of NumPy's pairwise recursion over the m compacted terms, without walking the tree serially (one
    // lane doing that took half of the kernel): the tree is at most 6 levels deep for m <= 2048, so lane c of
    // wave 0 follows the 6-bit path c (bit = right child) from the root; a leaf reached before the path is used
    // up belongs to the path whose remaining bits are zero.  Valid paths in increasing c are the leaves in
    // order; their ranks give the compacted leaf list.
    unsigned long long leaf_mask = 0;
    int leaf_rank = 0;
    bool leaf_valid = false;
    if (wave == 0) {
        int off = 0, n = m;
        leaf_valid = m > 0;
        for (int level = 5; level >= 0; --level) {
            if (n <= 128) {
                if (lane & ((2 << level) - 1)) leaf_valid = false;       // bits level..0 must be zero
                break;
            }
            const int n2 = pw_half(n);
            if ((lane >> level) & 1) { off += n2; n -= n2; } else { n = n2; }
        }
        if (n > 128) leaf_valid = false;
        leaf_mask = __ballot(leaf_valid);
        leaf_rank = __popcll(leaf_mask & ((1ull << lane) - 1ull));
        if (leaf_valid) { s_leaf_off[leaf_rank] = off; s_leaf_len[leaf_rank] = n; }
        if (lane == 0) s_nleaf = __popcll(leaf_mask);
    }
    __syncthreads();

    // leaves in parallel: 8 lanes = NumPy's 8 strided accumulators
    // (32 leaves per pass: a sum of up to 2048 terms has at most 17 leaves, of 4096 terms 33)
#pragma unroll
    for (int L0 = 0; L0 < (B > 2048 ? 64 : 32); L0 += 32) {
        const int L = L0 + (tid >> 3), k = tid & 7;
        const bool live = L < s_nleaf;
        const int off = live ? s_leaf_off[L] : 0;
        const int len = live ? s_leaf_len[L] : 0;
        double r = 0.0;
        if (len >= 8) {
            r = sT[off + k];
            const int lim = len - (len & 7);
            for (int i = 8; i < lim; i += 8) r += sT[off + i + k];
        }
        const int g = lane & ~7;
        const double r0 = __shfl(r, g + 0, kWave), r1 = __shfl(r, g + 1, kWave);
        const double r2 = __shfl(r, g + 2, kWave), r3 = __shfl(r, g + 3, kWave);
        const double r4 = __shfl(r, g + 4, kWave), r5 = __shfl(r, g + 5, kWave);
        const double r6 = __shfl(r, g + 6, kWave), r7 = __shfl(r, g + 7, kWave);
        if (live && k == 0) {
            double res;
            if (len < 8) {
                res = 0.0;
                for (int i = 0; i < len; ++i) res += sT[off + i];
            } else {
                res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
                for (int i = len - (len & 7); i < len; ++i) res += sT[off + i];
            }
            s_leaf_val[L] = res;
        }
    }
    __syncthreads();
    // combine the leaves along the same tree: the subtree whose leftmost leaf sits on path c keeps its running value
    // in lane c; at stride s the node at c (c % 2s == 0) adds its right child, the subtree that starts at c + s
    if (wave == 0) {
        double v = leaf_valid ? s_leaf_val[leaf_rank] : 0.0;
#pragma unroll
        for (int sft = 0; sft < 6; ++sft) {
            const int sd = 1 << sft;
            const double other = __shfl_down(v, sd, kWave);
            const bool right = lane + sd < kWave && ((leaf_mask >> (lane + sd)) & 1ull);
            if ((lane & (2 * sd - 1)) == 0 && right) v = v + other;
        }
        if (lane == 0) *out = v;                                         // m == 0: np.sum of nothing = 0.0
    }
}

// ---- sweep: one workgroup per (threshold, row) --------------------------------------------------
__global__ __launch_bounds__(kKlBlock) void kl_sweep_kernel(const double* __restrict__ Pw, const double* __restrict__ tailw,
                                                            double* __restrict__ klw) {
    __shared__ SweepSmem sm;
    const int t = kTarget + blockIdx.x;       // threshold
    const int row = blockIdx.y;
    kl_exact_candidate<FQ_BINS>(sm, Pw + (size_t)row * FQ_BINS, tailw[(size_t)row * FQ_BINS + t], t, klw + (size_t)row * kCand + blockIdx.x);
}

// the same sweep for another INTERVAL_NUM: B - 128 candidates per row; the two B-sized LDS arrays are dynamic (64 KB at 4096)
template <int B>
__global__ __launch_bounds__(kKlBlock) void kl_sweep_n_kernel(const double* __restrict__ Pw, const double* __restrict__ tailw,
                                                              double* __restrict__ klw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char kl_dyn_smem[];
    SweepSmemT<B>& sm = *reinterpret_cast<SweepSmemT<B>*>(kl_dyn_smem);
    const int t = kTarget + blockIdx.x;
    const int row = blockIdx.y;
    kl_exact_candidate<B>(sm, Pw + (size_t)row * B, tailw[(size_t)row * B + t], t, klw + (size_t)row * (B - kTarget) + blockIdx.x);
}

// ---- screened path ------------------------------------------------------------------------------
// Closed form of the same sum.  With a_j = P[j] (a_{t-1} = P[t-1] + tail_t), E_j the expanded value and E'_j = E_j + 1e-12,
//     KL(t) = sum_{a_j != 0} a_j * log(a_j / E'_j + 1e-12).
// (i)  log(r + 1e-12) = log r + x - x^2/2 + ..., x = 1e-12 / r = 1e-12 E'_j / a_j: the first-order part of the sum is
//      1e-12 * sum_j E'_j (kept, "corr" below); what is dropped is below a_j x^2 / 2 = 1e-24 E'_j^2 / (2 a_j) per term,
//      < 1e-15 in total for any histogram of fewer than 1e9 samples per bin.
// (ii) sum a_j log(a_j / E'_j) = sum a_j log a_j - sum a_j log E'_j; E'_j is the same number for every non-empty bin j in
//      the interior of quantised bin i, so the second sum collapses to one logarithm per quantised bin (times the
//      interior mass, a difference of two prefix sums), one per fractional bin edge and one for the folded last bin:
//      <= 257 logarithms per candidate instead of ~t, and no dependence on which bins are empty beyond a prefix count.
// (iii) Evaluated in float64 (ocml log, < 1 ulp) with the prefix sums held as double-double (a difference of two plain
//      prefixes would carry 2^-53 of the PREFIX, which a logarithm of a tiny expanded value multiplies by up to 21:
//      3e-12 on the golden "bimodal"); the rest is <= 260 products of magnitude <= 8 summed along a tree: < 1e-13.
// Total |S(t) - KL(t)| < FQ_KL_SCREEN_BOUND = 2e-13 (the numpy restatement in tests/test_kl_screen_cpu.py stays below 1e-13
// against the exact oracle on every golden and fuzz histogram).  Candidates are kept when S(t) <= min S +
// FQ_KL_SCREEN_MARGIN = 1e-10 (+ 1e-12 |min S|): 500 times the bound (include/fq.h holds both numbers).  A NaN S(t) (the reference's incremental tail can go slightly negative; log of a negative number)
// is always kept, so the exact pass reproduces the NaN, which never wins.
constexpr double kScreenMargin = FQ_KL_SCREEN_MARGIN;
static_assert(2.0 * FQ_KL_SCREEN_BOUND < FQ_KL_SCREEN_MARGIN, "the true minimum must survive the screen");
constexpr int kListPerRow = 64;               // rows with more survivors than this are swept exhaustively
constexpr int kNzStride = FQ_BINS + 4;

// exclusive prefixes of one row as double-double (hi, lo): 8 consecutive bins per thread, block scan of the 256 partial
// sums, every addition error-free (two_sum) so that prefix DIFFERENCES are accurate to 2^-53 of the difference
__device__ __forceinline__ fq_dd dd_acc(fq_dd s, double v) {          // s + v, renormalised
    const fq_dd t = fq_two_sum(s.hi, v);
    return fq_fast_two_sum(t.hi, t.lo + s.lo);
}

__global__ __launch_bounds__(kKlBlock) void kl_prefix_kernel(const double* __restrict__ Pw, double* __restrict__ SPw,
                                                             double* __restrict__ ALw, unsigned short* __restrict__ NZw,
                                                             int* __restrict__ counters) {
    __shared__ double s_ph[kKlBlock], s_pl[kKlBlock], s_ah[kKlBlock], s_al[kKlBlock];
    __shared__ int s_n[kKlBlock];
    const int row = blockIdx.x, tid = threadIdx.x;
    if (row == 0 && tid < 2) counters[tid] = 0;
    const double* P = Pw + (size_t)row * FQ_BINS + tid * 8;
    double p[8], a[8];
    fq_dd sp = {0.0, 0.0}, sa = {0.0, 0.0};
    int sn = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        p[k] = P[k];
        a[k] = p[k] != 0.0 ? p[k] * log(p[k]) : 0.0;
        sp = dd_acc(sp, p[k]); sa = dd_acc(sa, a[k]); sn += p[k] != 0.0 ? 1 : 0;
    }
    s_ph[tid] = sp.hi; s_pl[tid] = sp.lo; s_ah[tid] = sa.hi; s_al[tid] = sa.lo; s_n[tid] = sn;
    __syncthreads();
    for (int d = 1; d < kKlBlock; d <<= 1) {              // Hillis-Steele inclusive scan in double-double
        fq_dd vp = {0.0, 0.0}, va = {0.0, 0.0}; int vn = 0;
        if (tid >= d) { vp.hi = s_ph[tid - d]; vp.lo = s_pl[tid - d]; va.hi = s_ah[tid - d]; va.lo = s_al[tid - d]; vn = s_n[tid - d]; }
        __syncthreads();
        if (tid >= d) {
            fq_dd cp = {s_ph[tid], s_pl[tid]}, ca = {s_ah[tid], s_al[tid]};
            cp = fq_dd_add(cp, vp); ca = fq_dd_add(ca, va);
            s_ph[tid] = cp.hi; s_pl[tid] = cp.lo; s_ah[tid] = ca.hi; s_al[tid] = ca.lo; s_n[tid] += vn;
        }
        __syncthreads();
    }
    // exclusive base of this thread's 8 bins = inclusive value of the thread before
    fq_dd bp = {0.0, 0.0}, ba = {0.0, 0.0};
    int bn = 0;
    if (tid > 0) { bp.hi = s_ph[tid - 1]; bp.lo = s_pl[tid - 1]; ba.hi = s_ah[tid - 1]; ba.lo = s_al[tid - 1]; bn = s_n[tid - 1]; }
    double* SP = SPw + (size_t)row * 2 * (FQ_BINS + 1) + tid * 16;     // interleaved (hi, lo)
    double* AL = ALw + (size_t)row * 2 * (FQ_BINS + 1) + tid * 16;
    unsigned short* NZ = NZw + (size_t)row * kNzStride + tid * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        SP[2 * k] = bp.hi; SP[2 * k + 1] = bp.lo; AL[2 * k] = ba.hi; AL[2 * k + 1] = ba.lo; NZ[k] = (unsigned short)bn;
        bp = dd_acc(bp, p[k]); ba = dd_acc(ba, a[k]); bn += p[k] != 0.0 ? 1 : 0;
    }
    if (tid == kKlBlock - 1) { SP[16] = bp.hi; SP[17] = bp.lo; AL[16] = ba.hi; AL[17] = ba.lo; NZ[8] = (unsigned short)bn; }
}

// log for the screen: the table-driven reduction of fq_log's fast path (include/fq_log.h) without its double-double
// bookkeeping and without the rounding test -- x = 2^k z, z in [0.6875, 1.375), r = z * invc - 1 (|r| < 2^-7),
// log x = k ln2 + log(1/invc) + log1p(r) with a degree-7 polynomial: relative error < 2^-58, ~25 flops where ocml's
// (< 1 ulp) log costs ~120 instructions (the screen was 640 vector instructions per candidate and wave, two
// thirds of them its two logarithms: 8.4 -> 5.4 ms per 4 096 rows).  Anything but a positive normal number goes to ocml's log
// (a negative folded bin must give NaN).
__device__ __forceinline__ double screen_log(double x) {
    unsigned long long ix;
    memcpy(&ix, &x, 8);
    if (ix - 0x0010000000000000ULL >= 0x7ff0000000000000ULL - 0x0010000000000000ULL) return log(x);
    const unsigned long long tmp = ix - 0x3fe6000000000000ULL;
    const int i = (int)((tmp >> 45) & 127);
    const long long k = (long long)tmp >> 52;
    const unsigned long long iz = ix - (tmp & 0xfff0000000000000ULL);
    double z;
    memcpy(&z, &iz, 8);
    const double invc = fq_log_table[i].invc, lch = fq_log_table[i].logc_hi, lcl = fq_log_table[i].logc_lo;
    const double r = __builtin_fma(z, invc, -1.0);
    double c = 1.0 / 7.0;
    c = __builtin_fma(c, r, -1.0 / 6.0);
    c = __builtin_fma(c, r, 1.0 / 5.0);
    c = __builtin_fma(c, r, -1.0 / 4.0);
    c = __builtin_fma(c, r, 1.0 / 3.0);
    c = __builtin_fma(c, r, -0.5);
    const double dk = (double)k;
    const double hi = __builtin_fma(dk, 0x1.62e42fefa38p-1, lch);               // k * LN2_HI is exact
    const double lo = __builtin_fma(dk, 0x1.ef35793c7673p-45, lcl);
    return hi + (r + __builtin_fma(r * r, c, lo));
}

// S(t) for all 1920 candidates of one row: the row's P, prefix sums and prefix counts staged in LDS once, one wave per
// candidate (no workgroup barrier inside the loop), two quantised bins per lane.  Every lane's terms are those of the first
// form of this kernel bit for bit (same expressions, same order); the 64 lanes' sums now meet in a different order (and the
// 1e-12 correction in fp32), which moves S(t) by parts in 1e16.  What changed in round 4 is what a candidate costs -- 520
// vector instructions (205 of them fp64) with 39 waits became ~320:
//   * bin edges in integers: start = i t / 128, so ceil(start) = (i t + 127) >> 7, floor(end) = ((i + 1) t) >> 7 and the two
//     fractional weights are (128 - (i t & 127)) / 128 and (((i + 1) t) & 127) / 128 -- exact, no fp64 ceil / floor / compare;
//   * the prefix sums as (hi, lo) pairs: one 16-byte LDS read per index; one mass per quantised bin (the interior of bin 127
//     stops at t - 1: its pair is read by every lane from the uniform address and selected);
//   * the four logarithms of a lane (two interiors, two fractional edges) are computed unconditionally, side by side, with the
//     table in LDS, and selected afterwards -- no divergent branch per term; an argument the fast path does not take (never for
//     these strictly positive sums) sends the wave through the library logarithm for exactly those lanes;
//   * the neighbour's expanded value and the wave's sums by DPP (wave_rol:1; row_shr / row_bcast) instead of 30 ds_bpermute;
//   * the closing step of a candidate (the folded last bin: two more logarithms, three global loads) ran in ONE lane per
//     candidate; now lane j keeps the reduced sums of the j-th candidate of a batch of 64 and the 64 closing steps run as one.
constexpr int kScreenBlock = 512;                 // 8 waves, 240 candidates each; 56 KB of LDS: two workgroups per CU
struct alignas(16) ScreenLogRow { double invc, logc_hi, logc_lo, pad; };
struct alignas(16) DdPair { double hi, lo; };

__device__ __forceinline__ double lane_plus_one(double v) {           // v of lane (lane + 1) & 63
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x134, 0xf, 0xf, false);  // wave_rol:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x134, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int kCtrl, int kRowMask, int kBankMask>
__device__ __forceinline__ double dpp_or_zero(double v) {             // the DPP source lane's v; 0 where there is none / masked
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, kRowMask, kBankMask, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, kRowMask, kBankMask, true);
    return __hiloint2double(hi, lo);
}
template <int kCtrl, int kRowMask, int kBankMask>
__device__ __forceinline__ float dpp_or_zero_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, kRowMask, kBankMask, true));
}
__device__ __forceinline__ double wave_sum_to_last(double v) {        // lane 63: the sum over the wave
    double s = v + dpp_or_zero<0x111, 0xf, 0xf>(v);                    // row_shr:1
    s += dpp_or_zero<0x112, 0xf, 0xf>(v);                             // row_shr:2
    s += dpp_or_zero<0x113, 0xf, 0xf>(v);                             // row_shr:3   lane i: v[i-3 .. i] of its row
    s += dpp_or_zero<0x114, 0xf, 0xe>(s);                             // row_shr:4, lanes 4..15 of a row
    s += dpp_or_zero<0x118, 0xf, 0xc>(s);                             // row_shr:8, lanes 8..15: lane 15 = the row
    s += dpp_or_zero<0x142, 0xa, 0xf>(s);                             // row_bcast:15 into rows 1 and 3
    s += dpp_or_zero<0x143, 0xc, 0xf>(s);                             // row_bcast:31 into rows 2 and 3
    return s;
}
__device__ __forceinline__ float wave_sum_to_last_f(float v) {
    float s = v + dpp_or_zero_f<0x111, 0xf, 0xf>(v);
    s += dpp_or_zero_f<0x112, 0xf, 0xf>(v);
    s += dpp_or_zero_f<0x113, 0xf, 0xf>(v);
    s += dpp_or_zero_f<0x114, 0xf, 0xe>(s);
    s += dpp_or_zero_f<0x118, 0xf, 0xc>(s);
    s += dpp_or_zero_f<0x142, 0xa, 0xf>(s);
    s += dpp_or_zero_f<0x143, 0xc, 0xf>(s);
    return s;
}
__device__ __forceinline__ double lane63(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), kWave - 1), __builtin_amdgcn_readlane(__double2loint(v), kWave - 1));
}
__device__ __forceinline__ float lane63f(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), kWave - 1)); }
// screen_log's test for "not a positive normal number" and its fast path, on the high word (every constant involved has a
// zero low word)
__device__ __forceinline__ bool screen_log_odd(double x) {
    return (unsigned)__double2hiint(x) - 0x00100000u >= 0x7fe00000u;
}
__device__ __forceinline__ double screen_log_core(double x, const ScreenLogRow* __restrict__ tab) {
    const unsigned hx = (unsigned)__double2hiint(x);
    const unsigned tmp = hx - 0x3fe60000u;
    const int i = (int)((tmp >> 13) & 127u);
    const int k = (int)tmp >> 20;
    const double z = __hiloint2double((int)(hx - (tmp & 0xfff00000u)), __double2loint(x));
    const double invc = tab[i].invc, lch = tab[i].logc_hi, lcl = tab[i].logc_lo;
    const double r = __builtin_fma(z, invc, -1.0);
    double c = 1.0 / 7.0;
    c = __builtin_fma(c, r, -1.0 / 6.0);
    c = __builtin_fma(c, r, 1.0 / 5.0);
    c = __builtin_fma(c, r, -1.0 / 4.0);
    c = __builtin_fma(c, r, 1.0 / 3.0);
    c = __builtin_fma(c, r, -0.5);
    const double dk = (double)k;
    const double hi = __builtin_fma(dk, 0x1.62e42fefa38p-1, lch);
    const double lo = __builtin_fma(dk, 0x1.ef35793c7673p-45, lcl);
    return hi + (r + __builtin_fma(r * r, c, lo));
}

__global__ __launch_bounds__(kScreenBlock) void kl_screen_kernel(const double* __restrict__ Pw, const double* __restrict__ tailw,
                                                                 const double* __restrict__ SPw, const double* __restrict__ ALw,
                                                                 const unsigned short* __restrict__ NZw, double* __restrict__ klw) {
    __shared__ double sP[FQ_BINS];
    __shared__ DdPair sSP[FQ_BINS + 1];
    __shared__ unsigned short sNZ[kNzStride];
    __shared__ ScreenLogRow sLog[128];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    for (int j = tid; j < FQ_BINS; j += kScreenBlock) sP[j] = Pw[(size_t)row * FQ_BINS + j];
    for (int j = tid; j <= FQ_BINS; j += kScreenBlock) {
        sSP[j] = reinterpret_cast<const DdPair*>(SPw)[(size_t)row * (FQ_BINS + 1) + j];
        sNZ[j] = NZw[(size_t)row * kNzStride + j];
    }
    if (tid < 128) { sLog[tid].invc = fq_log_table[tid].invc; sLog[tid].logc_hi = fq_log_table[tid].logc_hi; sLog[tid].logc_lo = fq_log_table[tid].logc_lo; }
    __syncthreads();
    const double* tail = tailw + (size_t)row * FQ_BINS;
    const double* AL = ALw + (size_t)row * 2 * (FQ_BINS + 1);

    // what lane j keeps of the j-th candidate of the running batch of 64
    double cap_part = 0.0, cap_corr = 0.0, cap_ev = 0.0;
    int cap_c = -1;
    auto close_batch = [&]() {
        if (cap_c >= 0) {
            const int t = kTarget + cap_c;
            const double plast = sP[t - 1];
            const double a_last = plast + tail[t];
            double s = (AL[2 * (t - 1)] - cap_part) + AL[2 * (t - 1) + 1];
            double corr = cap_corr;
            if (a_last != 0.0) {                                      // negative (rounding of the tail chain): NaN, kept
                const double e_last = (plast != 0.0 ? 1e-9 + cap_ev : 1e-9) + 1e-12;
                s += a_last * screen_log(a_last) - a_last * screen_log(e_last);
                corr += e_last;
            }
            klw[(size_t)row * kCand + cap_c] = s + 1e-12 * corr;
        }
        cap_c = -1;
    };

    constexpr int kWaves = kScreenBlock / kWave;
    for (int it = 0, c = wave; c < kCand; ++it, c += kWaves) {
        const int t = kTarget + c;
        const DdPair sp_t1 = sSP[t - 1];                              // uniform: the end of bin 127's interior
        const int nz_t1 = (int)sNZ[t - 1];
        double ev[2], mint[2], rs_[2], pr_[2];
        int nint[2], rl_[2];
        bool has_r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = lane + h * kWave;
            const int a0 = i * t, a1 = a0 + t;
            const int lu = (a0 + (kTarget - 1)) >> 7, rl = a1 >> 7;
            const int fl = a0 & (kTarget - 1), fr = a1 & (kTarget - 1);
            const double ls = (double)(fl ? kTarget - fl : 0) * (1.0 / kTarget);      // ceil(start) - start, exact
            const double rs = (double)fr * (1.0 / kTarget);                           // end - floor(end), exact
            const double pl_raw = sP[lu > 0 ? lu - 1 : 0], pr_raw = sP[rl < FQ_BINS ? rl : FQ_BINS - 1];
            const double pl = fl ? pl_raw : 0.0, pr = fr ? pr_raw : 0.0;
            const DdPair sa = sSP[lu], sb = sSP[rl];
            const int nza = (int)sNZ[lu], nzb = (int)sNZ[rl];
            const double m = (sb.hi - sa.hi) + (sb.lo - sa.lo);                       // sum of P[lu .. rl)
            const double q = ls * pl + rs * pr + m;
            double count = 1e-12 + (double)(nzb - nza);
            if (pl != 0.0) count += ls;
            if (pr != 0.0) count += rs;
            ev[h] = q / count;
            // interior of this quantised bin; the folded last bin t-1 (always interior of bin 127) is handled apart
            const bool last = h == 1 && lane == kWave - 1;
            const double eh = last ? sp_t1.hi : sb.hi, el = last ? sp_t1.lo : sb.lo;
            mint[h] = last ? (eh - sa.hi) + (el - sa.lo) : m;
            nint[h] = (last ? nz_t1 : nzb) - nza;
            rl_[h] = rl; rs_[h] = rs; pr_[h] = pr_raw; has_r[h] = fr != 0 && pr_raw != 0.0;
        }
        // fractional right edges: bin rl belongs to quantised bins i (weight rs) and i + 1 (weight 1 - rs).
        // bin i = lane      -> neighbour i + 1 is lane + 1's first bin, or (lane 63) bin 64 = lane 0's SECOND bin
        // bin i = lane + 64 -> neighbour is lane + 1's second bin; bin 127 has no right edge (end == t is an integer)
        const double nxt0 = lane_plus_one(ev[0]);
        const double nxt1 = lane_plus_one(ev[1]);                     // in lane 63: lane 0's second bin = bin 64
        const double evn0 = (lane == kWave - 1) ? nxt1 : nxt0;
        const bool use[4] = {nint[0] > 0, nint[1] > 0, has_r[0], has_r[1] && lane != kWave - 1};
        double e[4];
        e[0] = (1e-9 + ev[0]) + 1e-12;
        e[1] = (1e-9 + ev[1]) + 1e-12;
        e[2] = ((1e-9 + ev[0] * rs_[0]) + evn0 * (1.0 - rs_[0])) + 1e-12;
        e[3] = ((1e-9 + ev[1] * rs_[1]) + nxt1 * (1.0 - rs_[1])) + 1e-12;
        double lg[4];
        bool odd = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lg[k] = screen_log_core(e[k], sLog);
            odd |= use[k] && screen_log_odd(e[k]);
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0, 0)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (use[k] && screen_log_odd(e[k])) lg[k] = log(e[k]);
        }
        double part = 0.0, corr = 0.0;
        part += use[0] ? mint[0] * lg[0] : 0.0;  corr += use[0] ? (double)nint[0] * e[0] : 0.0;
        part += use[1] ? mint[1] * lg[1] : 0.0;  corr += use[1] ? (double)nint[1] * e[1] : 0.0;
        part += use[2] ? pr_[0] * lg[2] : 0.0;   corr += use[2] ? e[2] : 0.0;
        part += use[3] ? pr_[1] * lg[3] : 0.0;   corr += use[3] ? e[3] : 0.0;
        // the wave's sums, in lane 63 (row_shr 1, 2, 3, 4, 8, row_bcast 15, 31: no LDS, no wait); corr only ever enters as
        // 1e-12 * corr, so it is added up in fp32 (6e-8 of 1e-12 of a sum of order one)
        const double part_sum = lane63(wave_sum_to_last(part));
        const double corr_sum = (double)lane63f(wave_sum_to_last_f((float)corr));
        const double ev_last = lane63(ev[1]);
        if ((it & (kWave - 1)) == lane) { cap_part = part_sum; cap_corr = corr_sum; cap_ev = ev_last; cap_c = c; }
        if ((it & (kWave - 1)) == kWave - 1) close_batch();
    }
    close_batch();
}

// survivors of one row -> work list (or the row -> exhaustive list when there are too many)
__global__ __launch_bounds__(kKlBlock) void kl_select_kernel(const double* __restrict__ klw, int* __restrict__ list,
                                                             int* __restrict__ full_rows, int* __restrict__ counters,
                                                             const int list_capacity) {
    __shared__ double s_v[kKlBlock];
    __shared__ int s_cnt;
    const int row = blockIdx.x, tid = threadIdx.x;
    const double* kl = klw + (size_t)row * kCand;
    double mn = __builtin_inf();
    for (int i = tid; i < kCand; i += kKlBlock) { const double v = kl[i]; if (v < mn) mn = v; }     // NaN never lowers it
    s_v[tid] = mn;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int s2 = kKlBlock / 2; s2 >= 1; s2 >>= 1) {
        if (tid < s2 && s_v[tid + s2] < s_v[tid]) s_v[tid] = s_v[tid + s2];
        __syncthreads();
    }
    mn = s_v[0];
    const double limit = mn + (kScreenMargin + 1e-12 * fabs(mn));       // inf when no S is finite: everything survives
    int mine = 0;
    for (int i = tid; i < kCand; i += kKlBlock) mine += !(kl[i] > limit) ? 1 : 0;
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    const int total = s_cnt;
    if (total > kListPerRow) {
        if (tid == 0) full_rows[atomicAdd(&counters[1], 1)] = row;
        return;
    }
    for (int i = tid; i < kCand; i += kKlBlock)
        if (!(kl[i] > limit)) {
            const int slot = atomicAdd(&counters[0], 1);
            if (slot < list_capacity) list[slot] = (row << 11) | i;
        }
}

// exact KL of the listed candidates and of every candidate of the listed rows; a fixed grid strides over the work,
// whose size lives on the device (no host round trip between the screen and the exact pass)
__global__ __launch_bounds__(kKlBlock) void kl_exact_list_kernel(const double* __restrict__ Pw, const double* __restrict__ tailw,
                                                                 double* __restrict__ klw, const int* __restrict__ list,
                                                                 const int* __restrict__ full_rows,
                                                                 const int* __restrict__ counters, const int list_capacity) {
    __shared__ SweepSmem sm;
    const int n_list = counters[0] < list_capacity ? counters[0] : list_capacity;
    const long long total = (long long)n_list + (long long)counters[1] * kCand;
    for (long long w = blockIdx.x; w < total; w += gridDim.x) {
        int row, c;
        if (w < n_list) {
            const int e = list[w];
            row = e >> 11; c = e & 2047;
        } else {
            const long long r = w - n_list;
            row = full_rows[r / kCand]; c = (int)(r % kCand);
        }
        const int t = kTarget + c;
        kl_exact_candidate<FQ_BINS>(sm, Pw + (size_t)row * FQ_BINS, tailw[(size_t)row * FQ_BINS + t], t, klw + (size_t)row * kCand + c);
        __syncthreads();                                               // sm is reused by the next item
    }
}

// ---- argmin: first strict minimum below 66666 (quantizer.py:99,:163-165), default 2047 (:101) ----
// Also reports how decisive the choice was: best = KL(t*), runner_up = the smallest KL of any OTHER candidate (+inf when
// there is none): a runner-up within an ulp-sized distance of the best is a near-tie that a differently rounded
// logarithm (np.log vs fq_log, DESIGN.md "The logarithm") could have decided the other way.
template <int B = FQ_BINS>
__global__ __launch_bounds__(kKlBlock) void kl_argmin_kernel(const double* __restrict__ klw, int* __restrict__ thr_out,
                                                             double* __restrict__ best_out, double* __restrict__ runner_out) {
    __shared__ double s_v[kKlBlock], s_w[kKlBlock];
    __shared__ int s_t[kKlBlock];
    const int row = blockIdx.x, tid = threadIdx.x;
    constexpr int kCandB = B - kTarget;
    const double* kl = klw + (size_t)row * kCandB;
    double bv = 66666.0, second = __builtin_inf();
    int bt = 0x7fffffff;
    for (int i = tid; i < kCandB; i += kKlBlock) {
        const double v = kl[i];
        if (v < bv) { if (bt != 0x7fffffff) second = bv; bv = v; bt = kTarget + i; }   // NaN never wins
        else if (v < second) second = v;
    }
    s_v[tid] = bv; s_t[tid] = bt; s_w[tid] = second;
    __syncthreads();
    for (int s = kKlBlock / 2; s >= 1; s >>= 1) {
        if (tid < s) {
            const double v2 = s_v[tid + s], w2 = s_w[tid + s];
            const int t2 = s_t[tid + s];
            const bool take = v2 < s_v[tid] || (v2 == s_v[tid] && t2 < s_t[tid]);
            // the loser's best is a candidate for runner-up (only if it stands for a real candidate)
            double lose = take ? s_v[tid] : v2;
            const int lose_t = take ? s_t[tid] : t2;
            if (lose_t == 0x7fffffff) lose = __builtin_inf();
            double w = s_w[tid] < w2 ? s_w[tid] : w2;
            if (lose < w) w = lose;
            if (take) { s_v[tid] = v2; s_t[tid] = t2; }
            s_w[tid] = w;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const bool none = s_t[0] == 0x7fffffff;
        thr_out[row] = none ? (B - 1) : s_t[0];
        if (best_out) best_out[row] = none ? __builtin_inf() : s_v[0];
        if (runner_out) runner_out[row] = s_w[0];
    }
}

constexpr int kRowChunk = 4096;              // rows per pass over the workspace (keeps it at ~340 MB for any row count)
constexpr int kScreenMinRows = 256;          // FQ_KL_AUTO: below this the exhaustive sweep is latency bound anyway

constexpr size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }
constexpr size_t kWsPerRow = (size_t)(FQ_BINS + FQ_BINS + kCand) * sizeof(double)             // P, tail, KL curve
                             + 4 * (size_t)(FQ_BINS + 1) * sizeof(double)                      // SP, AL (double-double)
                             + align8((size_t)kNzStride * sizeof(unsigned short))              // NZ
                             + (size_t)(kListPerRow + 1) * sizeof(int) + 8;                    // work list, full-row list

static bool kl_env_exhaustive() {
    static const bool v = [] { const char* e = getenv("FQ_KL_EXHAUSTIVE"); return e && e[0] && e[0] != '0'; }();
    return v;
}

}  // namespace fq

extern "C" size_t fq_kl_workspace_bytes(int rows) {
    if (rows <= 0) return 0;
    const size_t r = rows < fq::kRowChunk ? (size_t)rows : (size_t)fq::kRowChunk;
    return r * fq::kWsPerRow + 64;
}

extern "C" int fq_kl_threshold_ex(const int64_t* hist, int rows, int32_t* thr_out, double* best_kl_out,
                                  double* runner_up_kl_out, double* kl_curve_out, int mode,
                                  void* workspace, size_t workspace_bytes, fq_stream_t stream) {
    using namespace fq;
    if (rows < 0 || mode < FQ_KL_AUTO || mode > FQ_KL_SCREENED) return FQ_ERR_INVALID_ARG;
    if (rows == 0) return FQ_OK;
    if (!hist || !thr_out || !workspace) return FQ_ERR_INVALID_ARG;
    if (workspace_bytes < fq_kl_workspace_bytes(rows)) return FQ_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 7u) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    const bool screened = mode == FQ_KL_SCREENED ||
                          (mode == FQ_KL_AUTO && rows >= kScreenMinRows && !kl_curve_out && !kl_env_exhaustive());
    const size_t ch = rows < kRowChunk ? (size_t)rows : (size_t)kRowChunk;
    char* w = reinterpret_cast<char*>(workspace);
    double* Pw = reinterpret_cast<double*>(w);            w += ch * FQ_BINS * sizeof(double);
    double* tailw = reinterpret_cast<double*>(w);         w += ch * FQ_BINS * sizeof(double);
    double* kl_ws = reinterpret_cast<double*>(w);         w += ch * kCand * sizeof(double);
    double* SPw = reinterpret_cast<double*>(w);           w += ch * 2 * (FQ_BINS + 1) * sizeof(double);
    double* ALw = reinterpret_cast<double*>(w);           w += ch * 2 * (FQ_BINS + 1) * sizeof(double);
    unsigned short* NZw = reinterpret_cast<unsigned short*>(w);  w += align8(ch * kNzStride * sizeof(unsigned short));
    int* list = reinterpret_cast<int*>(w);                w += ch * kListPerRow * sizeof(int);
    int* full_rows = reinterpret_cast<int*>(w);           w += align8(ch * sizeof(int));
    int* counters = reinterpret_cast<int*>(w);
    for (int r0 = 0; r0 < rows; r0 += kRowChunk) {
        const int nr = rows - r0 < kRowChunk ? rows - r0 : kRowChunk;
        double* klw = kl_curve_out ? kl_curve_out + (size_t)r0 * kCand : kl_ws;
        hipLaunchKernelGGL(kl_prepare_kernel<FQ_BINS>, dim3(nr), dim3(kKlBlock), 0, st,
                           reinterpret_cast<const long long*>(hist) + (size_t)r0 * FQ_BINS, Pw, tailw);
        FQ_LAUNCH_CHECK();
        if (!screened) {
            hipLaunchKernelGGL(kl_sweep_kernel, dim3(kCand, nr), dim3(kKlBlock), 0, st, Pw, tailw, klw);
            FQ_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(kl_prefix_kernel, dim3(nr), dim3(kKlBlock), 0, st, Pw, SPw, ALw, NZw, counters);
            FQ_LAUNCH_CHECK();
            hipLaunchKernelGGL(kl_screen_kernel, dim3(nr), dim3(kScreenBlock), 0, st, Pw, tailw, SPw, ALw, NZw, klw);
            FQ_LAUNCH_CHECK();
            hipLaunchKernelGGL(kl_select_kernel, dim3(nr), dim3(kKlBlock), 0, st, klw, list, full_rows, counters,
                               nr * kListPerRow);
            FQ_LAUNCH_CHECK();
            const int wgs = nr * 8 < kCUs * 8 ? nr * 8 : kCUs * 8;
            hipLaunchKernelGGL(kl_exact_list_kernel, dim3(wgs), dim3(kKlBlock), 0, st, Pw, tailw, klw, list, full_rows,
                               counters, nr * kListPerRow);
            FQ_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(kl_argmin_kernel<FQ_BINS>, dim3(nr), dim3(kKlBlock), 0, st, klw, thr_out + r0,
                           best_kl_out ? best_kl_out + r0 : nullptr, runner_up_kl_out ? runner_up_kl_out + r0 : nullptr);
        FQ_LAUNCH_CHECK();
    }
    return FQ_OK;
}

namespace fq {
template <int B>
static int kl_threshold_n(const int64_t* hist, int rows, int32_t* thr_out, double* kl_curve_out, void* workspace, size_t workspace_bytes,
                          hipStream_t st) {
    constexpr int kCandB = B - kTarget;
    constexpr size_t per_row = (size_t)(2 * B + kCandB) * sizeof(double);
    const size_t ch = rows < kRowChunk ? (size_t)rows : (size_t)kRowChunk;
    if (workspace_bytes < ch * per_row) return FQ_ERR_WORKSPACE;
    char* w = reinterpret_cast<char*>(workspace);
    double* Pw = reinterpret_cast<double*>(w);            w += ch * B * sizeof(double);
    double* tailw = reinterpret_cast<double*>(w);         w += ch * B * sizeof(double);
    double* kl_ws = reinterpret_cast<double*>(w);
    static bool lds_ok[kMaxDevices] = {};
    if (sizeof(SweepSmemT<B>) > 48 * 1024 &&
        !ensure_dynamic_lds(reinterpret_cast<const void*>(kl_sweep_n_kernel<B>), (int)sizeof(SweepSmemT<B>), lds_ok))
        return FQ_ERR_UNSUPPORTED;
    for (int r0 = 0; r0 < rows; r0 += kRowChunk) {
        const int nr = rows - r0 < kRowChunk ? rows - r0 : kRowChunk;
        double* klw = kl_curve_out ? kl_curve_out + (size_t)r0 * kCandB : kl_ws;
        hipLaunchKernelGGL(kl_prepare_kernel<B>, dim3(nr), dim3(kKlBlock), 0, st, reinterpret_cast<const long long*>(hist) + (size_t)r0 * B,
                           Pw, tailw);
        FQ_LAUNCH_CHECK();
        hipLaunchKernelGGL(kl_sweep_n_kernel<B>, dim3(kCandB, nr), dim3(kKlBlock), sizeof(SweepSmemT<B>), st, Pw, tailw, klw);
        FQ_LAUNCH_CHECK();
        hipLaunchKernelGGL(kl_argmin_kernel<B>, dim3(nr), dim3(kKlBlock), 0, st, klw, thr_out + r0, (double*)nullptr, (double*)nullptr);
        FQ_LAUNCH_CHECK();
    }
    return FQ_OK;
}
}  // namespace fq

extern "C" size_t fq_kl_workspace_bytes_n(int rows, int bins) {
    if (rows <= 0 || bins <= FQ_KL_TARGET_BINS) return 0;
    if (bins == FQ_BINS) return fq_kl_workspace_bytes(rows);
    const size_t r = rows < fq::kRowChunk ? (size_t)rows : (size_t)fq::kRowChunk;
    return r * (size_t)(3 * bins - FQ_KL_TARGET_BINS) * sizeof(double) + 64;
}

extern "C" int fq_kl_threshold_n(const int64_t* hist, int rows, int bins, int32_t* thr_out, double* kl_curve_out,
                                 void* workspace, size_t workspace_bytes, fq_stream_t stream) {
    using namespace fq;
    if (bins == FQ_BINS) return fq_kl_threshold(hist, rows, thr_out, kl_curve_out, workspace, workspace_bytes, stream);
    if (bins != 512 && bins != 1024 && bins != 4096) return FQ_ERR_UNSUPPORTED;
    if (rows < 0) return FQ_ERR_INVALID_ARG;
    if (rows == 0) return FQ_OK;
    if (!hist || !thr_out || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 7u)) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    if (bins == 512) return kl_threshold_n<512>(hist, rows, thr_out, kl_curve_out, workspace, workspace_bytes, st);
    if (bins == 1024) return kl_threshold_n<1024>(hist, rows, thr_out, kl_curve_out, workspace, workspace_bytes, st);
    return kl_threshold_n<4096>(hist, rows, thr_out, kl_curve_out, workspace, workspace_bytes, st);
}

extern "C" int fq_kl_threshold(const int64_t* hist, int rows, int32_t* thr_out, double* kl_curve_out,
                               void* workspace, size_t workspace_bytes, fq_stream_t stream) {
    return fq_kl_threshold_ex(hist, rows, thr_out, nullptr, nullptr, kl_curve_out, FQ_KL_AUTO, workspace, workspace_bytes,
                              stream);
}
