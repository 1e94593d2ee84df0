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
__global__ __launch_bounds__(kKlBlock) void kl_prepare_kernel(const long long* __restrict__ hist, double* __restrict__ Pw,
                                                              double* __restrict__ tailw) {
    __shared__ double sP[FQ_BINS];
    __shared__ long long s_part[kKlBlock / kWave];
    const int row = blockIdx.x, tid = threadIdx.x;
    const long long* h = hist + (size_t)row * FQ_BINS;
    long long part = 0;
    for (int j = tid; j < FQ_BINS; j += kKlBlock) part += h[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, kWave);
    if ((tid & (kWave - 1)) == 0) s_part[tid / kWave] = part;
    __syncthreads();
    long long total = 0;
#pragma unroll
    for (int w = 0; w < kKlBlock / kWave; ++w) total += s_part[w];
    const double denom = (double)total + 1e-12;                  // quantizer.py:96
    for (int j = tid; j < FQ_BINS; j += kKlBlock) {
        const double p = (double)(float)h[j] / denom;            // astype(float32) / float64 scalar
        sP[j] = p;
        Pw[(size_t)row * FQ_BINS + j] = p;
    }
    __syncthreads();
    if (tid == 0) {
        double* tail = tailw + (size_t)row * FQ_BINS;
        double ts = pw_sum_serial(sP + kTarget, FQ_BINS - kTarget);      // :100
        tail[kTarget] = ts;
#pragma unroll 8
        for (int t = kTarget; t < FQ_BINS - 1; ++t) {                    // :108
            ts = ts - sP[t];
            tail[t + 1] = ts;
        }
    }
}

// ---- sweep: one workgroup per (threshold, row) --------------------------------------------------
__global__ __launch_bounds__(kKlBlock) void kl_sweep_kernel(const double* __restrict__ Pw, const double* __restrict__ tailw,
                                                            double* __restrict__ klw) {
    __shared__ double sP[FQ_BINS];
    __shared__ double sE[FQ_BINS];            // expand_distribution, later reused for the compacted KL terms
    __shared__ double s_leaf_val[kMaxLeaves];
    __shared__ int s_leaf_off[kMaxLeaves];
    __shared__ int s_leaf_len[kMaxLeaves];
    __shared__ int s_wave_cnt[kKlBlock / kWave];
    __shared__ int s_nleaf;

    const int t = kTarget + blockIdx.x;       // threshold
    const int row = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const double* P = Pw + (size_t)row * FQ_BINS;

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
        q += pw_leaf_nz(sP + lu, rl - lu, &nz);                           // slice .sum(), length <= 16
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

    // KL terms (quantizer.py:169-174) for 8 consecutive bins per lane, compacted over a != 0
    const double tail = tailw[(size_t)row * FQ_BINS + t];
    double term[8];
    unsigned nzmask = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = tid * 8 + k;
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
    for (int k = 0; k < 8; ++k)
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
    {
        const int L = tid >> 3, k = tid & 7;
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
        if (lane == 0) klw[(size_t)row * kCand + blockIdx.x] = v;        // m == 0: np.sum of nothing = 0.0
    }
}

// ---- argmin: first strict minimum below 66666 (quantizer.py:99,:163-165), default 2047 (:101) ----
__global__ __launch_bounds__(kKlBlock) void kl_argmin_kernel(const double* __restrict__ klw, int* __restrict__ thr_out) {
    __shared__ double s_v[kKlBlock];
    __shared__ int s_t[kKlBlock];
    const int row = blockIdx.x, tid = threadIdx.x;
    const double* kl = klw + (size_t)row * kCand;
    double bv = 66666.0;
    int bt = 0x7fffffff;
    for (int i = tid; i < kCand; i += kKlBlock) {
        const double v = kl[i];
        if (v < bv) { bv = v; bt = kTarget + i; }                         // NaN never wins
    }
    s_v[tid] = bv; s_t[tid] = bt;
    __syncthreads();
    for (int s = kKlBlock / 2; s >= 1; s >>= 1) {
        if (tid < s) {
            const double v2 = s_v[tid + s];
            const int t2 = s_t[tid + s];
            if (v2 < s_v[tid] || (v2 == s_v[tid] && t2 < s_t[tid])) { s_v[tid] = v2; s_t[tid] = t2; }
        }
        __syncthreads();
    }
    if (tid == 0) thr_out[row] = (s_t[0] == 0x7fffffff) ? (FQ_BINS - 1) : s_t[0];
}

constexpr size_t kWsPerRow = (size_t)(FQ_BINS + FQ_BINS + kCand) * sizeof(double);

}  // namespace fq

extern "C" size_t fq_kl_workspace_bytes(int rows) {
    return rows > 0 ? (size_t)rows * fq::kWsPerRow : 0;
}

extern "C" int fq_kl_threshold(const int64_t* hist, int rows, int32_t* thr_out, double* kl_curve_out,
                               void* workspace, size_t workspace_bytes, fq_stream_t stream) {
    using namespace fq;
    if (rows < 0) return FQ_ERR_INVALID_ARG;
    if (rows == 0) return FQ_OK;
    if (!hist || !thr_out || !workspace) return FQ_ERR_INVALID_ARG;
    if (workspace_bytes < fq_kl_workspace_bytes(rows)) return FQ_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 7u) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    double* Pw = reinterpret_cast<double*>(workspace);
    double* tailw = Pw + (size_t)rows * FQ_BINS;
    double* klw = kl_curve_out ? kl_curve_out : tailw + (size_t)rows * FQ_BINS;
    const int kChunk = 32768;                                    // grid.y limit
    for (int r0 = 0; r0 < rows; r0 += kChunk) {
        const int nr = rows - r0 < kChunk ? rows - r0 : kChunk;
        hipLaunchKernelGGL(kl_prepare_kernel, dim3(nr), dim3(kKlBlock), 0, st,
                           reinterpret_cast<const long long*>(hist) + (size_t)r0 * FQ_BINS,
                           Pw + (size_t)r0 * FQ_BINS, tailw + (size_t)r0 * FQ_BINS);
        FQ_LAUNCH_CHECK();
        hipLaunchKernelGGL(kl_sweep_kernel, dim3(kCand, nr), dim3(kKlBlock), 0, st, Pw + (size_t)r0 * FQ_BINS,
                           tailw + (size_t)r0 * FQ_BINS, klw + (size_t)r0 * kCand);
        FQ_LAUNCH_CHECK();
        hipLaunchKernelGGL(kl_argmin_kernel, dim3(nr), dim3(kKlBlock), 0, st, klw + (size_t)r0 * kCand, thr_out + r0);
        FQ_LAUNCH_CHECK();
    }
    return FQ_OK;
}
