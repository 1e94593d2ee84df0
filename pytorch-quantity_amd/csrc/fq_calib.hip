// fq_calib.hip -- calibration statistics for gfx950: segmented abs-max and 2048-bin histogram.
//
// Replaces the reference's per-element Python loop and per-call multiprocessing.Pool
// (quantity/common/quantity/distribution_collector.py:70-78 and :80-142).  One launch covers every
// hooked tensor of a forward pass ("segments"): the segments form one stream of 16 KB chunks, every
// workgroup takes an equal share of it and finds its segments through a prefix table that travels in the
// kernel-argument block, so there is no per-tensor launch, no table memcpy and no CU with more bytes than another.
//
// Both kernels are HBM-bound streaming reads (4 B per element, algorithmic bytes = 4 * elements):
//   * 16-byte non-temporal loads, 4 in flight per lane; 256 threads, 2 workgroups per CU, all resident.
//   * histogram bins live in LDS (8 KB per workgroup shared by its 4 waves, ds_add_u32); a
//     workgroup flushes only its non-zero bins to the int64 global rows with 64-bit atomics
//     (T*2048 counters, L2-resident) when it leaves a row.  Bin = trunc(|x| / interval) through a
//     3-instruction quotient proven identical to the IEEE divide (profiles/r01_verify_fastdiv.log).
//   * abs-max keeps a per-lane running max, reduces across the wave with DPP shuffles, across waves
//     through LDS, and publishes with one 32-bit atomic max on the (non-negative) float's bits.
#include <cstdlib>
#include <type_traits>

#include "fq_common.h"
#include "fq_hist_bin.h"
#include "fq_producer_stat.h"

namespace fq {

thread_local int g_last_hip_error = 0;
thread_local int g_last_conv_variant = 0;

constexpr int kSegChunk = 96;          // segments per launch (kernarg block stays < 4 KB)
constexpr int kBlock = 256;            // abs-max: 4 waves per workgroup
constexpr int kHistBlock = 256;        // histogram: 4 waves per LDS histogram (measured optimum;
                                       // 1024-thread workgroups at 2 per CU were 10 % slower)
// workgroups per CU, all resident at once.  Fewer, longer streams are faster (batch 128, 8.6 GB: 1 per CU 6.2 / 4.5 TB/s
// abs-max / histogram, 2: 6.4-6.8 / 6.5-6.7, 4: 6.4-6.8 / 6.3-6.6, 8: 6.2 / 6.4, 16: 5.8 / 6.2 -- box-to-box spread 3 %)
constexpr int kWgPerCUAbsmax = 2;
constexpr int kWgPerCUHist = 2;
constexpr uint32_t kChunkVec = 4 * 256;   // 16-byte vectors per chunk: one workgroup step (4 loads in flight per lane) = 16 KB
constexpr uint32_t kMinChunksPerWg = 4;   // at least 64 KB per workgroup, so zeroing + flushing 2048 bins stays amortised
constexpr int kHistFastQuotientDefault = 1;   // exhaustive proof: profiles/r01_verify_fastdiv.log

// 16-byte streaming load.  The statistics kernels read every activation exactly once and a calibration batch
// (8.6 GB for ResNet-50 at batch 128) is far larger than the 256 MB Infinity Cache, so the lines are marked
// non-temporal: scripts/hbm_read_probe.hip measures 6.1-6.3 TB/s with plain loads, 7.0-7.1 TB/s with these.
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v stream_load(const f4v* p) { return __builtin_nontemporal_load(p); }

// All segments of a launch form ONE virtual stream of 16 KB chunks; every workgroup takes the same number of
// consecutive chunks of that stream, whichever segments they fall into.  (Round 1 gave each workgroup one tile of
// one segment: with 71 segments of 0.5-411 MB the CUs ended up with unequal byte counts and the launch ran at
// 6.1 TB/s where a single 8 GiB segment reached 7.1.)  A workgroup whose share crosses a segment boundary
// publishes its partial result for the finished row and carries on with the next one.
struct SegTable {
    const float* ptr[kSegChunk];
    uint64_t n[kSegChunk];
    uint32_t chunk_begin[kSegChunk + 1];  // exclusive prefix of chunks per segment
    int32_t row[kSegChunk];
    uint32_t chunks_per_wg;
    uint32_t total_chunks;
    int32_t nseg;
};

// chunk index -> segment index (largest s with chunk_begin[s] <= c). Uniform per workgroup.
__device__ __forceinline__ int find_seg(const SegTable& t, uint32_t c) {
    int lo = 0, hi = t.nseg - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (t.chunk_begin[mid] <= c) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The chunks [c0, c1) of one segment: scalar head (elements before the first 16-byte boundary) and scalar tail
// (the last n % 4 elements) belong to the segment's chunk 0; the body is 16-byte vectors, 4 loads in flight per lane.
template <int kThreads, bool kFenceLoads, typename F>
__device__ __forceinline__ void for_each_in_chunks(const float* p, uint64_t cnt, uint32_t c0, uint32_t c1, F&& f) {
    static_assert(kChunkVec == 4 * kThreads, "one chunk = four 16-byte loads per lane");
    const int tid = threadIdx.x;
    const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(p) & 15u) >> 2);
    uint32_t head = mis ? 4u - mis : 0u;
    if (head > cnt) head = (uint32_t)cnt;
    if (c0 == 0 && (uint32_t)tid < head) f(p[tid]);
    p += head;
    cnt -= head;
    const f4v* __restrict__ v4 = reinterpret_cast<const f4v*>(p);
    const uint64_t nvec = cnt >> 2;
    uint64_t base = (uint64_t)c0 * kChunkVec;
    const uint64_t end = (uint64_t)c1 * kChunkVec < nvec ? (uint64_t)c1 * kChunkVec : nvec;
    for (; base + kChunkVec <= end; base += kChunkVec) {
        const uint64_t i = base + tid;
        const f4v a = stream_load(&v4[i]);
        const f4v b = stream_load(&v4[i + kThreads]);
        const f4v c = stream_load(&v4[i + 2 * kThreads]);
        const f4v d = stream_load(&v4[i + 3 * kThreads]);
        // histogram: keep the loads in flight together.  Without this fence the scheduler sinks
        // each load below the previous element group's LDS atomics (one 16-byte load in flight per
        // lane: 5.0 -> 5.5 TB/s with it; 8 fenced loads per lane were slower again, 5.3).  The abs-max
        // loop schedules well on its own and is slower with the fence.
        if (kFenceLoads) __builtin_amdgcn_sched_barrier(0);
        f(a.x); f(a.y); f(a.z); f(a.w);
        f(b.x); f(b.y); f(b.z); f(b.w);
        f(c.x); f(c.y); f(c.z); f(c.w);
        f(d.x); f(d.y); f(d.z); f(d.w);
    }
    for (uint64_t i = base + tid; i < end; i += kThreads) {        // the segment's last, partial chunk
        const f4v a = stream_load(&v4[i]);
        f(a.x); f(a.y); f(a.z); f(a.w);
    }
    const uint32_t tail = (uint32_t)(cnt & 3u);
    if (c0 == 0 && (uint32_t)tid < tail) f(p[(nvec << 2) + tid]);
}

// Walk this workgroup's share of the chunk stream: piece(segment, first chunk, end chunk) per segment touched.
template <typename Piece>
__device__ __forceinline__ void for_each_piece(const SegTable& t, Piece&& piece) {
    uint32_t c = blockIdx.x * t.chunks_per_wg;
    const uint32_t c_end = c + t.chunks_per_wg < t.total_chunks ? c + t.chunks_per_wg : t.total_chunks;
    int s = find_seg(t, c);
    while (c < c_end) {
        const uint32_t seg_end = t.chunk_begin[s + 1];
        const uint32_t stop = seg_end < c_end ? seg_end : c_end;
        piece(s, c - t.chunk_begin[s], stop - t.chunk_begin[s]);
        c = stop;
        ++s;
    }
}

// ---------------------------------------------------------------------------------------------
// abs-max  (distribution_collector.py:70-78)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void absmax_seg_kernel(const SegTable tab, float* __restrict__ max_inout) {
    __shared__ float s_wave[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for_each_piece(tab, [&](int s, uint32_t c0, uint32_t c1) {
        float m = 0.0f;
        for_each_in_chunks<kBlock, false>(tab.ptr[s], tab.n[s], c0, c1, [&](float v) { m = fmaxf(m, fabsf(v)); });   // fmaxf drops NaN
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
        __syncthreads();                              // s_wave of the previous piece has been read
        if (lane == 0) s_wave[wave] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < kBlock / kWave; ++w) m = fmaxf(m, s_wave[w]);
            // m >= 0, so the IEEE bit pattern orders like an unsigned integer
            atomicMax(reinterpret_cast<unsigned int*>(max_inout + tab.row[s]), __float_as_uint(m));
        }
    });
}

// ---------------------------------------------------------------------------------------------
// 2048-bin histogram of |x|, x != 0  (distribution_collector.py:127-135)
// ---------------------------------------------------------------------------------------------
template <bool kFast, int kBins>
__device__ __forceinline__ void hist_piece(const float* p, uint64_t n, uint32_t c0, uint32_t c1, float iv, unsigned int* s_bins) {
    const float y = 1.0f / iv;                        // IEEE, once per lane
    // branch-free: lanes holding an exact zero add into a private scratch slot (kBins + lane)
    unsigned int* park = s_bins + kBins + (threadIdx.x & (kWave - 1));
    for_each_in_chunks<kHistBlock, true>(p, n, c0, c1, [&](float v) {
        unsigned int* slot = (v != 0.0f) ? (s_bins + bin_of<kFast, kBins>(v, iv, y)) : park;
        atomicAdd(slot, 1u);                          // ds_add_u32
    });
}

// kBins = INTERVAL_NUM (tools/configs.yml:24; distribution_collector.py:9-14 is generic in it): 2048 as shipped, the entry point
// fq_hist_seg_n also instantiates 512 / 1024 / 4096
template <int kBins>
__device__ __forceinline__ void hist_seg_body(const SegTable& tab, const float* __restrict__ interval, unsigned long long* __restrict__ hist,
                                              const int allow_fast, unsigned int* s_bins) {
    for (int b = threadIdx.x; b < kBins + kWave; b += kHistBlock) s_bins[b] = 0u;
    __syncthreads();
    for_each_piece(tab, [&](int s, uint32_t c0, uint32_t c1) {
        const int row = tab.row[s];
        const float iv = interval[row];
        // The exhaustive proof of the fast quotient covers every significand pair but assumes that no
        // intermediate under/overflows.  For 2^-60 <= iv <= 2^60 that holds for every element that can
        // land above bin 0 (|x| >= iv >= 2^-60 keeps the fma residual normal; quotients that overflow
        // become inf/nan and fall into the last bin exactly like the IEEE path).  Calibration intervals
        // are max/2048 + 1e-12, far inside that range; anything else takes the IEEE divide.  Uniform per
        // workgroup.
        const unsigned int ivb = __float_as_uint(iv);
        const bool fast = allow_fast && ivb >= 0x21800000u && ivb <= 0x5d800000u;
        if (fast) hist_piece<true, kBins>(tab.ptr[s], tab.n[s], c0, c1, iv, s_bins);
        else hist_piece<false, kBins>(tab.ptr[s], tab.n[s], c0, c1, iv, s_bins);
        __syncthreads();
        // publish the non-zero bins of this row and clear them for the next piece
        unsigned long long* __restrict__ dst = hist + (size_t)row * kBins;
        for (int b = threadIdx.x; b < kBins; b += kHistBlock) {
            const unsigned int c = s_bins[b];
            if (c) {
                atomicAdd(dst + b, (unsigned long long)c);
                s_bins[b] = 0u;
            }
        }
        __syncthreads();
    });
}

__global__ __launch_bounds__(kHistBlock) void hist2048_seg_kernel(const SegTable tab,
                                                              const float* __restrict__ interval,
                                                              unsigned long long* __restrict__ hist,
                                                              const int allow_fast) {
    __shared__ unsigned int s_bins[FQ_BINS + kWave];
    hist_seg_body<FQ_BINS>(tab, interval, hist, allow_fast, s_bins);
}

template <int kBins>
__global__ __launch_bounds__(kHistBlock) void hist_seg_n_kernel(const SegTable tab, const float* __restrict__ interval,
                                                               unsigned long long* __restrict__ hist, const int allow_fast) {
    __shared__ unsigned int s_bins[kBins + kWave];
    hist_seg_body<kBins>(tab, interval, hist, allow_fast, s_bins);
}

// ---------------------------------------------------------------------------------------------
// pairs: the histogram of a tensor a AND of the sum a + b in one pass over both (fq_hist2048_pair_seg)
// ---------------------------------------------------------------------------------------------
// A residual block's Eltwise output is conv3's output + the shortcut.  When pass 1 keeps conv3's output for pass 2 anyway and
// the shortcut (the previous block's ReLU output) exists anyway, the sum need not be written at all: pass 2 reads the two
// operands once and bins a into row_a and fl32(a + b) -- the very addition the Eltwise performs -- into row_sum.  Same bytes
// as reading a and a stored sum; pass 1 writes 4 B per element less.  Both pointers 16-byte aligned (host check).
constexpr int kPairChunk = 48;         // pairs per launch (kernarg block 2.2 KB)
struct PairTable {
    const float* a[kPairChunk];
    const float* b[kPairChunk];
    float* r[kPairChunk];                 // nullable: receives relu(a + b) (the next block's shortcut, re-made instead of kept)
    uint64_t n[kPairChunk];
    uint32_t chunk_begin[kPairChunk + 1];
    int32_t row_a[kPairChunk];            // -1: a is not histogrammed (only the sum)
    int32_t row_s[kPairChunk];
    uint32_t chunks_per_wg;
    uint32_t total_chunks;
    int32_t nseg;
};

// chunks [c0, c1) of one pair: 16-byte vectors, two of each operand in flight per lane; the last n % 4 elements are scalar.
// f(a, b) -> the value to store at the same index of r (when r is given)
template <int kThreads, bool kStore, typename F>
__device__ __forceinline__ void for_each_pair_in_chunks(const float* pa, const float* pb, float* pr, uint64_t cnt, uint32_t c0, uint32_t c1,
                                                        F&& f) {
    const int tid = threadIdx.x;
    const f4v* __restrict__ va = reinterpret_cast<const f4v*>(pa);
    const f4v* __restrict__ vb = reinterpret_cast<const f4v*>(pb);
    f4v* __restrict__ vr = reinterpret_cast<f4v*>(pr);
    const uint64_t nvec = cnt >> 2;
    uint64_t base = (uint64_t)c0 * kChunkVec;
    const uint64_t end = (uint64_t)c1 * kChunkVec < nvec ? (uint64_t)c1 * kChunkVec : nvec;
    auto four = [&](const f4v a, const f4v b, uint64_t i) {
        f4v o;
        o.x = f(a.x, b.x); o.y = f(a.y, b.y); o.z = f(a.z, b.z); o.w = f(a.w, b.w);
        if (kStore) vr[i] = o;
    };
    for (; base + 2 * kThreads <= end; base += 2 * kThreads) {
        const uint64_t i = base + tid;
        const f4v a0 = stream_load(&va[i]);
        const f4v a1 = stream_load(&va[i + kThreads]);
        const f4v b0 = stream_load(&vb[i]);
        const f4v b1 = stream_load(&vb[i + kThreads]);
        __builtin_amdgcn_sched_barrier(0);            // the four loads in flight together (see for_each_in_chunks)
        four(a0, b0, i);
        four(a1, b1, i + kThreads);
    }
    for (uint64_t i = base + tid; i < end; i += kThreads) four(stream_load(&va[i]), stream_load(&vb[i]), i);
    const uint32_t tail = (uint32_t)(cnt & 3u);
    if (c0 == 0 && (uint32_t)tid < tail) {
        const float o = f(pa[(nvec << 2) + tid], pb[(nvec << 2) + tid]);
        if (kStore) pr[(nvec << 2) + tid] = o;
    }
}

template <bool kFastA, bool kFastS, bool kStore>
__device__ __forceinline__ void pair_piece(const float* pa, const float* pb, float* pr, uint64_t n, uint32_t c0, uint32_t c1, float iva,
                                           float ivs, bool want_a, unsigned int* bins_a, unsigned int* bins_s) {
    const float ya = 1.0f / iva, ys = 1.0f / ivs;
    unsigned int* park = bins_s + FQ_BINS + (threadIdx.x & (kWave - 1));
    if (want_a) {
        for_each_pair_in_chunks<kHistBlock, kStore>(pa, pb, pr, n, c0, c1, [&](float a, float b) {
            const float s = a + b;                    // (-ffp-contract=off: one rounded fp32 addition, the Eltwise's)
            atomicAdd((a != 0.0f) ? (bins_a + bin_of<kFastA>(a, iva, ya)) : park, 1u);
            atomicAdd((s != 0.0f) ? (bins_s + bin_of<kFastS>(s, ivs, ys)) : park, 1u);
            return relu_like_torch(s);
        });
    } else {
        for_each_pair_in_chunks<kHistBlock, kStore>(pa, pb, pr, n, c0, c1, [&](float a, float b) {
            const float s = a + b;
            atomicAdd((s != 0.0f) ? (bins_s + bin_of<kFastS>(s, ivs, ys)) : park, 1u);
            return relu_like_torch(s);
        });
    }
}

__global__ __launch_bounds__(kHistBlock) void hist2048_pair_seg_kernel(const PairTable tab, const float* __restrict__ interval,
                                                                   unsigned long long* __restrict__ hist, const int allow_fast) {
    __shared__ unsigned int s_a[FQ_BINS];
    __shared__ unsigned int s_s[FQ_BINS + kWave];
    for (int b = threadIdx.x; b < FQ_BINS; b += kHistBlock) s_a[b] = 0u;
    for (int b = threadIdx.x; b < FQ_BINS + kWave; b += kHistBlock) s_s[b] = 0u;
    __syncthreads();
    uint32_t c = blockIdx.x * tab.chunks_per_wg;
    const uint32_t c_end = c + tab.chunks_per_wg < tab.total_chunks ? c + tab.chunks_per_wg : tab.total_chunks;
    int s = 0;
    {
        int lo = 0, hi = tab.nseg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab.chunk_begin[mid] <= c) lo = mid; else hi = mid - 1;
        }
        s = lo;
    }
    while (c < c_end) {
        const uint32_t seg_end = tab.chunk_begin[s + 1];
        const uint32_t stop = seg_end < c_end ? seg_end : c_end;
        const uint32_t c0 = c - tab.chunk_begin[s], c1 = stop - tab.chunk_begin[s];
        const int ra = tab.row_a[s], rs = tab.row_s[s];
        const float ivs = interval[rs], iva = ra >= 0 ? interval[ra] : 1.0f;
        const bool fa = allow_fast && fast_quotient_ok(iva), fs = allow_fast && fast_quotient_ok(ivs);      // uniform per workgroup
        float* const pr = tab.r[s];
#define FQ_PAIR_PIECE(FA, FS)                                                                                                 \
    do {                                                                                                                      \
        if (pr) pair_piece<FA, FS, true>(tab.a[s], tab.b[s], pr, tab.n[s], c0, c1, iva, ivs, ra >= 0, s_a, s_s);              \
        else pair_piece<FA, FS, false>(tab.a[s], tab.b[s], pr, tab.n[s], c0, c1, iva, ivs, ra >= 0, s_a, s_s);                \
    } while (0)
        if (fa && fs) FQ_PAIR_PIECE(true, true);
        else if (fa) FQ_PAIR_PIECE(true, false);
        else if (fs) FQ_PAIR_PIECE(false, true);
        else FQ_PAIR_PIECE(false, false);
#undef FQ_PAIR_PIECE
        __syncthreads();
        unsigned long long* __restrict__ ds = hist + (size_t)rs * FQ_BINS;
        unsigned long long* __restrict__ da = hist + (size_t)(ra >= 0 ? ra : rs) * FQ_BINS;
        for (int b = threadIdx.x; b < FQ_BINS; b += kHistBlock) {
            const unsigned int cs = s_s[b];
            if (cs) { atomicAdd(ds + b, (unsigned long long)cs); s_s[b] = 0u; }
            const unsigned int ca = s_a[b];
            if (ca) { atomicAdd(da + b, (unsigned long long)ca); s_a[b] = 0u; }
        }
        __syncthreads();
        c = stop;
        ++s;
    }
}

// ---------------------------------------------------------------------------------------------
// chains: a whole stage of residual blocks in one pass (fq_hist2048_chain_seg)
// ---------------------------------------------------------------------------------------------
// The blocks of a stage chain on their shortcut: S_1 = y_1 + head, and for k > 1  S_k = y_k + max(S_{k-1}, 0) -- the shortcut of an
// identity block IS the previous block's ReLU output.  So when pass 1 keeps the conv3 outputs y_1 .. y_L of a stage and the
// stage's first shortcut (the projection's output), neither the sums nor the intermediate shortcuts need exist in HBM: a thread
// reads head[i], y_1[i] .. y_L[i] once, walks the chain in registers and counts 2 L values.  (L + 1) x 4 bytes per position
// instead of the 3 L x 4 of writing, keeping and re-reading every sum.  2 L histograms of 2048 32-bit bins in LDS (dynamic:
// 16 KB per block of the chain), 512 threads, as many workgroups per CU as that leaves room for, each with an equal run of 16 KB
// chunks; the 16-byte loads of all L + 1 streams of two vectors are in flight together.
constexpr int kChainThreadsOnePerCu = 1024;  // L >= 5 leaves LDS for ONE workgroup per CU: 16 waves instead of 8 (stage 3: 294 -> 261 us, profiles/r06_chain_hist_probe.txt)
constexpr int kChainMinChunks = 16;     // chunks (16 KB of every stream) per workgroup at least: amortises zeroing + flushing 2 L x 2048 bins
template <int L>
struct ChainArgs {
    const float* head;
    const float* y[L];
    uint64_t n;
    int32_t row_y[L];                     // -1: y_k is not histogrammed
    int32_t row_s[L];
    uint32_t chunks_per_wg, total_chunks;
};

template <int L, bool kFast, int kChainBlock>
__device__ __forceinline__ void chain_body(const ChainArgs<L>& t, const float* __restrict__ interval, unsigned int* bins) {
    // bins: [2 L][FQ_BINS] then kWave parking slots; histogram 2 k = y_k, 2 k + 1 = S_k
    float ivy[L], ivs[L], ry[L], rs[L];
#pragma unroll
    for (int k = 0; k < L; ++k) {
        ivy[k] = t.row_y[k] >= 0 ? interval[t.row_y[k]] : 1.0f;
        ivs[k] = interval[t.row_s[k]];
        ry[k] = 1.0f / ivy[k];
        rs[k] = 1.0f / ivs[k];
    }
    unsigned int* const park = bins + 2 * L * FQ_BINS + (threadIdx.x & (kWave - 1));
    auto one = [&](float o, const float (&a)[L]) {
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const float s = a[k] + o;                 // (-ffp-contract=off: the Eltwise's one rounded fp32 addition)
            if (t.row_y[k] >= 0) atomicAdd((a[k] != 0.0f) ? (bins + (2 * k) * FQ_BINS + bin_of<kFast>(a[k], ivy[k], ry[k])) : park, 1u);
            atomicAdd((s != 0.0f) ? (bins + (2 * k + 1) * FQ_BINS + bin_of<kFast>(s, ivs[k], rs[k])) : park, 1u);
            o = relu_like_torch(s);                   // the next block's shortcut, as nn.ReLU computes it
        }
    };
    const int tid = threadIdx.x;
    const uint64_t nvec = t.n >> 2;
    const uint32_t c0 = blockIdx.x * t.chunks_per_wg;
    const uint32_t c1 = c0 + t.chunks_per_wg < t.total_chunks ? c0 + t.chunks_per_wg : t.total_chunks;
    uint64_t base = (uint64_t)c0 * kChunkVec;
    const uint64_t end = (uint64_t)c1 * kChunkVec < nvec ? (uint64_t)c1 * kChunkVec : nvec;
    const f4v* __restrict__ vh = reinterpret_cast<const f4v*>(t.head);
    for (; base + 2 * kChainBlock <= end; base += 2 * kChainBlock) {
        const uint64_t i = base + tid;
        const f4v h0 = stream_load(&vh[i]), h1 = stream_load(&vh[i + kChainBlock]);
        f4v y0[L], y1[L];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            y0[k] = stream_load(reinterpret_cast<const f4v*>(t.y[k]) + i);
            y1[k] = stream_load(reinterpret_cast<const f4v*>(t.y[k]) + i + kChainBlock);
        }
        __builtin_amdgcn_sched_barrier(0);            // all 2 (L + 1) loads in flight before the first LDS atomic
        float a[L];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int k = 0; k < L; ++k) a[k] = y0[k][e];
            one(h0[e], a);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int k = 0; k < L; ++k) a[k] = y1[k][e];
            one(h1[e], a);
        }
    }
    for (uint64_t i = base + tid; i < end; i += kChainBlock) {
        const f4v h0 = stream_load(&vh[i]);
        f4v y0[L];
#pragma unroll
        for (int k = 0; k < L; ++k) y0[k] = stream_load(reinterpret_cast<const f4v*>(t.y[k]) + i);
        float a[L];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int k = 0; k < L; ++k) a[k] = y0[k][e];
            one(h0[e], a);
        }
    }
    const uint32_t tail = (uint32_t)(t.n & 3u);
    if (c0 == 0 && (uint32_t)tid < tail) {
        float a[L];
#pragma unroll
        for (int k = 0; k < L; ++k) a[k] = t.y[k][(nvec << 2) + tid];
        one(t.head[(nvec << 2) + tid], a);
    }
}

template <int L, int kChainBlock>
__global__ __launch_bounds__(kChainBlock) void hist2048_chain_kernel(const ChainArgs<L> t, const float* __restrict__ interval,
                                                                  unsigned long long* __restrict__ hist, const int allow_fast) {
    extern __shared__ __attribute__((aligned(16))) unsigned int chain_bins[];
    for (int b = threadIdx.x; b < 2 * L * FQ_BINS + kWave; b += kChainBlock) chain_bins[b] = 0u;
    __syncthreads();
    bool fast = allow_fast != 0;                      // the 3-instruction quotient only when EVERY row's interval allows it (uniform)
#pragma unroll
    for (int k = 0; k < L; ++k) {
        if (t.row_y[k] >= 0) fast = fast && fast_quotient_ok(interval[t.row_y[k]]);
        fast = fast && fast_quotient_ok(interval[t.row_s[k]]);
    }
    if (fast) chain_body<L, true, kChainBlock>(t, interval, chain_bins);
    else chain_body<L, false, kChainBlock>(t, interval, chain_bins);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < L; ++k) {
        if (t.row_y[k] >= 0) {
            unsigned long long* __restrict__ d = hist + (size_t)t.row_y[k] * FQ_BINS;
            for (int b = threadIdx.x; b < FQ_BINS; b += kChainBlock) {
                const unsigned int c = chain_bins[(2 * k) * FQ_BINS + b];
                if (c) atomicAdd(d + b, (unsigned long long)c);
            }
        }
        unsigned long long* __restrict__ d = hist + (size_t)t.row_s[k] * FQ_BINS;
        for (int b = threadIdx.x; b < FQ_BINS; b += kChainBlock) {
            const unsigned int c = chain_bins[(2 * k + 1) * FQ_BINS + b];
            if (c) atomicAdd(d + b, (unsigned long long)c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-channel rows: one histogram row per (tensor, channel) of an NCHW activation
// ---------------------------------------------------------------------------------------------
// A row is no longer one contiguous run: channel c of tensor [N][C][HW] is the N planes at (n*C + c)*HW.  The tensor is
// read in place (no channel-major copy); blockIdx -> (tensor, channel block, image group) through a kernarg prefix
// table as above.  Two shapes of workgroup, both sweeping their elements as ONE index space with four 16-byte loads in
// flight per lane (round 1 let each wave walk one plane at a time: a 28x28 plane is 196 vectors, three per lane, so the
// 4-deep loop never ran and the kernel sat at 4.1 TB/s):
//   big planes (HW >= 1024, 16-byte aligned): a workgroup owns ONE channel and a group of images; its LDS histogram
//     belongs to a single row (32-bit bins, up to 256 K elements per workgroup);
//   small planes: a workgroup owns a block of 8 CONSECUTIVE channels and a group of images, i.e. one contiguous run of
//     8*HW floats per image (a 7x7 plane is 196 bytes: owning one channel would use a quarter of every cache line it
//     touches, and the neighbouring channels' workgroups sit on other XCDs); 8 histograms of 2048 16-bit bins packed
//     two to a dword (32 KB of LDS; a workgroup never gives one row more than 32 768 elements, so a bin cannot carry
//     into its neighbour).
constexpr int kChanChunk = 64;            // kernarg block 3.4 KB; ResNet-50's 71 tensors take two launches
constexpr uint32_t kChanElemsPerWg = 262144;   // per workgroup: enough to amortise zeroing + flushing the bins
constexpr int kChanBlock = 8;             // channels per workgroup, small planes
constexpr uint32_t kBigPlane = 1024;

struct ChanTable {
    const float* ptr[kChanChunk];
    uint32_t N[kChanChunk], C[kChanChunk], HW[kChanChunk];
    uint32_t groups[kChanChunk], nb[kChanChunk];          // image groups, images per group
    uint32_t cb[kChanChunk];                              // channels per workgroup: 1 (big planes) or kChanBlock
    uint32_t vec[kChanChunk];                             // 16-byte loads possible
    uint32_t own[kChanChunk];                             // 8-channel blocks: every row of this tensor belongs to ONE workgroup of this call
    int32_t row0[kChanChunk];
    uint32_t wg_begin[kChanChunk + 1];
    int32_t nseg;
};

struct ChanView {
    const float* base;          // first element of (image n0, channel c0)
    size_t img_stride;          // C * HW
    uint32_t HW, run;           // run = cb * HW contiguous floats per image
    uint32_t nimg, cb, vec, own;
    int row;                    // row of channel c0
};

__device__ __forceinline__ ChanView chan_of(const ChanTable& t) {
    const uint32_t b = blockIdx.x;
    int lo = 0, hi = t.nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.wg_begin[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const uint32_t local = b - t.wg_begin[lo];
    const uint32_t cblk = local / t.groups[lo], g = local - cblk * t.groups[lo];
    const uint32_t c0 = cblk * t.cb[lo];
    const uint32_t n0 = g * t.nb[lo];
    const uint32_t n1 = n0 + t.nb[lo] < t.N[lo] ? n0 + t.nb[lo] : t.N[lo];
    ChanView v;
    v.HW = t.HW[lo];
    v.cb = c0 + t.cb[lo] <= t.C[lo] ? t.cb[lo] : t.C[lo] - c0;        // ragged last block
    v.run = v.cb * v.HW;
    v.img_stride = (size_t)t.C[lo] * v.HW;
    v.nimg = n1 - n0;
    v.base = t.ptr[lo] + (size_t)n0 * v.img_stride + (size_t)c0 * v.HW;
    v.row = t.row0[lo] + (int)c0;
    v.vec = t.vec[lo] && ((v.run & 3u) == 0);
    v.own = t.own[lo];
    return v;
}

// n / d for n * d < 2^32 (here n < 2^19, d < 2^13): one multiply-high
__device__ __forceinline__ uint32_t magic_of(uint32_t d) { return (uint32_t)((0x100000000ull + d - 1) / d); }
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t magic, uint32_t d) { return d == 1 ? n : __umulhi(n, magic); }

// f(value, channel within the block) for every element of this workgroup: the images' runs of cv.run floats form one
// index space; 16-byte loads (four in flight per lane) when every run starts on a 16-byte boundary
template <int kThreads, bool kFenceLoads, typename F>
__device__ __forceinline__ void for_each_in_block(const ChanView& cv, F&& f) {
    const uint32_t tid = threadIdx.x;
    const uint32_t m_hw = magic_of(cv.HW);
    if (cv.vec) {
        const uint32_t runv = cv.run >> 2;
        const uint32_t m_run = magic_of(runv);
        const uint32_t total = cv.nimg * runv;
        const size_t stridev = cv.img_stride >> 2;
        const f4v* __restrict__ v4 = reinterpret_cast<const f4v*>(cv.base);
        auto locate = [&](uint32_t v, uint32_t& e0) -> const f4v* {
            const uint32_t img = cv.nimg == 1 ? 0u : div_magic(v, m_run, runv);    // (one image: any plane size)
            const uint32_t vin = v - img * runv;
            e0 = vin << 2;
            return v4 + (size_t)img * stridev + vin;
        };
        auto emit = [&](const f4v a, uint32_t e0) {
            if (cv.cb == 1) { f(a.x, 0u); f(a.y, 0u); f(a.z, 0u); f(a.w, 0u); return; }
            const uint32_t ch = div_magic(e0, m_hw, cv.HW);
            const uint32_t rem = e0 - ch * cv.HW;                  // elements of this vector may cross into channel ch + 1
            if (cv.HW >= 4) {
                f(a.x, ch); f(a.y, ch + (rem + 1 >= cv.HW)); f(a.z, ch + (rem + 2 >= cv.HW)); f(a.w, ch + (rem + 3 >= cv.HW));
            } else {
                f(a.x, ch); f(a.y, div_magic(e0 + 1, m_hw, cv.HW)); f(a.z, div_magic(e0 + 2, m_hw, cv.HW));
                f(a.w, div_magic(e0 + 3, m_hw, cv.HW));
            }
        };
        uint32_t v = tid;
        for (; v + 3 * kThreads < total; v += 4 * kThreads) {
            uint32_t ea, eb, ec, ed;
            const f4v a = stream_load(locate(v, ea));
            const f4v b = stream_load(locate(v + kThreads, eb));
            const f4v c = stream_load(locate(v + 2 * kThreads, ec));
            const f4v d = stream_load(locate(v + 3 * kThreads, ed));
            if (kFenceLoads) __builtin_amdgcn_sched_barrier(0);       // see for_each_in_chunks
            emit(a, ea); emit(b, eb); emit(c, ec); emit(d, ed);
        }
        for (; v < total; v += kThreads) {
            uint32_t ea;
            const f4v a = stream_load(locate(v, ea));
            emit(a, ea);
        }
    } else {
        const uint32_t m_run = magic_of(cv.run);
        const uint32_t total = cv.nimg * cv.run;
        for (uint32_t e = tid; e < total; e += kThreads) {
            const uint32_t img = cv.nimg == 1 ? 0u : div_magic(e, m_run, cv.run);
            const uint32_t ein = e - img * cv.run;
            f(cv.base[(size_t)img * cv.img_stride + ein], cv.cb == 1 ? 0u : div_magic(ein, m_hw, cv.HW));
        }
    }
}

__global__ __launch_bounds__(kBlock) void absmax_chan_kernel(const ChanTable tab, float* __restrict__ max_inout) {
    __shared__ unsigned int s_m[kChanBlock];
    __shared__ float s_wave[kBlock / kWave];
    const ChanView cv = chan_of(tab);
    if (cv.cb == 1) {                                         // one row: per-lane running maximum
        float m = 0.0f;
        for_each_in_block<kBlock, false>(cv, [&](float v, uint32_t) { m = fmaxf(m, fabsf(v)); });
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
        if (lane == 0) s_wave[wave] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < kBlock / kWave; ++w) m = fmaxf(m, s_wave[w]);
            atomicMax(reinterpret_cast<unsigned int*>(max_inout + cv.row), __float_as_uint(m));
        }
        return;
    }
    if (threadIdx.x < kChanBlock) s_m[threadIdx.x] = 0u;
    __syncthreads();
    // a lane keeps the maximum of the channel it is in and publishes it (ds_max_u32) when the channel changes
    uint32_t cur = 0xffffffffu;
    float m = 0.0f;
    for_each_in_block<kBlock, false>(cv, [&](float v, uint32_t ch) {
        if (ch != cur) {
            if (cur != 0xffffffffu) atomicMax(&s_m[cur], __float_as_uint(m));
            cur = ch; m = 0.0f;
        }
        m = fmaxf(m, fabsf(v));                               // fmaxf drops NaN
    });
    if (cur != 0xffffffffu) atomicMax(&s_m[cur], __float_as_uint(m));
    __syncthreads();
    if (threadIdx.x < cv.cb) atomicMax(reinterpret_cast<unsigned int*>(max_inout + cv.row + threadIdx.x), s_m[threadIdx.x]);
}

// The flush of a row that ONE workgroup owns for the whole call (every image of the batch in one group, no other segment on the
// row; the host decides: ChanTable::own): four 64-bit counters per lane read, added to and written back as two 16-byte accesses
// each way instead of up to four 64-bit atomics.  A 14x14 plane gives a row 50 K elements per 256 images, a 7x7 plane 12.5 K --
// against up to 2 048 bins to publish: with atomics (memory-side, 8 bytes each) the flush of the 34 000 small-plane rows of
// ResNet-50 was a third of the kernel.  Launches on one stream are ordered, so the previous batch's sums are visible.
__device__ __forceinline__ void own_add4(unsigned long long* __restrict__ d, unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2* const p = reinterpret_cast<u64x2*>(d);
    u64x2 a = p[0], b = p[1];
    a.x += c0; a.y += c1; b.x += c2; b.y += c3;
    p[0] = a; p[1] = b;
}

__global__ __launch_bounds__(kHistBlock) void hist2048_chan_kernel(const ChanTable tab, const float* __restrict__ interval,
                                                               unsigned long long* __restrict__ hist, const int allow_fast) {
    // big planes: s_bins[0 .. 2047] 32-bit bins of one row (+ 64 parking slots)
    // small planes: row r's bin b is the 16-bit half (b & 1) of dword r * 1024 + (b >> 1); parking dwords behind
    __shared__ unsigned int s_bins[kChanBlock * (FQ_BINS / 2) + kWave];
    __shared__ float2 s_ivy[kChanBlock];
    __shared__ int s_fast;
    const ChanView cv = chan_of(tab);
    const int nwords = cv.cb == 1 ? FQ_BINS : kChanBlock * (FQ_BINS / 2);
    for (int b = threadIdx.x; b < nwords + kWave; b += kHistBlock) s_bins[b] = 0u;
    if (threadIdx.x == 0) s_fast = allow_fast;
    __syncthreads();
    if (threadIdx.x < cv.cb) {
        const float iv = interval[cv.row + threadIdx.x];
        s_ivy[threadIdx.x] = make_float2(iv, 1.0f / iv);      // IEEE
        const unsigned int ivb = __float_as_uint(iv);
        if (!(ivb >= 0x21800000u && ivb <= 0x5d800000u)) s_fast = 0;     // see hist2048_seg_kernel; any row outside: IEEE divide
    }
    __syncthreads();
    const bool fast = s_fast != 0;
    unsigned int* park = s_bins + nwords + (threadIdx.x & (kWave - 1));
    unsigned long long* __restrict__ dst = hist + (size_t)cv.row * FQ_BINS;
    if (cv.cb == 1) {
        const float iv = s_ivy[0].x, y = s_ivy[0].y;
        if (fast) for_each_in_block<kHistBlock, true>(cv, [&](float v, uint32_t) {
            atomicAdd((v != 0.0f) ? (s_bins + bin_of<true>(v, iv, y)) : park, 1u); });
        else for_each_in_block<kHistBlock, true>(cv, [&](float v, uint32_t) {
            atomicAdd((v != 0.0f) ? (s_bins + bin_of<false>(v, iv, y)) : park, 1u); });
        __syncthreads();
        for (int b = threadIdx.x; b < FQ_BINS; b += kHistBlock) {
            const unsigned int c = s_bins[b];
            if (c) atomicAdd(dst + b, (unsigned long long)c);
        }
        return;
    }
    auto add = [&](float v, uint32_t ch, auto fast_tag) {
        const float2 p = s_ivy[ch];                                        // one ds_read_b64: (interval, 1 / interval)
        const int b = bin_of<decltype(fast_tag)::value>(v, p.x, p.y);
        unsigned int* slot = (v != 0.0f) ? (s_bins + ch * (FQ_BINS / 2) + (b >> 1)) : park;
        atomicAdd(slot, (v != 0.0f && (b & 1)) ? 0x10000u : 1u);          // ds_add_u32 on a 16-bit half
    };
    if (fast) for_each_in_block<kHistBlock, true>(cv, [&](float v, uint32_t ch) { add(v, ch, std::true_type{}); });
    else for_each_in_block<kHistBlock, true>(cv, [&](float v, uint32_t ch) { add(v, ch, std::false_type{}); });
    __syncthreads();
    if (cv.own) {                                             // four consecutive bins (two packed dwords) per lane and step
        for (uint32_t q = threadIdx.x; q < cv.cb * (FQ_BINS / 4); q += kHistBlock) {
            const uint2 c = reinterpret_cast<const uint2*>(s_bins)[q];
            if (c.x | c.y) own_add4(dst + 4u * (size_t)q, c.x & 0xffffu, c.x >> 16, c.y & 0xffffu, c.y >> 16);
        }
        return;
    }
    for (uint32_t w = threadIdx.x; w < cv.cb * (FQ_BINS / 2); w += kHistBlock) {
        const unsigned int c = s_bins[w];
        if (c) {
            const uint32_t r = w / (FQ_BINS / 2), b = (w - r * (FQ_BINS / 2)) * 2;
            unsigned long long* d = dst + (size_t)r * FQ_BINS + b;
            if (c & 0xffffu) atomicAdd(d, (unsigned long long)(c & 0xffffu));
            if (c >> 16) atomicAdd(d + 1, (unsigned long long)(c >> 16));
        }
    }
}

template <typename Launch>
static int for_each_chan_chunk(const fq_chan_seg* segs, int nseg, Launch&& launch, bool may_own = false) {
    if (nseg < 0 || nseg > FQ_MAX_SEGS || (nseg > 0 && segs == nullptr)) return FQ_ERR_INVALID_ARG;
    for (int i = 0; i < nseg; ++i) {
        const fq_chan_seg& s = segs[i];
        if (s.N < 0 || s.C <= 0 || s.HW <= 0 || s.row0 < 0 || s.reserved != 0) return FQ_ERR_INVALID_ARG;
        if (s.N > 0 && (s.ptr == nullptr || (reinterpret_cast<uintptr_t>(s.ptr) & 3u))) return FQ_ERR_INVALID_ARG;
        if (s.HW > 0x7fffffffLL) return FQ_ERR_UNSUPPORTED;
    }
    int i = 0;
    while (i < nseg) {
        ChanTable tab;
        int k = 0;
        uint64_t wgs = 0;
        while (i < nseg && k < kChanChunk) {
            const fq_chan_seg& s = segs[i++];
            if (s.N == 0) continue;
            const bool aligned = (reinterpret_cast<uintptr_t>(s.ptr) & 15u) == 0;
            // one channel per workgroup wherever its planes can be read with 16-byte loads and are at least one
            // wave-load long (measured per shape, batch 128: 28x28 planes 2.8 -> one-channel form; 7x7 planes, 196 bytes
            // and never 16-byte aligned, stay with the 8-channel blocks), and for every plane of 1024 elements or more
            const bool planes_vec = aligned && (s.HW & 3) == 0 && (((uint64_t)s.C * (uint64_t)s.HW) & 3u) == 0;
            const bool big = (uint64_t)s.HW >= kBigPlane || (planes_vec && s.HW >= 64);
            uint32_t cb, nb, vec;
            if (big) {
                // one channel per workgroup; planes of more than 256 K elements are still one image per workgroup
                cb = 1;
                vec = planes_vec;
                nb = (uint32_t)(kChanElemsPerWg / (uint64_t)s.HW);
                if (nb < 1) nb = 1;
                // the magic division by the run length needs n * d < 2^32; a single image per workgroup needs none
                const uint64_t runl = (uint64_t)s.HW / (vec ? 4 : 1);
                while (nb > 1 && (uint64_t)nb * runl * runl >= (1ull << 32)) nb >>= 1;
            } else {
                cb = kChanBlock;
                // every image's run of cb * HW floats starts 16-byte aligned when the base is, the image stride is a
                // multiple of 4 floats and so is a full block (the ragged last block is checked in the kernel)
                vec = aligned && (((uint64_t)s.C * (uint64_t)s.HW) & 3u) == 0 && (((uint64_t)cb * (uint64_t)s.HW) & 3u) == 0;
                nb = (uint32_t)(32768u / (uint64_t)s.HW);         // <= 32 768 elements per row and workgroup: 16-bit bins
                if (nb < 1) nb = 1;
            }
            if (nb > (uint32_t)s.N) nb = (uint32_t)s.N;
            const uint32_t groups = ((uint32_t)s.N + nb - 1) / nb;
            // a row is one workgroup's own when the batch is a single image group and no other segment of the call feeds the row.
            // Used for the 8-channel blocks only (measured at 256 images, steady state, scripts/chan_hist_probe.py: 7x7 planes
            // 2.5 -> 3.1 TB/s -- 8 x 2 048 counters to publish per 100 K elements; one-channel workgroups on 14x14 / 28x28
            // planes 5.2 -> 4.65 / 5.9 -> 5.5: their atomics are fire-and-forget, the read-add-write is a round trip at the end
            // of a workgroup that has nothing left to overlap it with)
            bool own = may_own && groups == 1 && cb != 1;
            for (int j = 0; own && j < nseg; ++j) {
                const fq_chan_seg& o = segs[j];
                if (&o != &s && o.N != 0 && (int64_t)o.row0 < (int64_t)s.row0 + s.C && (int64_t)s.row0 < (int64_t)o.row0 + o.C) own = false;
            }
            const uint64_t n_wg = (uint64_t)groups * (((uint64_t)s.C + cb - 1) / cb);
            if (wgs + n_wg > 0x7fffffffULL) { --i; break; }
            tab.ptr[k] = s.ptr; tab.N[k] = (uint32_t)s.N; tab.C[k] = (uint32_t)s.C; tab.HW[k] = (uint32_t)s.HW;
            tab.groups[k] = groups; tab.nb[k] = nb; tab.cb[k] = cb; tab.vec[k] = vec; tab.row0[k] = s.row0; tab.own[k] = own ? 1u : 0u;
            tab.wg_begin[k] = (uint32_t)wgs;
            wgs += n_wg;
            ++k;
        }
        if (k == 0) {
            if (i < nseg && segs[i].N != 0) return FQ_ERR_INVALID_ARG;
            continue;
        }
        tab.nseg = k;
        for (int j = k; j <= kChanChunk; ++j) tab.wg_begin[j] = (uint32_t)wgs;
        for (int j = k; j < kChanChunk; ++j) {
            tab.ptr[j] = nullptr; tab.N[j] = 0; tab.C[j] = 1; tab.HW[j] = 1; tab.groups[j] = 1; tab.nb[j] = 1; tab.cb[j] = 1;
            tab.vec[j] = 0; tab.row0[j] = 0; tab.own[j] = 0;
        }
        const int rc = launch(tab, (uint32_t)wgs);
        if (rc != FQ_OK) return rc;
    }
    return FQ_OK;
}

// ---------------------------------------------------------------------------------------------
// host side: tiling and chunked launches
// ---------------------------------------------------------------------------------------------
// FQ_HIST_IEEE_DIV=1 forces the IEEE divide sequence (A/B timing, paranoia).
static int hist_fast_quotient_enabled() {
    static const int v = [] {
        const char* e = getenv("FQ_HIST_IEEE_DIV");
        return (e && e[0] && e[0] != '0') ? 0 : kHistFastQuotientDefault;
    }();
    return v;
}

// FQ_WG_PER_CU overrides the workgroups per CU of both statistics kernels (tuning knob).
static int wg_per_cu(int default_per_cu) {
    static const int env_per_cu = [] {
        const char* e = getenv("FQ_WG_PER_CU");
        return e ? atoi(e) : 0;
    }();
    return env_per_cu > 0 ? env_per_cu : default_per_cu;
}

// chunks of one segment: 16-byte vectors after the scalar head, in units of kChunkVec; at least one (head / tail)
static uint64_t chunks_of(const fq_seg& s) {
    const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(s.ptr) & 15u) >> 2);
    uint64_t head = mis ? 4u - mis : 0u;
    if (head > s.n) head = s.n;
    const uint64_t nvec = (s.n - head) >> 2;
    const uint64_t c = (nvec + fq::kChunkVec - 1) / fq::kChunkVec;
    return c ? c : 1;
}

static int validate(const fq_seg* segs, int nseg) {
    if (nseg < 0 || nseg > FQ_MAX_SEGS) return FQ_ERR_INVALID_ARG;
    if (nseg > 0 && segs == nullptr) return FQ_ERR_INVALID_ARG;
    for (int i = 0; i < nseg; ++i) {
        if (segs[i].row < 0 || segs[i].reserved != 0) return FQ_ERR_INVALID_ARG;
        if (segs[i].n != 0 && segs[i].ptr == nullptr) return FQ_ERR_INVALID_ARG;
        if (reinterpret_cast<uintptr_t>(segs[i].ptr) & 3u) return FQ_ERR_INVALID_ARG;
    }
    return FQ_OK;
}

template <typename Launch>
static int for_each_chunk(const fq_seg* segs, int nseg, int per_cu, Launch&& launch) {
    int i = 0;
    while (i < nseg) {
        SegTable tab;
        int k = 0;
        uint64_t chunks = 0;
        while (i < nseg && k < kSegChunk) {
            const fq_seg& s = segs[i++];
            if (s.n == 0) continue;
            const uint64_t nc = chunks_of(s);
            if (chunks + nc > 0x7fffffffULL) { --i; break; }    // 32-bit chunk index: start a new launch
            tab.ptr[k] = s.ptr;
            tab.n[k] = s.n;
            tab.row[k] = s.row;
            tab.chunk_begin[k] = (uint32_t)chunks;
            chunks += nc;
            ++k;
        }
        if (k == 0) {
            if (i < nseg && segs[i].n != 0) return FQ_ERR_INVALID_ARG;   // single segment over the index limit
            continue;
        }
        tab.nseg = k;
        for (int j = k; j <= kSegChunk; ++j) tab.chunk_begin[j] = (uint32_t)chunks;
        for (int j = k; j < kSegChunk; ++j) { tab.ptr[j] = nullptr; tab.n[j] = 0; tab.row[j] = 0; }
        // equal shares of the chunk stream, every workgroup resident at once
        const uint64_t slots = (uint64_t)kCUs * wg_per_cu(per_cu);
        uint64_t per_wg = (chunks + slots - 1) / slots;
        if (per_wg < kMinChunksPerWg) per_wg = kMinChunksPerWg;
        tab.chunks_per_wg = (uint32_t)per_wg;
        tab.total_chunks = (uint32_t)chunks;
        int rc = launch(tab, (uint32_t)((chunks + per_wg - 1) / per_wg));
        if (rc != FQ_OK) return rc;
    }
    return FQ_OK;
}

}  // namespace fq

extern "C" int fq_absmax_seg(const fq_seg* segs, int nseg, float* max_inout, fq_stream_t stream) {
    using namespace fq;
    int rc = validate(segs, nseg);
    if (rc != FQ_OK) return rc;
    if (nseg == 0) return FQ_OK;
    if (max_inout == nullptr) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    return for_each_chunk(segs, nseg, kWgPerCUAbsmax, [&](const SegTable& tab, uint32_t wgs) -> int {
        hipLaunchKernelGGL(absmax_seg_kernel, dim3(wgs), dim3(kBlock), 0, st, tab, max_inout);
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    });
}

extern "C" int fq_hist2048_seg(const fq_seg* segs, int nseg, const float* interval, int64_t* hist,
                               fq_stream_t stream) {
    using namespace fq;
    int rc = validate(segs, nseg);
    if (rc != FQ_OK) return rc;
    if (nseg == 0) return FQ_OK;
    if (interval == nullptr || hist == nullptr) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    return for_each_chunk(segs, nseg, kWgPerCUHist, [&](const SegTable& tab, uint32_t wgs) -> int {
        hipLaunchKernelGGL(hist2048_seg_kernel, dim3(wgs), dim3(kHistBlock), 0, st, tab, interval,
                           reinterpret_cast<unsigned long long*>(hist), hist_fast_quotient_enabled());
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    });
}

extern "C" int fq_hist_seg_n(const fq_seg* segs, int nseg, const float* interval, int64_t* hist, int bins, fq_stream_t stream) {
    using namespace fq;
    if (bins == FQ_BINS) return fq_hist2048_seg(segs, nseg, interval, hist, stream);
    if (bins != 512 && bins != 1024 && bins != 4096) return FQ_ERR_UNSUPPORTED;
    int rc = validate(segs, nseg);
    if (rc != FQ_OK) return rc;
    if (nseg == 0) return FQ_OK;
    if (interval == nullptr || hist == nullptr) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    return for_each_chunk(segs, nseg, kWgPerCUHist, [&](const SegTable& tab, uint32_t wgs) -> int {
        unsigned long long* h = reinterpret_cast<unsigned long long*>(hist);
        if (bins == 512) hipLaunchKernelGGL(hist_seg_n_kernel<512>, dim3(wgs), dim3(kHistBlock), 0, st, tab, interval, h, hist_fast_quotient_enabled());
        else if (bins == 1024) hipLaunchKernelGGL(hist_seg_n_kernel<1024>, dim3(wgs), dim3(kHistBlock), 0, st, tab, interval, h, hist_fast_quotient_enabled());
        else hipLaunchKernelGGL(hist_seg_n_kernel<4096>, dim3(wgs), dim3(kHistBlock), 0, st, tab, interval, h, hist_fast_quotient_enabled());
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    });
}

namespace fq {
template <int L, int TB>
static int launch_chain_tb(const ChainArgs<L>& a, uint64_t chunks, uint64_t per_wg, size_t lds, const float* interval, int64_t* hist, hipStream_t st) {
    static bool lds_ok[kMaxDevices] = {};
    if (lds > 64 * 1024 && !ensure_dynamic_lds(reinterpret_cast<const void*>(hist2048_chain_kernel<L, TB>), (int)lds, lds_ok)) return FQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((hist2048_chain_kernel<L, TB>), dim3((uint32_t)((chunks + per_wg - 1) / per_wg)), dim3(TB), lds, st, a, interval,
                       reinterpret_cast<unsigned long long*>(hist), hist_fast_quotient_enabled());
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

template <int L>
static int launch_chain(const fq_chain_seg& c, const float* interval, int64_t* hist, hipStream_t st) {
    ChainArgs<L> a;
    a.head = c.head;
    a.n = c.n;
    for (int k = 0; k < L; ++k) { a.y[k] = c.y[k]; a.row_y[k] = c.row_y[k]; a.row_s[k] = c.row_sum[k]; }
    const uint64_t nvec = c.n >> 2;
    uint64_t chunks = (nvec + kChunkVec - 1) / kChunkVec;
    if (chunks == 0) chunks = 1;
    if (chunks > 0x7fffffffULL) return FQ_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(2 * L * FQ_BINS + kWave) * sizeof(unsigned int);
    int per_cu = (int)((size_t)150 * 1024 / lds);                 // 160 KB of LDS per CU, some of it the runtime's
    if (per_cu > 3) per_cu = 3;
    if (per_cu < 1) per_cu = 1;
    // FQ_CHAIN_WG_PER_CU / FQ_CHAIN_MIN_CHUNKS / FQ_CHAIN_THREADS: tuning knobs (scripts/chain_hist_probe.py)
    static const int per_cu_env = [] { const char* e = getenv("FQ_CHAIN_WG_PER_CU"); return e ? atoi(e) : 0; }();
    static const int min_chunks_env = [] { const char* e = getenv("FQ_CHAIN_MIN_CHUNKS"); return e ? atoi(e) : 0; }();
    static const int threads_env = [] { const char* e = getenv("FQ_CHAIN_THREADS"); return e ? atoi(e) : 0; }();
    if (per_cu_env > 0 && per_cu_env < per_cu) per_cu = per_cu_env;
    const uint64_t slots = (uint64_t)kCUs * per_cu;
    uint64_t per_wg = (chunks + slots - 1) / slots;
    const uint64_t min_chunks = min_chunks_env > 0 ? (uint64_t)min_chunks_env : (uint64_t)kChainMinChunks;
    if (per_wg < min_chunks) per_wg = min_chunks;
    a.chunks_per_wg = (uint32_t)per_wg;
    a.total_chunks = (uint32_t)chunks;
    // a workgroup of 16 waves where the LDS leaves room for one workgroup only (L >= 5): as many waves per CU as two of 8
    const int threads = threads_env == 512 || threads_env == 1024 ? threads_env : (per_cu == 1 ? kChainThreadsOnePerCu : 512);
    if (threads == 1024) return launch_chain_tb<L, 1024>(a, chunks, per_wg, lds, interval, hist, st);
    return launch_chain_tb<L, 512>(a, chunks, per_wg, lds, interval, hist, st);
}

}  // namespace fq

extern "C" int fq_hist2048_chain_seg(const fq_chain_seg* segs, int nseg, const float* interval, int64_t* hist, fq_stream_t stream) {
    using namespace fq;
    if (nseg < 0 || nseg > FQ_MAX_SEGS || (nseg > 0 && segs == nullptr)) return FQ_ERR_INVALID_ARG;
    for (int i = 0; i < nseg; ++i) {
        const fq_chain_seg& c = segs[i];
        if (c.len < 1 || c.len > FQ_CHAIN_MAX) return FQ_ERR_INVALID_ARG;
        uintptr_t bits = reinterpret_cast<uintptr_t>(c.head);
        for (int k = 0; k < c.len; ++k) {
            if (c.row_sum[k] < 0 || c.row_y[k] < -1 || c.row_y[k] == c.row_sum[k]) return FQ_ERR_INVALID_ARG;
            if (c.n != 0 && c.y[k] == nullptr) return FQ_ERR_INVALID_ARG;
            bits |= reinterpret_cast<uintptr_t>(c.y[k]);
        }
        if (c.n != 0 && c.head == nullptr) return FQ_ERR_INVALID_ARG;
        if (bits & 3u) return FQ_ERR_INVALID_ARG;
        if (c.n != 0 && (bits & 15u)) return FQ_ERR_UNSUPPORTED;
    }
    if (nseg == 0) return FQ_OK;
    if (interval == nullptr || hist == nullptr) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    for (int i = 0; i < nseg; ++i) {                       // one launch per chain: the chains of a forward differ in length and size
        const fq_chain_seg& c = segs[i];
        if (c.n == 0) continue;
        int rc;
        switch (c.len) {
            case 1: rc = launch_chain<1>(c, interval, hist, st); break;
            case 2: rc = launch_chain<2>(c, interval, hist, st); break;
            case 3: rc = launch_chain<3>(c, interval, hist, st); break;
            case 4: rc = launch_chain<4>(c, interval, hist, st); break;
            case 5: rc = launch_chain<5>(c, interval, hist, st); break;
            default: rc = launch_chain<6>(c, interval, hist, st); break;
        }
        if (rc != FQ_OK) return rc;
    }
    return FQ_OK;
}

extern "C" int fq_hist2048_pair_seg(const fq_pair_seg* segs, int nseg, const float* interval, int64_t* hist, fq_stream_t stream) {
    using namespace fq;
    if (nseg < 0 || nseg > FQ_MAX_SEGS || (nseg > 0 && segs == nullptr)) return FQ_ERR_INVALID_ARG;
    for (int i = 0; i < nseg; ++i) {
        const fq_pair_seg& p = segs[i];
        if (p.row_sum < 0 || p.row_a < -1 || p.row_a == p.row_sum) return FQ_ERR_INVALID_ARG;
        if (p.n != 0 && (p.a == nullptr || p.b == nullptr)) return FQ_ERR_INVALID_ARG;
        if ((reinterpret_cast<uintptr_t>(p.a) | reinterpret_cast<uintptr_t>(p.b)) & 3u) return FQ_ERR_INVALID_ARG;
        if (p.n != 0 && ((reinterpret_cast<uintptr_t>(p.a) | reinterpret_cast<uintptr_t>(p.b) | reinterpret_cast<uintptr_t>(p.relu_out)) & 15u))
            return FQ_ERR_UNSUPPORTED;
    }
    if (nseg == 0) return FQ_OK;
    if (interval == nullptr || hist == nullptr) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    int i = 0;
    while (i < nseg) {
        PairTable tab;
        int k = 0;
        uint64_t chunks = 0;
        while (i < nseg && k < kPairChunk) {
            const fq_pair_seg& p = segs[i++];
            if (p.n == 0) continue;
            const uint64_t nvec = p.n >> 2;
            uint64_t nc = (nvec + kChunkVec - 1) / kChunkVec;
            if (nc == 0) nc = 1;                                     // fewer than four elements: the scalar tail of chunk 0
            if (chunks + nc > 0x7fffffffULL) { --i; break; }
            tab.a[k] = p.a; tab.b[k] = p.b; tab.r[k] = p.relu_out; tab.n[k] = p.n; tab.row_a[k] = p.row_a; tab.row_s[k] = p.row_sum;
            tab.chunk_begin[k] = (uint32_t)chunks;
            chunks += nc;
            ++k;
        }
        if (k == 0) {
            if (i < nseg && segs[i].n != 0) return FQ_ERR_INVALID_ARG;
            continue;
        }
        tab.nseg = k;
        for (int j = k; j <= kPairChunk; ++j) tab.chunk_begin[j] = (uint32_t)chunks;
        for (int j = k; j < kPairChunk; ++j) { tab.a[j] = tab.b[j] = nullptr; tab.r[j] = nullptr; tab.n[j] = 0; tab.row_a[j] = -1; tab.row_s[j] = 0; }
        const uint64_t slots = (uint64_t)kCUs * wg_per_cu(kWgPerCUHist);
        uint64_t per_wg = (chunks + slots - 1) / slots;
        if (per_wg < kMinChunksPerWg) per_wg = kMinChunksPerWg;
        tab.chunks_per_wg = (uint32_t)per_wg;
        tab.total_chunks = (uint32_t)chunks;
        hipLaunchKernelGGL(hist2048_pair_seg_kernel, dim3((uint32_t)((chunks + per_wg - 1) / per_wg)), dim3(kHistBlock), 0, st, tab, interval,
                           reinterpret_cast<unsigned long long*>(hist), hist_fast_quotient_enabled());
        FQ_LAUNCH_CHECK();
    }
    return FQ_OK;
}

extern "C" int fq_absmax_chan(const fq_chan_seg* segs, int nseg, float* max_inout, fq_stream_t stream) {
    using namespace fq;
    if (nseg == 0) return FQ_OK;
    if (max_inout == nullptr) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    return for_each_chan_chunk(segs, nseg, [&](const ChanTable& tab, uint32_t wgs) -> int {
        hipLaunchKernelGGL(absmax_chan_kernel, dim3(wgs), dim3(kBlock), 0, st, tab, max_inout);
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    });
}

extern "C" int fq_hist2048_chan(const fq_chan_seg* segs, int nseg, const float* interval, int64_t* hist, fq_stream_t stream) {
    using namespace fq;
    if (nseg == 0) return FQ_OK;
    if (interval == nullptr || hist == nullptr) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    // FQ_CHAN_OWN_FLUSH=0: every flush with atomics (A/B timing)
    const char* const own_env = getenv("FQ_CHAN_OWN_FLUSH");
    const bool own_ok = !(own_env && own_env[0] == '0');
    return for_each_chan_chunk(segs, nseg, [&](const ChanTable& tab, uint32_t wgs) -> int {
        hipLaunchKernelGGL(hist2048_chan_kernel, dim3(wgs), dim3(kHistBlock), 0, st, tab, interval,
                           reinterpret_cast<unsigned long long*>(hist), hist_fast_quotient_enabled());
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    }, own_ok && (reinterpret_cast<uintptr_t>(hist) & 15u) == 0);
}
