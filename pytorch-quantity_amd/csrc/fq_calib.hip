// fq_calib.hip -- calibration statistics for gfx950: segmented abs-max and 2048-bin histogram.
//
// Replaces the reference's per-element Python loop and per-call multiprocessing.Pool
// (quantity/common/quantity/distribution_collector.py:70-78 and :80-142).  One launch covers every
// hooked tensor of a forward pass ("segments"): blockIdx -> (segment, tile) through a prefix table
// that travels in the kernel-argument block, so there is no per-tensor launch and no table memcpy.
//
// Both kernels are HBM-bound streaming reads (4 B per element, algorithmic bytes = 4 * elements):
//   * 16-byte loads, 4 in flight per lane; tiles sized per kernel (abs-max: 256 threads, ~4
//     workgroups per CU; histogram: 256 threads, ~8 per CU -- both measured optima).
//   * histogram bins live in LDS (8 KB per workgroup shared by its 4 waves, ds_add_u32); a
//     workgroup flushes only its non-zero bins to the int64 global rows with 64-bit atomics
//     (T*2048 counters, L2-resident).  Bin = trunc(|x| / interval) through a 3-instruction quotient
//     proven identical to the IEEE divide (profiles/r01_verify_fastdiv.log).
//   * abs-max keeps a per-lane running max, reduces across the wave with DPP shuffles, across waves
//     through LDS, and publishes with one 32-bit atomic max on the (non-negative) float's bits.
#include <cstdlib>

#include "fq_common.h"

namespace fq {

thread_local int g_last_hip_error = 0;

constexpr int kSegChunk = 96;          // segments per launch (kernarg block stays < 4 KB)
constexpr int kBlock = 256;            // abs-max: 4 waves per workgroup, ~4 workgroups per CU
constexpr int kHistBlock = 256;        // histogram: 4 waves per LDS histogram, ~8 workgroups per CU (measured optimum;
                                       // 1024-thread workgroups at 2 per CU were 10 % slower)
constexpr uint32_t kMinTile = 4096;    // elements
constexpr uint32_t kMaxTile = 1u << 22;
constexpr int kTilesPerCUAbsmax = 4;
constexpr int kTilesPerCUHist = 8;
constexpr int kHistFastQuotientDefault = 1;   // exhaustive proof: profiles/r01_verify_fastdiv.log

struct SegTable {
    const float* ptr[kSegChunk];
    uint64_t n[kSegChunk];
    uint32_t tile_begin[kSegChunk + 1];   // exclusive prefix of tiles per segment
    int32_t row[kSegChunk];
    uint32_t tile_elems;                  // multiple of 4
    int32_t nseg;
};

// blockIdx.x -> segment index (largest s with tile_begin[s] <= b). Uniform per workgroup.
__device__ __forceinline__ int find_seg(const SegTable& t, uint32_t b) {
    int lo = 0, hi = t.nseg - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (t.tile_begin[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

struct TileView {
    const float* p;      // first element of this tile
    uint64_t cnt;        // elements in this tile
    int row;
};

__device__ __forceinline__ TileView tile_of(const SegTable& t) {
    const uint32_t b = blockIdx.x;
    const int s = find_seg(t, b);
    const uint64_t off = (uint64_t)(b - t.tile_begin[s]) * t.tile_elems;
    const uint64_t rem = t.n[s] - off;
    TileView v;
    v.p = t.ptr[s] + off;
    v.cnt = rem < t.tile_elems ? rem : t.tile_elems;
    v.row = t.row[s];
    return v;
}

// Visit every element of the tile: scalar head until 16-B aligned, float4 body with 4 loads in
// flight per lane, scalar tail.
template <int kThreads, bool kFenceLoads, typename F>
__device__ __forceinline__ void for_each_in_tile(const TileView& tv, F&& f) {
    const int tid = threadIdx.x;
    const float* p = tv.p;
    uint64_t cnt = tv.cnt;
    const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(p) & 15u) >> 2);
    uint32_t head = mis ? 4u - mis : 0u;
    if (head > cnt) head = (uint32_t)cnt;
    if ((uint32_t)tid < head) f(p[tid]);
    p += head;
    cnt -= head;
    const float4* __restrict__ v4 = reinterpret_cast<const float4*>(p);
    const uint64_t nvec = cnt >> 2;
    uint64_t i = tid;
    for (; i + 3 * kThreads < nvec; i += 4 * kThreads) {
        const float4 a = v4[i];
        const float4 b = v4[i + kThreads];
        const float4 c = v4[i + 2 * kThreads];
        const float4 d = v4[i + 3 * kThreads];
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
    for (; i < nvec; i += kThreads) {
        const float4 a = v4[i];
        f(a.x); f(a.y); f(a.z); f(a.w);
    }
    const uint32_t tail = (uint32_t)(cnt & 3u);
    if ((uint32_t)tid < tail) f(p[(nvec << 2) + tid]);
}

// ---------------------------------------------------------------------------------------------
// abs-max  (distribution_collector.py:70-78)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void absmax_seg_kernel(const SegTable tab, float* __restrict__ max_inout) {
    __shared__ float s_wave[kBlock / kWave];
    const TileView tv = tile_of(tab);
    float m = 0.0f;
    for_each_in_tile<kBlock, false>(tv, [&](float v) { m = fmaxf(m, fabsf(v)); });   // fmaxf drops NaN
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) s_wave[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBlock / kWave; ++w) m = fmaxf(m, s_wave[w]);
        // m >= 0, so the IEEE bit pattern orders like an unsigned integer
        atomicMax(reinterpret_cast<unsigned int*>(max_inout + tv.row), __float_as_uint(m));
    }
}

// ---------------------------------------------------------------------------------------------
// 2048-bin histogram of |x|, x != 0  (distribution_collector.py:127-135)
// ---------------------------------------------------------------------------------------------
// Bin of one element.  kFast = false: the IEEE divide sequence.  kFast = true: the 3-instruction
// quotient  q0 = a*y, r = fma(-q0, iv, a), q = fma(r, y, q0)  with y = RN(1/iv); it equals the
// correctly rounded a/iv for every fp32 significand pair (checked exhaustively on the GPU,
// scripts/verify_fastdiv.hip, result under profiles/), and overflow / inf / nan fall through to the
// same "last bin" as the IEEE path because  q < 2048  is false for inf and nan.
template <bool kFast>
__device__ __forceinline__ int bin_of(float v, float iv, float y) {
    const float a = fabsf(v);
    float q;
    if (kFast) {
        const float q0 = a * y;
        const float r = __builtin_fmaf(-q0, iv, a);
        q = __builtin_fmaf(r, y, q0);
    } else {
        q = a / iv;                                   // v_div_scale / v_rcp / fma x4 / v_div_fmas / v_div_fixup
    }
    return (q < 2048.0f) ? (int)q : (FQ_BINS - 1);   // >= 2048, inf, nan -> last bin
}

template <bool kFast>
__device__ __forceinline__ void hist_tile(const TileView& tv, float iv, unsigned int* s_bins) {
    const float y = 1.0f / iv;                        // IEEE, once per lane
    // branch-free: lanes holding an exact zero add into a private scratch slot (2048 + lane)
    unsigned int* park = s_bins + FQ_BINS + (threadIdx.x & (kWave - 1));
    for_each_in_tile<kHistBlock, true>(tv, [&](float v) {
        unsigned int* slot = (v != 0.0f) ? (s_bins + bin_of<kFast>(v, iv, y)) : park;
        atomicAdd(slot, 1u);                          // ds_add_u32
    });
}

__global__ __launch_bounds__(kHistBlock) void hist2048_seg_kernel(const SegTable tab,
                                                              const float* __restrict__ interval,
                                                              unsigned long long* __restrict__ hist,
                                                              const int allow_fast) {
    __shared__ unsigned int s_bins[FQ_BINS + kWave];
    for (int b = threadIdx.x; b < FQ_BINS + kWave; b += kHistBlock) s_bins[b] = 0u;
    const TileView tv = tile_of(tab);
    const float iv = interval[tv.row];
    __syncthreads();
    // The exhaustive proof of the fast quotient covers every significand pair but assumes that no
    // intermediate under/overflows.  For 2^-60 <= iv <= 2^60 that holds for every element that can
    // land above bin 0 (|x| >= iv >= 2^-60 keeps the fma residual normal; quotients that overflow
    // become inf/nan and fall into the last bin exactly like the IEEE path).  Calibration intervals
    // are max/2048 + 1e-12, far inside that range; anything else takes the IEEE divide.  Uniform per
    // workgroup.
    const unsigned int ivb = __float_as_uint(iv);
    const bool fast = allow_fast && ivb >= 0x21800000u && ivb <= 0x5d800000u;
    if (fast) hist_tile<true>(tv, iv, s_bins); else hist_tile<false>(tv, iv, s_bins);
    __syncthreads();
    unsigned long long* __restrict__ dst = hist + (size_t)tv.row * FQ_BINS;
    for (int b = threadIdx.x; b < FQ_BINS; b += kHistBlock) {
        const unsigned int c = s_bins[b];
        if (c) atomicAdd(dst + b, (unsigned long long)c);
    }
}

// ---------------------------------------------------------------------------------------------
// per-channel rows: one histogram row per (tensor, channel) of an NCHW activation
// ---------------------------------------------------------------------------------------------
// A row is no longer one contiguous run: channel c of tensor [N][C][HW] is the N planes at (n*C + c)*HW.  A
// workgroup owns ONE channel and a group of images, so its LDS histogram (or running max) belongs to a single
// row and is flushed once; blockIdx -> (tensor, channel, image group) through a kernarg prefix table as above.
// Each wave sweeps one plane at a time (16-byte loads, four in flight per lane, when planes are 16-byte aligned
// and at least 256 elements; 4-byte loads otherwise): no per-element index arithmetic, and the tensor is read
// in place -- no channel-major copy.
constexpr int kChanChunk = 96;            // kernarg block 3.4 KB: ResNet-50 (71 tensors) in one launch
constexpr uint32_t kChanElemsPerWg = 262144;   // per workgroup: enough to amortise zeroing + flushing 2048 bins (32 K: 3.8 TB/s)

struct ChanTable {
    const float* ptr[kChanChunk];
    uint32_t N[kChanChunk], C[kChanChunk], HW[kChanChunk];
    uint32_t groups[kChanChunk], nb[kChanChunk];          // image groups per channel, images per group
    int32_t row0[kChanChunk];
    uint32_t wg_begin[kChanChunk + 1];
    int32_t nseg;
};

struct ChanView { const float* base; uint32_t C, HW, n0, n1; int row; };

__device__ __forceinline__ ChanView chan_of(const ChanTable& t) {
    const uint32_t b = blockIdx.x;
    int lo = 0, hi = t.nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.wg_begin[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const uint32_t local = b - t.wg_begin[lo];
    const uint32_t c = local / t.groups[lo], g = local - c * t.groups[lo];
    ChanView v;
    v.C = t.C[lo]; v.HW = t.HW[lo];
    v.n0 = g * t.nb[lo];
    v.n1 = v.n0 + t.nb[lo] < t.N[lo] ? v.n0 + t.nb[lo] : t.N[lo];
    v.base = t.ptr[lo] + (size_t)c * v.HW;
    v.row = t.row0[lo] + (int)c;
    return v;
}

// f(value) for every element of this workgroup's planes: one plane per wave at a time (a plane of the 56x56 /
// 28x28 stages is too short for the 256-thread tile loader to keep four loads per lane in flight; a wave does)
template <int kThreads, bool kFenceLoads, typename F>
__device__ __forceinline__ void for_each_in_channel(const ChanView& cv, F&& f) {
    const size_t plane_stride = (size_t)cv.C * cv.HW;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    // 16-byte loads when every plane starts on a 16-byte boundary
    const bool vec = (cv.HW & 3u) == 0 && (reinterpret_cast<uintptr_t>(cv.base) & 15u) == 0 && cv.HW >= 256;
    for (uint32_t n = cv.n0 + wave; n < cv.n1; n += kThreads / kWave) {
        const float* __restrict__ p = cv.base + (size_t)n * plane_stride;
        if (vec) {
            const float4* __restrict__ v4 = reinterpret_cast<const float4*>(p);
            const uint32_t nvec = cv.HW >> 2;
            uint32_t i = lane;
            for (; i + 3 * kWave < nvec; i += 4 * kWave) {
                const float4 a = v4[i];
                const float4 b = v4[i + kWave];
                const float4 c = v4[i + 2 * kWave];
                const float4 d = v4[i + 3 * kWave];
                if (kFenceLoads) __builtin_amdgcn_sched_barrier(0);       // see for_each_in_tile
                f(a.x); f(a.y); f(a.z); f(a.w);
                f(b.x); f(b.y); f(b.z); f(b.w);
                f(c.x); f(c.y); f(c.z); f(c.w);
                f(d.x); f(d.y); f(d.z); f(d.w);
            }
            for (; i < nvec; i += kWave) {
                const float4 a = v4[i];
                f(a.x); f(a.y); f(a.z); f(a.w);
            }
        } else {
            for (uint32_t e = lane; e < cv.HW; e += kWave) f(p[e]);
        }
    }
}

__global__ __launch_bounds__(kBlock) void absmax_chan_kernel(const ChanTable tab, float* __restrict__ max_inout) {
    __shared__ float s_wave[kBlock / kWave];
    const ChanView cv = chan_of(tab);
    float m = 0.0f;
    for_each_in_channel<kBlock, false>(cv, [&](float v) { m = fmaxf(m, fabsf(v)); });
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
}

template <bool kFast>
__device__ __forceinline__ void hist_channel(const ChanView& cv, float iv, unsigned int* s_bins) {
    const float y = 1.0f / iv;
    unsigned int* park = s_bins + FQ_BINS + (threadIdx.x & (kWave - 1));
    for_each_in_channel<kHistBlock, true>(cv, [&](float v) {
        unsigned int* slot = (v != 0.0f) ? (s_bins + bin_of<kFast>(v, iv, y)) : park;
        atomicAdd(slot, 1u);
    });
}

__global__ __launch_bounds__(kHistBlock) void hist2048_chan_kernel(const ChanTable tab, const float* __restrict__ interval,
                                                               unsigned long long* __restrict__ hist, const int allow_fast) {
    __shared__ unsigned int s_bins[FQ_BINS + kWave];
    for (int b = threadIdx.x; b < FQ_BINS + kWave; b += kHistBlock) s_bins[b] = 0u;
    const ChanView cv = chan_of(tab);
    const float iv = interval[cv.row];
    __syncthreads();
    const unsigned int ivb = __float_as_uint(iv);
    const bool fast = allow_fast && ivb >= 0x21800000u && ivb <= 0x5d800000u;      // see hist2048_seg_kernel
    if (fast) hist_channel<true>(cv, iv, s_bins); else hist_channel<false>(cv, iv, s_bins);
    __syncthreads();
    unsigned long long* __restrict__ dst = hist + (size_t)cv.row * FQ_BINS;
    for (int b = threadIdx.x; b < FQ_BINS; b += kHistBlock) {
        const unsigned int c = s_bins[b];
        if (c) atomicAdd(dst + b, (unsigned long long)c);
    }
}

template <typename Launch>
static int for_each_chan_chunk(const fq_chan_seg* segs, int nseg, Launch&& launch) {
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
            uint32_t nb = (uint32_t)(kChanElemsPerWg / (uint64_t)s.HW);
            if (nb < kBlock / kWave) nb = kBlock / kWave;                  // at least one plane per wave
            if (nb > (uint32_t)s.N) nb = (uint32_t)s.N;
            const uint32_t groups = ((uint32_t)s.N + nb - 1) / nb;
            const uint64_t n_wg = (uint64_t)groups * (uint64_t)s.C;
            if (wgs + n_wg > 0x7fffffffULL) { --i; break; }
            tab.ptr[k] = s.ptr; tab.N[k] = (uint32_t)s.N; tab.C[k] = (uint32_t)s.C; tab.HW[k] = (uint32_t)s.HW;
            tab.groups[k] = groups; tab.nb[k] = nb; tab.row0[k] = s.row0;
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
            tab.ptr[j] = nullptr; tab.N[j] = 0; tab.C[j] = 1; tab.HW[j] = 1; tab.groups[j] = 1; tab.nb[j] = 1; tab.row0[j] = 0;
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

static uint32_t pick_tile_elems(const fq_seg* segs, int nseg, int default_per_cu) {
    uint64_t total = 0;
    for (int i = 0; i < nseg; ++i) total += segs[i].n;
    // aim for ~kTilesPerCU workgroups per CU over the whole call (FQ_TILES_PER_CU overrides: tuning knob)
    static const int env_per_cu = [] {
        const char* e = getenv("FQ_TILES_PER_CU");
        return e ? atoi(e) : 0;
    }();
    const int per_cu = env_per_cu > 0 ? env_per_cu : default_per_cu;
    uint64_t want = total / (uint64_t)(kCUs * per_cu);
    uint32_t tile = kMinTile;
    while (tile < want && tile < kMaxTile) tile <<= 1;
    return tile;
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
    const uint32_t tile = pick_tile_elems(segs, nseg, per_cu);
    int i = 0;
    while (i < nseg) {
        SegTable tab;
        tab.tile_elems = tile;
        int k = 0;
        uint64_t tiles = 0;
        while (i < nseg && k < kSegChunk) {
            const fq_seg& s = segs[i++];
            if (s.n == 0) continue;
            const uint64_t nt = (s.n + tile - 1) / tile;
            if (tiles + nt > 0x7fffffffULL) { --i; break; }     // grid limit: start a new launch
            tab.ptr[k] = s.ptr;
            tab.n[k] = s.n;
            tab.row[k] = s.row;
            tab.tile_begin[k] = (uint32_t)tiles;
            tiles += nt;
            ++k;
        }
        if (k == 0) {
            if (i < nseg && segs[i].n != 0) return FQ_ERR_INVALID_ARG;   // single segment over the grid limit
            continue;
        }
        tab.tile_begin[k] = (uint32_t)tiles;
        tab.nseg = k;
        for (int j = k + 1; j <= kSegChunk; ++j) tab.tile_begin[j] = (uint32_t)tiles;
        for (int j = k; j < kSegChunk; ++j) { tab.ptr[j] = nullptr; tab.n[j] = 0; tab.row[j] = 0; }
        int rc = launch(tab, (uint32_t)tiles);
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
    return for_each_chunk(segs, nseg, kTilesPerCUAbsmax, [&](const SegTable& tab, uint32_t tiles) -> int {
        hipLaunchKernelGGL(absmax_seg_kernel, dim3(tiles), dim3(kBlock), 0, st, tab, max_inout);
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
    return for_each_chunk(segs, nseg, kTilesPerCUHist, [&](const SegTable& tab, uint32_t tiles) -> int {
        hipLaunchKernelGGL(hist2048_seg_kernel, dim3(tiles), dim3(kHistBlock), 0, st, tab, interval,
                           reinterpret_cast<unsigned long long*>(hist), hist_fast_quotient_enabled());
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    });
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
    return for_each_chan_chunk(segs, nseg, [&](const ChanTable& tab, uint32_t wgs) -> int {
        hipLaunchKernelGGL(hist2048_chan_kernel, dim3(wgs), dim3(kHistBlock), 0, st, tab, interval,
                           reinterpret_cast<unsigned long long*>(hist), hist_fast_quotient_enabled());
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    });
}
