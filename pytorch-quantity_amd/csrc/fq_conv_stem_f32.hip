// fq_conv_stem_f32.hip -- the float stem convolution of the calibration forward (7x7, stride 2, 3 input channels -> <= 64
// output channels: ResNet-50/101's conv1) on the fp32 matrix cores, with the calibration's statistic in the epilogue.
//
// Companion of fq_conv1x1_f32.hip: after the 1x1 layers the stem is the most expensive single launch of the float forward
// (1.4 ms of library convolution + 0.35 ms of bias-add producer per 256 images, twice: pass 2 re-runs the prefix of the
// network).  Here it is an implicit GEMM  D[co][position] = sum_k W[co][k] * patch[k][position]  with k = (c, r, s) and
// the tap axis s padded from 7 to 8 (zero weights), so that K = 3 * 7 * 8 = 168 = 84 steps of v_mfma_f32_32x32x2_f32.
//
// A persistent workgroup (4 waves) keeps the whole weight matrix in LDS (168 x 64 floats, 43 KB) and walks over output
// tiles of 8 rows x 16 columns; wave w owns rows 2w, 2w+1, so the 32 columns of its MFMA tile are (dy, dx) = (j >> 4,
// j & 15) and it produces all 64 output channels for them (two MFMA tiles sharing the B operand).  The tile's input
// patch (3 x 21 x 38 floats, zero outside the image) is staged in LDS with even and odd columns in separate planes:
// for a stride-2 convolution all lanes of a k-pair (s = 2q + h, h = lane >> 5) then read consecutive words -- the
// address of every operand read is a per-lane base plus a compile-time immediate, no address arithmetic in the loop,
// no bank conflicts (plane pitch 20: dy moves the read by 16 banks).  Two patch buffers: the next tile's patch is
// fetched into registers before this tile's MFMAs and stored after them, one barrier per tile.  (4 x 32 tiles -- 128-byte
// instead of 64-byte store runs, 12.5 % of the columns of a 112-wide plane wasted -- were measured: 0.89 instead of 0.81 ms.)
//
// Numerics: an fmaf chain over k in the order (c, r, s) from 0, then one rounding for the bias; deterministic; not
// bit-identical to the library's convolution (summation order), like fq_conv1x1_f32.
#include "fq_common.h"
#include "fq_producer_stat.h"

namespace fq {
namespace {

constexpr int kT = 256;
typedef float f16v __attribute__((ext_vector_type(16)));

template <int CIN, int R, int S>
struct StemShape {
    static constexpr int STRIDE = 2;
    static constexpr int S8 = (S + 1) & ~1;                   // taps padded to an even count
    static constexpr int KP = CIN * R * S8;                   // padded reduction length
    static constexpr int STEPS = KP / 2;
    static constexpr int TH = 8, TW = 16;                     // output tile
    static constexpr int PH = STRIDE * (TH - 1) + R;          // 21 input rows
    static constexpr int PW = STRIDE * (TW - 1) + S8;         // 38 input columns (the padded tap reads one more)
    static constexpr int PC = 20;                             // words per (row, parity) plane: >= PW / 2, = 4 mod 8
    static constexpr int PATCH = CIN * PH * 2 * PC;           // floats per patch buffer
    static constexpr int NFILL = (CIN * PH * PW + kT - 1) / kT;
    static constexpr int COUT = 64;
    static constexpr int kFloats = KP * COUT + 2 * PATCH + COUT;
    static_assert(PW / 2 <= PC && (PC % 8) == 4, "patch plane pitch");
};

struct StemArgs {
    const float* x;           // [N][CIN][H][W]
    const float* wp;          // [KP][64]: W[co][c][r][s] at row (c * R + r) * S8 + s, zero for s >= S and co >= Cout
    const float* bias;        // [Cout] or null
    float* y;                 // [N][Cout][Hout][Wout]
    float* relu;              // or null
    int H, W, Hout, Wout, Cout, pad;
    unsigned tiles_x, tiles_y, tiles;      // per image: tiles_y x tiles_x; tiles = N * tiles_y * tiles_x
    int stream_stores;
    unsigned x_bytes, y_bytes;
};

struct NoStat {
    __device__ __forceinline__ void add(float) {}
};

template <int I>
struct Buf {
    static constexpr int value = I;
};

template <bool kRelu, bool kStream, bool kFull, typename Stat>
__device__ __forceinline__ void stem_epilogue(const f16v& acc0, const f16v& acc1, const StemArgs& a, Stat& stat, const float* s_bias,
                                              unsigned base4, unsigned plane4, unsigned h) {
    // buffer stores: per-lane byte offset of (pixel, channel 4 h) + the channel advance as the scalar offset: a value costs its
    // bias add, the ReLU select and the statistic (a vector instruction of any wave takes issue cycles from the matrix pipe)
    typedef float f4 __attribute__((ext_vector_type(4)));
    // (y == null with a ReLU copy: a descriptor of zero records drops every store of y in the address unit)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y ? a.y : a.relu, 0, a.y ? a.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(kRelu ? a.relu : a.y, 0, a.y_bytes, 0x00020000);
    constexpr int aux = kStream ? 2 : 0;                      // nt
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = *reinterpret_cast<const f4*>(s_bias + 32 * half + 8 * q + 4 * (int)h);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int dco = 32 * half + (e & 3) + 8 * (e >> 2);                    // + 4 h
            if (kFull || dco + 4 * (int)h < a.Cout) {
                const float val = stat_map(stat, (half ? acc1[e] : acc0[e]) + b4[e >> 2][e & 3]);
                const int row4 = (int)((unsigned)dco * plane4);                     // uniform
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yrs, (int)base4, row4, aux);
                if (kRelu)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, relu_like_torch(val)), rrs, (int)base4, row4, aux);
                stat.add(val);
            }
        }
    }
}

template <int CIN, int R, int S, typename Stat>
__device__ __forceinline__ void stem_tiles(const StemArgs& a, Stat& stat, float* smem) {
    typedef StemShape<CIN, R, S> G;
    float* Wl = smem;                                         // [KP][64]
    float* P = smem + G::KP * G::COUT;                        // [2][PATCH]
    float* s_bias = P + 2 * G::PATCH;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned j = lane & 31u, h = lane >> 5, dy = j >> 4, dx = j & 15u;

    for (unsigned i = tid; i < (unsigned)(G::KP * G::COUT / 4); i += kT)
        reinterpret_cast<float4*>(Wl)[i] = reinterpret_cast<const float4*>(a.wp)[i];
    if (tid < (unsigned)G::COUT) s_bias[tid] = (a.bias && (int)tid < a.Cout) ? a.bias[tid] : 0.0f;

    const unsigned per_img = a.tiles_x * a.tiles_y;
    // What a thread fetches is the same patch element for every tile: its (channel, row, column) and its place in the LDS
    // patch are worked out once, here; per tile and element there remain the two coordinates, the two border tests, one
    // multiply-add and two selects.  The image comes through a buffer descriptor: scalar offset = the image, vector offset
    // = the pixel (0 for a pixel outside the image: its value is replaced by the padding zero).
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    unsigned meta[G::NFILL];                                  // LDS word index | row << 12 | col << 17 | valid << 23
    unsigned chan[G::NFILL];                                  // c * H * W
#pragma unroll
    for (int f = 0; f < G::NFILL; ++f) {
        const unsigned e = tid + (unsigned)f * kT;
        const unsigned c = e / (G::PH * G::PW), r2 = e - c * (G::PH * G::PW), row = r2 / G::PW, col = r2 - row * G::PW;
        const unsigned valid = e < (unsigned)(CIN * G::PH * G::PW) ? 1u : 0u;
        meta[f] = (((c * G::PH + row) * 2 + (col & 1u)) * G::PC + (col >> 1)) | (row << 12) | (col << 17) | (valid << 23);
        chan[f] = c * (unsigned)(a.H * a.W);
    }
    static_assert(G::PATCH < 4096 && G::PH < 32 && G::PW < 64, "meta packing");
    float stage[G::NFILL];
    auto fetch = [&](unsigned t) {                            // the input patch of tile t -> registers (zero outside the image)
        const unsigned n = t / per_img, rem = t - n * per_img, ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int iy0 = (int)(ty * G::TH) * G::STRIDE - a.pad, ix0 = (int)(tx * G::TW) * G::STRIDE - a.pad;
        const int img4 = (int)(n * (unsigned)(CIN * a.H * a.W) * 4u);              // < 2^32 (host check)
#pragma unroll
        for (int f = 0; f < G::NFILL; ++f) {
            const int iy = iy0 + (int)((meta[f] >> 12) & 31u), ix = ix0 + (int)((meta[f] >> 17) & 63u);
            const bool in = (meta[f] >> 23) != 0u && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned off4 = in ? (chan[f] + (unsigned)iy * (unsigned)a.W + (unsigned)ix) * 4u : 0u;
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)off4, img4, 0));
            stage[f] = in ? v : 0.0f;
        }
    };
    auto stash = [&](auto which) {                            // registers -> LDS, even / odd columns in separate planes
        constexpr int buf = decltype(which)::value;
#pragma unroll
        for (int f = 0; f < G::NFILL; ++f)
            if ((meta[f] >> 23) != 0u) P[buf * G::PATCH + (meta[f] & 4095u)] = stage[f];
    };

    unsigned t = blockIdx.x;
    if (t < a.tiles) fetch(t);
    stash(Buf<0>{});
    __syncthreads();
    // operand read bases (compile-time offsets from here on; the two A reads get unrelated bases so that they stay two
    // ds_read_b32 with 16-bit immediates instead of one ds_read2_b32 whose 8-bit offsets need a vector add per step)
    unsigned wi0 = h * G::COUT + j, wi1 = wi0 + 32u;
    asm volatile("" : "+v"(wi0));
    asm volatile("" : "+v"(wi1));
    const float* const wrow0 = Wl + wi0;                      // + step * 128
    const float* const wrow1 = Wl + wi1;
    const float* const prow0 = P + ((4u * wave + 2u * dy) * 2u + h) * G::PC + dx;        // + buf * PATCH + ((c * PH + r) * 2) * PC + q
    const unsigned plane4 = (unsigned)(a.Hout * a.Wout) * 4u;
    auto tile = [&](auto which) {
        constexpr int buf = decltype(which)::value;
        const unsigned tn = t + gridDim.x;
        if (tn < a.tiles) fetch(tn);
        f16v acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.0f; acc1[e] = 0.0f; }
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int q = 0; q < G::S8 / 2; ++q) {
                    const int step = (c * R + r) * (G::S8 / 2) + q;
                    float b = prow0[buf * G::PATCH + ((c * G::PH + r) * 2) * G::PC + q];
                    // taps 2 q + h >= S are padding: their weights are zero, but the patch word there is a real neighbouring
                    // pixel, and 0 * Inf (or NaN) must not reach an output whose own window is finite
                    if (2 * q + 1 >= S) b = (2 * q >= S || h) ? 0.0f : b;
                    const float a0 = wrow0[step * 2 * G::COUT], a1 = wrow1[step * 2 * G::COUT];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
                }
        // epilogue: D[i][j]: j = lane & 31 -> (dy, dx), i = (e & 3) + 8 (e >> 2) + 4 h -> output channel (+ 32 for acc1);
        // ReLU copy / store flavour / "all 64 channels exist" are uniform: compile-time inside stem_epilogue
        {
            const unsigned n = t / per_img, rem = t - n * per_img, ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
            const int oy = (int)(ty * G::TH + 2u * wave + dy), ox = (int)(tx * G::TW + dx);
            if (oy < a.Hout && ox < a.Wout) {
                const unsigned base4 = (n * (unsigned)a.Cout + 4u * h) * plane4 + (unsigned)(oy * a.Wout + ox) * 4u;   // < 2^32 (host)
                const bool full = a.Cout == G::COUT;
#define FQ_STEM_EPI(R, S, F) stem_epilogue<R, S, F>(acc0, acc1, a, stat, s_bias, base4, plane4, h)
                if (a.relu) {
                    if (a.stream_stores) { if (full) FQ_STEM_EPI(true, true, true); else FQ_STEM_EPI(true, true, false); }
                    else { if (full) FQ_STEM_EPI(true, false, true); else FQ_STEM_EPI(true, false, false); }
                } else {
                    if (a.stream_stores) { if (full) FQ_STEM_EPI(false, true, true); else FQ_STEM_EPI(false, true, false); }
                    else { if (full) FQ_STEM_EPI(false, false, true); else FQ_STEM_EPI(false, false, false); }
                }
#undef FQ_STEM_EPI
            }
        }
        if (tn < a.tiles) stash(Buf<buf ^ 1>{});              // that buffer was last read one tile ago: every wave is past it
        __syncthreads();
        t = tn;
    };
    while (t < a.tiles) {
        tile(Buf<0>{});
        if (t >= a.tiles) break;
        tile(Buf<1>{});
    }
}

template <int CIN, int R, int S>
__global__ __launch_bounds__(kT) void conv_stem_f32_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    NoStat st;
    stem_tiles<CIN, R, S>(a, st, smem);
}

template <int CIN, int R, int S>
__global__ __launch_bounds__(kT) void conv_stem_f32_qd_kernel(const StemArgs a, const QdStat qd) {   // TestConv's stem: QuanDequan on the way out
    extern __shared__ __attribute__((aligned(16))) float smem[];
    QdStat st = qd;
    stem_tiles<CIN, R, S>(a, st, smem);
}

template <int CIN, int R, int S>
__global__ __launch_bounds__(kT) void conv_stem_f32_absmax_kernel(const StemArgs a, unsigned int* __restrict__ max_bits) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    MaxStat st;
    stem_tiles<CIN, R, S>(a, st, smem);
    publish_max<kT>(st.m, max_bits);
}

template <int CIN, int R, int S>
__global__ __launch_bounds__(kT) void conv_stem_f32_hist_kernel(const StemArgs a, const float* __restrict__ interval,
                                                                unsigned long long* __restrict__ hist_row, const int allow_fast) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ unsigned int s_bins[FQ_BINS + kWave];
    for (int b = threadIdx.x; b < FQ_BINS + kWave; b += kT) s_bins[b] = 0u;
    __syncthreads();
    const float iv = *interval;
    unsigned int* park = s_bins + FQ_BINS + (threadIdx.x & (kWave - 1));
    if (allow_fast && fast_quotient_ok(iv)) {
        HistStat<true> st{s_bins, park, iv, 1.0f / iv};
        stem_tiles<CIN, R, S>(a, st, smem);
    } else {
        HistStat<false> st{s_bins, park, iv, 1.0f / iv};
        stem_tiles<CIN, R, S>(a, st, smem);
    }
    hist_flush<kT>(s_bins, hist_row);
}

template <typename K>
int resident_per_cu(K kernel, size_t dyn) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kT, dyn) != hipSuccess || n < 1) n = 1;
    return n;
}

template <int CIN, int R, int S>
int launch_stem(StemArgs a, float* max_inout, const float* interval, int64_t* hist_row, int fast, const QdStat* qd, hipStream_t st) {
    typedef StemShape<CIN, R, S> G;
    const size_t dyn = (size_t)G::kFloats * sizeof(float);
    // persistent: every workgroup loads the 43 KB weight matrix once and walks over tiles
    if (qd) {
        auto k = conv_stem_f32_qd_kernel<CIN, R, S>;
        static bool lds_ok[kMaxDevices] = {};
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), (int)dyn, lds_ok)) return hip_fail(hipErrorInvalidValue);
        static const int per_cu = resident_per_cu(k, dyn);
        unsigned grid = (unsigned)(kCUs * per_cu);
        if (grid > a.tiles) grid = a.tiles;
        hipLaunchKernelGGL(k, dim3(grid), dim3(kT), dyn, st, a, *qd);
    } else if (hist_row) {
        auto k = conv_stem_f32_hist_kernel<CIN, R, S>;
        static bool lds_ok[kMaxDevices] = {};
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), (int)dyn, lds_ok)) return hip_fail(hipErrorInvalidValue);
        static const int per_cu = resident_per_cu(k, dyn);
        unsigned grid = (unsigned)(kCUs * per_cu);
        if (grid > a.tiles) grid = a.tiles;
        hipLaunchKernelGGL(k, dim3(grid), dim3(kT), dyn, st, a, interval, reinterpret_cast<unsigned long long*>(hist_row), fast);
    } else if (max_inout) {
        auto k = conv_stem_f32_absmax_kernel<CIN, R, S>;
        static bool lds_ok[kMaxDevices] = {};
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), (int)dyn, lds_ok)) return hip_fail(hipErrorInvalidValue);
        static const int per_cu = resident_per_cu(k, dyn);
        unsigned grid = (unsigned)(kCUs * per_cu);
        if (grid > a.tiles) grid = a.tiles;
        hipLaunchKernelGGL(k, dim3(grid), dim3(kT), dyn, st, a, reinterpret_cast<unsigned int*>(max_inout));
    } else {
        auto k = conv_stem_f32_kernel<CIN, R, S>;
        static bool lds_ok[kMaxDevices] = {};
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), (int)dyn, lds_ok)) return hip_fail(hipErrorInvalidValue);
        static const int per_cu = resident_per_cu(k, dyn);
        unsigned grid = (unsigned)(kCUs * per_cu);
        if (grid > a.tiles) grid = a.tiles;
        hipLaunchKernelGGL(k, dim3(grid), dim3(kT), dyn, st, a);
    }
    return FQ_OK;
}

}  // namespace
}  // namespace fq

using namespace fq;

extern "C" int fq_conv_stem_f32_packed_rows(int Cin, int R, int S) {
    if (Cin == 3 && R == 7 && S == 7) return StemShape<3, 7, 7>::KP;
    return FQ_ERR_UNSUPPORTED;
}

static int stem_launch(const float* x, const float* wp, const float* bias, float* y, float* relu_out, int N, int Cin,
                       int H, int W, int Cout, int R, int S, int stride, int pad, float* max_inout,
                       const float* interval, int64_t* hist_row, const QdStat* qd, fq_stream_t stream) {
    if (N < 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0) return FQ_ERR_INVALID_ARG;
    if (max_inout && hist_row) return FQ_ERR_INVALID_ARG;
    if (hist_row && !interval) return FQ_ERR_INVALID_ARG;
    if (!(Cin == 3 && R == 7 && S == 7 && stride == 2) || Cout > 64) return FQ_ERR_UNSUPPORTED;
    if (H + 2 * pad < R || W + 2 * pad < S) return FQ_ERR_INVALID_ARG;
    if (N == 0) return FQ_OK;
    if (!x || !wp || (!y && (!relu_out || qd))) return FQ_ERR_INVALID_ARG;                   // (y may be null when only its ReLU is wanted)
    if (reinterpret_cast<uintptr_t>(wp) & 15u) return FQ_ERR_INVALID_ARG;
    StemArgs a;
    a.x = x; a.wp = wp; a.bias = bias; a.y = y; a.relu = relu_out;
    a.H = H; a.W = W; a.Cout = Cout; a.pad = pad;
    a.Hout = (H + 2 * pad - R) / stride + 1;
    a.Wout = (W + 2 * pad - S) / stride + 1;
    a.tiles_x = (unsigned)((a.Wout + 15) / 16);
    a.tiles_y = (unsigned)((a.Hout + 7) / 8);
    const size_t tiles = (size_t)N * a.tiles_x * a.tiles_y;
    if (tiles >= 0x7fffffffULL) return FQ_ERR_UNSUPPORTED;
    a.tiles = (unsigned)tiles;
    const size_t out_elems = (size_t)N * Cout * a.Hout * a.Wout, in_elems = (size_t)N * Cin * H * W;
    if (out_elems >= (1ULL << 30) || in_elems >= (1ULL << 30)) return FQ_ERR_UNSUPPORTED;          // 32-bit byte offsets
    a.x_bytes = (unsigned)(in_elems * 4);
    a.y_bytes = (unsigned)(out_elems * 4);
    a.stream_stores = out_elems * (relu_out && y ? 8 : 4) > ((size_t)256 << 20);
    static const int fast = [] { const char* e = getenv("FQ_HIST_IEEE_DIV"); return (e && e[0] && e[0] != '0') ? 0 : 1; }();
    const int rc = launch_stem<3, 7, 7>(a, max_inout, interval, hist_row, fast, qd, as_stream(stream));
    if (rc != FQ_OK) return rc;
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_conv_stem_f32(const float* x, const float* wp, const float* bias, float* y, float* relu_out, int N, int Cin,
                                int H, int W, int Cout, int R, int S, int stride, int pad, float* max_inout,
                                const float* interval, int64_t* hist_row, fq_stream_t stream) {
    return stem_launch(x, wp, bias, y, relu_out, N, Cin, H, W, Cout, R, S, stride, pad, max_inout, interval, hist_row, nullptr, stream);
}

// TestConv.forward of the stem in one kernel: y = QuanDequan(conv(x) + bias, bit) (fq_conv1x1_qd_f32's contract)
extern "C" int fq_conv_stem_qd_f32(const float* x, const float* wp, const float* bias, float* y, int N, int Cin, int H, int W,
                                   int Cout, int R, int S, int stride, int pad, int bit, int bitwidth, fq_stream_t stream) {
    if ((bitwidth != 8 && bitwidth != 16) || bit < -120 || bit > 120) return FQ_ERR_INVALID_ARG;
    QdStat qd;
    qd.scale = ldexpf(1.0f, bit); qd.inv = ldexpf(1.0f, -bit);
    qd.lo = bitwidth == 8 ? -128.0f : -32768.0f; qd.hi = bitwidth == 8 ? 127.0f : 32767.0f;
    return stem_launch(x, wp, bias, y, nullptr, N, Cin, H, W, Cout, R, S, stride, pad, nullptr, nullptr, nullptr, &qd, stream);
}
