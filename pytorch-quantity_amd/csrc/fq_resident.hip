// fq_resident.hip -- the integer-simulation model with activations kept RESIDENT as integers between
// layers (int8 / int16 NHWC in HBM) instead of crossing every nn.Module boundary as fp32 NCHW.
//
// The reference's ReconModel hands fp32 tensors from module to module (new_quantity_op.py:124-133
// NewConv2d.forward, :171-174 NewAdd.forward): DeQuantity writes 4 B per element, nn.ReLU reads and
// writes it, the next layer's Quantity reads it again and recovers -- exactly -- the integer the
// previous tail had in registers.  Every value on that path is an integer times a power of two:
//
//   conv / linear output   y = q * 2^-ob,          q in [-128, 127]      -> int8, grid ob
//   residual add           s = clamp(x + y)        (fp32 add of two such values: exact, no rounding,
//                                                   while 8 + grid <= 24 bits)
//                                                  -> S = s * 2^g integer, g = max(0, gx, gy),
//                                                     |S| <= 128 * 2^g: int16 while g <= 8
//   ReLU                   max(., 0)               commutes with the positive scale
//   next Quantity(ib)      clamp(rint(s * 2^ib))   a function of the exact s
//
// so the same fp32 arithmetic can be carried out on the integers and the stored bytes drop from
// 18-29 per activation to 2-6.  The kernels below evaluate the reference's fp32 expressions on the
// de-scaled integers (bit-identical by construction); fq_conv2d_i8_resident (fq_conv_i8.hip) writes
// the int8 form straight from the MFMA epilogue.
#include "fq_resident.h"

namespace fq {

constexpr int kResBlock = 256;

// NewAdd on resident operands (add_resident_16 in fq_resident.h).  All operands share one flat NHWC layout
// [N][HW][Cpad]; 16 elements per thread.
template <typename TX, typename TY>
__global__ __launch_bounds__(kResBlock) void add_resident_kernel(const TX* __restrict__ x, const TY* __restrict__ y,
                                                                 int16_t* __restrict__ wide, int8_t* __restrict__ narrow,
                                                                 size_t n16, const AddResParams p) {
    size_t i = (size_t)blockIdx.x * kResBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kResBlock;
    for (; i < n16; i += stride) {
        Vec16<TX> vx; Vec16<TY> vy;
        vx.load(x + i * 16);
        vy.load(y + i * 16);
        add_resident_16(vx, vy, wide ? wide + i * 16 : nullptr, narrow ? narrow + i * 16 : nullptr, p);
    }
}

// Leaving the resident domain (the consumer is a module this library does not own: pooling, View, a
// user op): integer NHWC [N][HW][Cpad] -> fp32 NCHW [N][C][HW], value = q * 2^-g (DeQuantity, exact).
// 64 pixels x 64 channels per workgroup through an LDS tile; coalesced on both sides.
template <typename T>
__global__ __launch_bounds__(kResBlock) void dequant_nhwc_to_nchw_kernel(const T* __restrict__ q, float* __restrict__ y, int C,
                                                                         int HW, int Cpad, float scale) {
    __shared__ float tile[64][65];                        // [c][hw]
    const int n = blockIdx.z, hw0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    {
        const int hw = hw0 + (tid >> 2), cb = c0 + 16 * (tid & 3);
        if (hw < HW && cb < Cpad) {
            Vec16<T> v;
            v.load(q + ((size_t)n * HW + hw) * Cpad + cb);
#pragma unroll
            for (int e = 0; e < 16; ++e) tile[16 * (tid & 3) + e][tid >> 2] = v.get(e) * scale;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int idx = tid + kResBlock * j;
        const int c = idx >> 6, hw = idx & 63;
        if (c0 + c < C && hw0 + hw < HW) y[((size_t)n * C + c0 + c) * HW + hw0 + hw] = tile[c][hw];
    }
}

inline unsigned res_grid(size_t items) {
    size_t g = (items + kResBlock - 1) / kResBlock;
    const size_t cap = (size_t)kCUs * 16;
    if (g > cap) g = cap;
    return (unsigned)(g ? g : 1);
}

template <typename TX, typename TY>
static void launch_add(hipStream_t st, const void* x, const void* y, int16_t* wide, int8_t* narrow, size_t n16,
                       const AddResParams& p) {
    hipLaunchKernelGGL((add_resident_kernel<TX, TY>), dim3(res_grid(n16)), dim3(kResBlock), 0, st, static_cast<const TX*>(x),
                       static_cast<const TY*>(y), wide, narrow, n16, p);
}

}  // namespace fq

using namespace fq;

extern "C" int fq_add_resident(const void* x, int x_bytes, int gx, const void* y, int y_bytes, int gy, int16_t* wide,
                               int g_wide, int8_t* narrow, int ib, int relu, size_t n, fq_stream_t stream) {
    if ((x_bytes != 1 && x_bytes != 2) || (y_bytes != 1 && y_bytes != 2)) return FQ_ERR_INVALID_ARG;
    if (n & 15u) return FQ_ERR_INVALID_ARG;               // NHWC rows are padded to 16 channels
    if (n == 0) return FQ_OK;
    if (!x || !y || (!wide && !narrow)) return FQ_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(wide) |
         reinterpret_cast<uintptr_t>(narrow)) & 15u)
        return FQ_ERR_INVALID_ARG;
    AddResParams p;
    const int rc = make_add_params(gx, gy, g_wide, wide != nullptr, ib, relu, &p);
    if (rc != FQ_OK) return rc;
    hipStream_t st = as_stream(stream);
    const size_t n16 = n >> 4;
    if (x_bytes == 1 && y_bytes == 1) launch_add<int8_t, int8_t>(st, x, y, wide, narrow, n16, p);
    else if (x_bytes == 1) launch_add<int8_t, int16_t>(st, x, y, wide, narrow, n16, p);
    else if (y_bytes == 1) launch_add<int16_t, int8_t>(st, x, y, wide, narrow, n16, p);
    else launch_add<int16_t, int16_t>(st, x, y, wide, narrow, n16, p);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_dequant_nhwc_to_nchw(const void* q_nhwc, int q_bytes, int g, float* y_nchw, int N, int C, int HW, int Cpad,
                                       fq_stream_t stream) {
    if ((q_bytes != 1 && q_bytes != 2) || g < -120 || g > 120) return FQ_ERR_INVALID_ARG;
    if (N < 0 || C <= 0 || HW < 0 || Cpad < C || (Cpad & 15)) return FQ_ERR_INVALID_ARG;
    if (N == 0 || HW == 0) return FQ_OK;
    if (!q_nhwc || !y_nchw || (reinterpret_cast<uintptr_t>(q_nhwc) & 15u)) return FQ_ERR_INVALID_ARG;
    if (N > 65535) return FQ_ERR_UNSUPPORTED;
    dim3 grid((HW + 63) / 64, (C + 63) / 64, N);
    const float scale = ldexpf(1.0f, -g);
    hipStream_t st = as_stream(stream);
    if (q_bytes == 1)
        hipLaunchKernelGGL(dequant_nhwc_to_nchw_kernel<int8_t>, grid, dim3(kResBlock), 0, st, static_cast<const int8_t*>(q_nhwc),
                           y_nchw, C, HW, Cpad, scale);
    else
        hipLaunchKernelGGL(dequant_nhwc_to_nchw_kernel<int16_t>, grid, dim3(kResBlock), 0, st,
                           static_cast<const int16_t*>(q_nhwc), y_nchw, C, HW, Cpad, scale);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

// ---- pooling layers between integer layers ----------------------------------------------------------
namespace fq {

// nn.MaxPool2d on a resident int8 NHWC activation.  max commutes with the (monotone) de-quantisation
// q -> q * 2^-g, so pooling the integers gives exactly the integers of the pooled fp32 tensor.  Padding
// behaves as -inf (torch): a window always holds at least one real element (pad <= kernel / 2).
// One thread = 16 channels of one output pixel.
//   * branch-free window: a tap outside the image is replaced by the nearest row / column inside it, which lies in
//     the same window (pad <= kernel / 2) -- max is idempotent, so the result is the one with -inf padding, and the
//     kh*kw loads are issued back to back instead of one per control-flow join;
//   * bytes are compared two at a time as the high bytes of packed int16 lanes (v_pk_max_i16): even bytes shifted
//     up by 8, odd bytes masked in place -- 4 vector instructions per dword and tap instead of 8.
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b) {
    const s2v x = __builtin_bit_cast(s2v, a), y = __builtin_bit_cast(s2v, b);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(x, y));
}

__global__ __launch_bounds__(kResBlock) void maxpool_i8_nhwc_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ y, int H, int W,
                                                                    int C16, int P, int Q, int kh, int kw, int sh, int sw, int ph,
                                                                    int pw, size_t total) {
    size_t i = (size_t)blockIdx.x * kResBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kResBlock;
    for (; i < total; i += stride) {                      // i = ((n*P + p)*Q + q)*C16 + c16
        const int c16 = (int)(i % C16);
        size_t r = i / C16;
        const int oq = (int)(r % Q); r /= Q;
        const int op = (int)(r % P);
        const size_t n = r / P;
        unsigned even[4], odd[4];                         // running maxima: bytes 0,2 / 1,3 of each dword, as int16 high bytes
#pragma unroll
        for (int d = 0; d < 4; ++d) even[d] = odd[d] = 0x80008000u;       // -32768: below every int8 << 8
        const int ih0 = op * sh - ph, iw0 = oq * sw - pw;
        for (int a = 0; a < kh; ++a) {
            const int ih = min(max(ih0 + a, 0), H - 1);
            const int8_t* row = x + ((n * H + ih) * W) * (size_t)C16 * 16 + (size_t)c16 * 16;
            for (int b = 0; b < kw; ++b) {
                const int iw = min(max(iw0 + b, 0), W - 1);
                const v4i_r v = *reinterpret_cast<const v4i_r*>(row + (size_t)iw * C16 * 16);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned u = (unsigned)v[d];
                    even[d] = pk_max_i16(even[d], (u << 8) & 0xff00ff00u);
                    odd[d] = pk_max_i16(odd[d], u & 0xff00ff00u);
                }
            }
        }
        v4i_r o;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = (int)(((even[d] >> 8) & 0x00ff00ffu) | (odd[d] & 0xff00ff00u));
        *reinterpret_cast<v4i_r*>(y + i * 16) = o;
    }
}

// The 3x3 / stride 2 / padding 1 pooling of a ResNet stem in its own form (round 5): the generic kernel above makes nine 16-byte
// loads per output and puts neighbouring output rows on different workgroups, i.e. XCDs with L2s of their own -- every input row
// that two output rows share came from memory twice (257 MB moved for 256 x 112 x 112 x 64 in 74 us: 3.4 TB/s).  Here a workgroup
// owns a band of kBand output rows of one image and a thread one (output column, 16-channel group) of it, walking down the band:
// the horizontal maximum of input row 2p + 1 is kept for output row p + 1, so an output costs six loads, every input row is
// read by ONE workgroup (the band's first row excepted), and the row loop is unrolled so that a thread has twelve loads in flight.
// Edges: a window position outside the plane re-reads the nearest row / column inside it (max is idempotent).
constexpr int kPoolBand = 8;

__device__ __forceinline__ void pool_split(const v4i_r v, unsigned (&ev)[4], unsigned (&od)[4]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned u = (unsigned)v[d];
        ev[d] = (u << 8) & 0xff00ff00u;                   // bytes 0, 2 as the high bytes of two int16
        od[d] = u & 0xff00ff00u;                          // bytes 1, 3
    }
}

// horizontal maximum of the three window columns of one input row, as (even, odd) int16 pairs
__device__ __forceinline__ void pool_hmax(const int8_t* __restrict__ row, int c0, int c1, int c2, unsigned (&ev)[4], unsigned (&od)[4]) {
    const v4i_r a = *reinterpret_cast<const v4i_r*>(row + c0), b = *reinterpret_cast<const v4i_r*>(row + c1),
                c = *reinterpret_cast<const v4i_r*>(row + c2);
    unsigned e1[4], o1[4], e2[4], o2[4];
    pool_split(a, ev, od); pool_split(b, e1, o1); pool_split(c, e2, o2);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        ev[d] = pk_max_i16(pk_max_i16(ev[d], e1[d]), e2[d]);
        od[d] = pk_max_i16(pk_max_i16(od[d], o1[d]), o2[d]);
    }
}

__global__ __launch_bounds__(kResBlock) void maxpool3x3s2_i8_nhwc_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ y, int H, int W,
                                                                         int C16, int P, int Q, int bands) {
    const int n = blockIdx.x / bands, band = blockIdx.x - n * bands;
    const int t = threadIdx.x;
    if (t >= Q * C16) return;
    const int oq = t / C16, c16 = t - oq * C16;
    const int rowb = W * C16 * 16;                        // bytes of one input row
    const int8_t* __restrict__ img = x + (size_t)n * H * rowb + c16 * 16;
    int8_t* __restrict__ out = y + ((size_t)n * P * Q + oq) * C16 * 16 + c16 * 16;
    const int iw = 2 * oq - 1;
    const int c0 = max(iw, 0) * C16 * 16, c1 = (iw + 1) * C16 * 16, c2 = min(iw + 2, W - 1) * C16 * 16;
    const int p0 = band * kPoolBand, p1 = min(p0 + kPoolBand, P);
    unsigned pe[4], po[4];                                // horizontal maximum of input row 2 p - 1
    pool_hmax(img + (size_t)max(2 * p0 - 1, 0) * rowb, c0, c1, c2, pe, po);
#pragma unroll 2
    for (int p = p0; p < p1; ++p) {
        unsigned ae[4], ao[4], be[4], bo[4];
        pool_hmax(img + (size_t)(2 * p) * rowb, c0, c1, c2, ae, ao);
        pool_hmax(img + (size_t)min(2 * p + 1, H - 1) * rowb, c0, c1, c2, be, bo);
        v4i_r o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned e = pk_max_i16(pk_max_i16(pe[d], ae[d]), be[d]), od = pk_max_i16(pk_max_i16(po[d], ao[d]), bo[d]);
            o[d] = (int)(((e >> 8) & 0x00ff00ffu) | (od & 0xff00ff00u));
            pe[d] = be[d]; po[d] = bo[d];
        }
        *reinterpret_cast<v4i_r*>(out + (size_t)p * Q * C16 * 16) = o;
    }
}

// Global average pooling (nn.AvgPool2d whose kernel covers the whole plane) on a resident activation:
//   y[n][c] = (sum_hw q[n][hw][c]) * 2^-g / HW.
// torch accumulates the window in fp32 and divides once; every partial sum here is an integer multiple of
// 2^-g below 2^24 * 2^-g, i.e. exact in fp32 in any order, so the integer sum, one scaling by a power of two
// and one correctly rounded fp32 division reproduce it bit for bit (HW * 2^15 < 2^24 is checked by the host).
// One workgroup = 64 channels of one image; 4 pixel phases x 64 channels, reduced through LDS.
// One workgroup per image; a thread owns 16 bytes' worth of consecutive channels (8 int16 / 16 int8) and every kPh-th pixel, read as
// 16-byte loads (round 5: the first form read one 2-byte element per lane and load -- 51 MB in 14.4 us); pixel phases meet in LDS.
template <typename T>
__global__ __launch_bounds__(kResBlock) void avgpool_global_nhwc_kernel(const T* __restrict__ q, float* __restrict__ y, int C, int HW,
                                                                        int Cpad, float scale, float divisor) {
    constexpr int V = 16 / (int)sizeof(T);                  // channels per 16-byte load
    __shared__ int part[kResBlock * V];
    const int n = blockIdx.x, groups = Cpad / V;            // 16-byte groups per pixel
    const int phases = kResBlock / groups > 0 ? kResBlock / groups : 1;
    int sum[V];
#pragma unroll
    for (int e = 0; e < V; ++e) sum[e] = 0;
    for (int g0 = 0; g0 < groups; g0 += kResBlock) {        // (more than 256 groups per pixel: several sweeps, one phase)
        const int g = g0 + (int)threadIdx.x % (groups < kResBlock ? groups : kResBlock);
        const int ph = groups < kResBlock ? (int)threadIdx.x / groups : 0;
        const bool live = g < groups && ph < phases;
#pragma unroll
        for (int e = 0; e < V; ++e) sum[e] = 0;
        if (live) {
            const T* __restrict__ src = q + (size_t)n * HW * Cpad + (size_t)g * V;
#pragma unroll 4
            for (int hw = ph; hw < HW; hw += phases) {
                const v4i_r v = *reinterpret_cast<const v4i_r*>(src + (size_t)hw * Cpad);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned u = (unsigned)v[d];
                    if constexpr (sizeof(T) == 2) {
                        sum[2 * d] += (int)(short)(u & 0xffffu);
                        sum[2 * d + 1] += (int)u >> 16;
                    } else {
                        sum[4 * d] += (int)(int8_t)(u & 0xffu);
                        sum[4 * d + 1] += (int)(int8_t)((u >> 8) & 0xffu);
                        sum[4 * d + 2] += (int)(int8_t)((u >> 16) & 0xffu);
                        sum[4 * d + 3] += (int)u >> 24;
                    }
                }
            }
        }
        if (phases > 1) {
            __syncthreads();                                // (part[] of the previous sweep has been read)
#pragma unroll
            for (int e = 0; e < V; ++e) part[threadIdx.x * V + e] = sum[e];
            __syncthreads();
            if (live && ph == 0) {
                for (int p = 1; p < phases; ++p)
#pragma unroll
                    for (int e = 0; e < V; ++e) sum[e] += part[(p * groups + g) * V + e];
            }
        }
        if (live && ph == 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int c = g * V + e;
                if (c < C) y[(size_t)n * C + c] = ((float)sum[e] * scale) / divisor;
            }
        }
    }
}

}  // namespace fq

extern "C" int fq_maxpool_i8_nhwc(const int8_t* x, int8_t* y, int N, int H, int W, int Cpad, int kh, int kw, int sh, int sw,
                                  int ph, int pw, fq_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || Cpad <= 0 || (Cpad & 15) || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0)
        return FQ_ERR_INVALID_ARG;
    if (2 * ph > kh || 2 * pw > kw) return FQ_ERR_INVALID_ARG;           // torch's own constraint: no empty windows
    const int P = (H + 2 * ph - kh) / sh + 1, Q = (W + 2 * pw - kw) / sw + 1;
    if (P <= 0 || Q <= 0) return FQ_ERR_INVALID_ARG;
    if (N == 0) return FQ_OK;
    if (!x || !y || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u)) return FQ_ERR_INVALID_ARG;
    const size_t total = (size_t)N * P * Q * (Cpad / 16);
    // FQ_POOL_GENERIC=1: always the generic kernel (A/B timing)
    static const bool generic = [] { const char* e = getenv("FQ_POOL_GENERIC"); return e && e[0] && e[0] != '0'; }();
    const long bands = (P + kPoolBand - 1) / kPoolBand;
    if (!generic && kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 && Q * (Cpad / 16) <= kResBlock && H >= 2 && W >= 2 &&
        (long)N * bands <= 0x7fffffffL && 2 * (P - 1) < H && 2 * (Q - 1) < W) {
        hipLaunchKernelGGL(maxpool3x3s2_i8_nhwc_kernel, dim3((unsigned)(N * bands)), dim3(kResBlock), 0, as_stream(stream), x, y, H, W,
                           Cpad / 16, P, Q, (int)bands);
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    }
    hipLaunchKernelGGL(maxpool_i8_nhwc_kernel, dim3(res_grid(total)), dim3(kResBlock), 0, as_stream(stream), x, y, H, W, Cpad / 16, P,
                       Q, kh, kw, sh, sw, ph, pw, total);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_avgpool_global_nhwc(const void* q_nhwc, int q_bytes, int g, float* y, int N, int C, int HW, int Cpad,
                                      fq_stream_t stream) {
    if ((q_bytes != 1 && q_bytes != 2) || g < -120 || g > 120) return FQ_ERR_INVALID_ARG;
    if (N < 0 || C <= 0 || HW <= 0 || Cpad < C || (Cpad & 15)) return FQ_ERR_INVALID_ARG;
    if ((long)HW * 32768 >= (1L << 24)) return FQ_ERR_UNSUPPORTED;       // partial sums must stay exact in fp32
    if (N == 0) return FQ_OK;
    if (!q_nhwc || !y || (reinterpret_cast<uintptr_t>(q_nhwc) & 15u)) return FQ_ERR_INVALID_ARG;
    dim3 grid((unsigned)N);
    hipStream_t st = as_stream(stream);
    const float scale = ldexpf(1.0f, -g), divisor = (float)HW;
    if (q_bytes == 1)
        hipLaunchKernelGGL(avgpool_global_nhwc_kernel<int8_t>, grid, dim3(kResBlock), 0, st, static_cast<const int8_t*>(q_nhwc), y, C,
                           HW, Cpad, scale, divisor);
    else
        hipLaunchKernelGGL(avgpool_global_nhwc_kernel<int16_t>, grid, dim3(kResBlock), 0, st, static_cast<const int16_t*>(q_nhwc), y,
                           C, HW, Cpad, scale, divisor);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}
