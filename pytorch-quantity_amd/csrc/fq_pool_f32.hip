// fq_pool_f32.hip -- the two pooling layers of the float calibration forward, as torch computes them bit for bit, at
// the rate their traffic allows.  torch's max_pool_forward_nchw takes 0.55 ms for ResNet-50's 3x3/2 pool at 256 images
// (1.03 GB of traffic: 0.2 ms at 5 TB/s) and avg_pool2d_out_cuda_frame 0.37 ms for the 7x7 global average (103 MB);
// both run once per forward of pass 1 and the max-pool again in the prefix pass 2 re-computes.
//
// max pool: one thread per output, window clipped to the image (padding is -inf: never selected), NaN propagates the way
//   torch's kernel does it (`val > max || isnan(val)`); any kernel / stride / padding, no dilation, floor mode.
// global average pool (kernel == plane, no padding): a workgroup stages 256 consecutive planes in LDS with coalesced
//   loads, then thread t adds plane t's values one by one in row-major order in fp32 and divides by the count --
//   the same sequence of roundings as torch's loop (an LDS stride of HW words is conflict-free for odd HW).
#include "fq_common.h"

namespace fq {
namespace {

constexpr int kPoolBlock = 256;

__global__ __launch_bounds__(kPoolBlock) void maxpool2d_f32_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                   unsigned total, int H, int W, int Ho, int Wo, int kh, int kw,
                                                                   int sh, int sw, int ph, int pw) {
    const unsigned stride = gridDim.x * kPoolBlock;
    for (unsigned o = blockIdx.x * kPoolBlock + threadIdx.x; o < total; o += stride) {
        const unsigned ox = o % (unsigned)Wo, t = o / (unsigned)Wo, oy = t % (unsigned)Ho, plane = t / (unsigned)Ho;
        const int y0 = (int)oy * sh - ph, x0 = (int)ox * sw - pw;
        const int ya = y0 < 0 ? 0 : y0, xa = x0 < 0 ? 0 : x0;
        const int yb = y0 + kh < H ? y0 + kh : H, xb = x0 + kw < W ? x0 + kw : W;
        const float* __restrict__ p = x + (size_t)plane * H * W;
        float m = -INFINITY;
        for (int iy = ya; iy < yb; ++iy)
            for (int ix = xa; ix < xb; ++ix) {
                const float v = p[iy * W + ix];
                if (v > m || v != v) m = v;
            }
        y[o] = m;
    }
}

// The ResNet pool (3x3, stride 2, padding 1) on rows of 16-byte-aligned quads: a thread produces four neighbouring
// outputs from 3 rows x (two aligned float4 + the one column to their left) -- 9 loads for 4 outputs instead of 36.
// The maximum of a window does not depend on the order it is scanned in, and the NaN rule above is sticky, so this
// gives the same bits as the scan in window order.
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float pmax(float m, float v) { return (v > m || v != v) ? v : m; }

__global__ __launch_bounds__(kPoolBlock) void maxpool3x3s2p1_f32_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                        unsigned quads, int H, int W, int Ho, int Wo) {
    const unsigned stride = gridDim.x * kPoolBlock, qrow = (unsigned)Wo >> 2;
    for (unsigned q = blockIdx.x * kPoolBlock + threadIdx.x; q < quads; q += stride) {
        const unsigned qx = q % qrow, t = q / qrow, oy = t % (unsigned)Ho, plane = t / (unsigned)Ho;
        const int ix0 = (int)qx * 8, y0 = (int)oy * 2 - 1;
        const float* __restrict__ p = x + (size_t)plane * H * W;
        float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = y0 + r;
            if (iy >= 0 && iy < H) {
                const float* row = p + (size_t)iy * W + ix0;
                const f4v a = *reinterpret_cast<const f4v*>(row), b = *reinterpret_cast<const f4v*>(row + 4);
                const float l = ix0 > 0 ? row[-1] : -INFINITY;
                m0 = pmax(pmax(pmax(m0, l), a.x), a.y);
                m1 = pmax(pmax(pmax(m1, a.y), a.z), a.w);
                m2 = pmax(pmax(pmax(m2, a.w), b.x), b.y);
                m3 = pmax(pmax(pmax(m3, b.y), b.z), b.w);
            }
        }
        *reinterpret_cast<f4v*>(y + (size_t)q * 4) = f4v{m0, m1, m2, m3};
    }
}

__global__ __launch_bounds__(kPoolBlock) void avgpool_global_f32_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                        unsigned planes, int HW) {
    extern __shared__ float s_planes[];                        // [256][HW]
    const unsigned first = blockIdx.x * kPoolBlock;
    const unsigned here = planes - first < (unsigned)kPoolBlock ? planes - first : (unsigned)kPoolBlock;
    const float* __restrict__ src = x + (size_t)first * HW;
    for (unsigned i = threadIdx.x; i < here * (unsigned)HW; i += kPoolBlock) s_planes[i] = src[i];
    __syncthreads();
    if (threadIdx.x < here) {
        const float* v = s_planes + threadIdx.x * HW;
        float acc = 0.0f;
        for (int k = 0; k < HW; ++k) acc += v[k];
        y[first + threadIdx.x] = acc / (float)HW;
    }
}

}  // namespace
}  // namespace fq

using namespace fq;

extern "C" int fq_maxpool2d_f32(const float* x, float* y, int planes, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                                fq_stream_t stream) {
    if (planes < 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return FQ_ERR_INVALID_ARG;
    if (2 * ph > kh || 2 * pw > kw || H + 2 * ph < kh || W + 2 * pw < kw) return FQ_ERR_INVALID_ARG;      // torch's own constraints
    const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    const size_t total = (size_t)planes * Ho * Wo;
    if (total == 0) return FQ_OK;
    if (!x || !y) return FQ_ERR_INVALID_ARG;
    if (total >= 0xffffffffULL || (size_t)planes * H * W >= 0xffffffffULL) return FQ_ERR_UNSUPPORTED;
    if (kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 && (W & 7) == 0 && (Wo & 3) == 0 && Wo * 2 == W &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0) {
        const size_t quads = total >> 2;
        size_t blocks = (quads + kPoolBlock - 1) / kPoolBlock;
        if (blocks > (size_t)kCUs * 32) blocks = (size_t)kCUs * 32;
        hipLaunchKernelGGL(maxpool3x3s2p1_f32_kernel, dim3((unsigned)blocks), dim3(kPoolBlock), 0, as_stream(stream), x, y,
                           (unsigned)quads, H, W, Ho, Wo);
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    }
    size_t blocks = (total + kPoolBlock - 1) / kPoolBlock;
    if (blocks > (size_t)kCUs * 64) blocks = (size_t)kCUs * 64;
    hipLaunchKernelGGL(maxpool2d_f32_kernel, dim3((unsigned)blocks), dim3(kPoolBlock), 0, as_stream(stream), x, y, (unsigned)total,
                       H, W, Ho, Wo, kh, kw, sh, sw, ph, pw);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_avgpool_global_f32(const float* x, float* y, int planes, int HW, fq_stream_t stream) {
    if (planes < 0 || HW <= 0) return FQ_ERR_INVALID_ARG;
    if (HW > 144) return FQ_ERR_UNSUPPORTED;                     // 256 planes of a workgroup must fit LDS (144 KB at HW = 144)
    if (planes == 0) return FQ_OK;
    if (!x || !y) return FQ_ERR_INVALID_ARG;
    const size_t dyn = (size_t)kPoolBlock * HW * sizeof(float);
    static bool lds_ok[kMaxDevices] = {};
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(avgpool_global_f32_kernel), 144 * kPoolBlock * (int)sizeof(float), lds_ok))
        return hip_fail(hipErrorInvalidValue);
    const unsigned blocks = ((unsigned)planes + kPoolBlock - 1) / kPoolBlock;
    hipLaunchKernelGGL(avgpool_global_f32_kernel, dim3(blocks), dim3(kPoolBlock), dyn, as_stream(stream), x, y, (unsigned)planes, HW);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}
