// fq_ops.hip -- element-wise quantisation ops of quantity/common/quantity/new_quantity_op.py as
// single-pass, 16-byte-vectorised, HBM-bound kernels for gfx950.
//
// The reference builds each op from 3-7 torch element-wise calls (one full tensor round trip each);
// here every op -- and the whole NewConv2d tail RightShift -> BiasAdd -> Sp -> DeQuantity -- is one
// read and one write per element.  Arithmetic follows the reference exactly:
//   torch.mul/div by pow(2,k)  -> one fp32 multiply/divide by the exact power of two
//   torch.round                -> round half to even (v_rndne_f32 / rintf)
//   torch.clamp                -> NaN-propagating clamp
//   RightShift                 -> trunc(v + (v > 0 ? 0.5 : -0.5)) through int32, saturating cast
#include <cstdlib>

#include "fq_common.h"
#include "fq_hist_bin.h"
#include "fq_producer_stat.h"

namespace fq {

constexpr int kOpsBlock = 256;

__device__ __forceinline__ float clamp_nan(float v, float lo, float hi) {
    return v < lo ? lo : (v > hi ? hi : v);        // NaN fails both compares and passes through
}

struct Range { float lo, hi; };
__host__ __device__ inline Range range_of(int bitwidth) {
    return bitwidth == 8 ? Range{-128.0f, 127.0f} : Range{-32768.0f, 32767.0f};
}

// ---- functors ---------------------------------------------------------------------------------
// Division by 2^k is done as multiplication by 2^-k: both are the correctly rounded image of the
// same real number (|k| <= 120 keeps 2^-k normal), so the bits are identical and the divide
// sequence is avoided.
struct QuanDequanOp {      // new_quantity_op.py:246-257
    float scale, inv, lo, hi;
    __device__ __forceinline__ float operator()(float x) const {
        return clamp_nan(rintf(x * scale), lo, hi) * inv;
    }
};
struct QuantityOp {        // :52-58
    float scale, lo, hi;
    __device__ __forceinline__ float operator()(float x) const { return clamp_nan(rintf(x * scale), lo, hi); }
};
struct DeQuantityOp {      // :66-68
    float inv;
    __device__ __forceinline__ float operator()(float x) const { return x * inv; }
};
struct SpOp {              // :76-91
    float lo, hi;
    __device__ __forceinline__ float operator()(float x) const { return clamp_nan(x, lo, hi); }
};
struct RightShiftOp {      // :17-44
    float inv; int ilo, ihi;
    __device__ __forceinline__ float operator()(float x) const {
        const float v = x * inv;
        const float w = v + (v > 0.0f ? 0.5f : -0.5f);
        int r = (int)w;                                   // v_cvt_i32_f32: truncates, saturates
        r = r < ilo ? ilo : (r > ihi ? ihi : r);
        return (float)r;
    }
};

// kStream: the tensor is larger than the 256 MB Infinity Cache, so nothing of it will be reused on chip --
// non-temporal loads and stores plus a grid of up to 128 workgroups per CU moved the fused fake-quant from
// 4.8 to 6.3-6.9 TB/s read+write on the 100 M-element ResNet-50 tensors (scripts/ops_sweep.py); on tensors that
// fit in the cache the same hints cost up to 15 % (the consumer finds nothing there), so they keep the plain form.
typedef float f4v __attribute__((ext_vector_type(4)));
template <typename Op, bool kStream>
__global__ __launch_bounds__(kOpsBlock) void unary_vec_kernel(const f4v* __restrict__ x, f4v* __restrict__ y, size_t nvec, Op op) {
    size_t i = (size_t)blockIdx.x * kOpsBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kOpsBlock;
    for (; i + stride < nvec; i += 2 * stride) {
        f4v a = kStream ? __builtin_nontemporal_load(&x[i]) : x[i];
        f4v b = kStream ? __builtin_nontemporal_load(&x[i + stride]) : x[i + stride];
        a.x = op(a.x); a.y = op(a.y); a.z = op(a.z); a.w = op(a.w);
        b.x = op(b.x); b.y = op(b.y); b.z = op(b.z); b.w = op(b.w);
        if (kStream) {
            __builtin_nontemporal_store(a, &y[i]);
            __builtin_nontemporal_store(b, &y[i + stride]);
        } else {
            y[i] = a; y[i + stride] = b;
        }
    }
    for (; i < nvec; i += stride) {
        f4v a = x[i];
        a.x = op(a.x); a.y = op(a.y); a.z = op(a.z); a.w = op(a.w);
        y[i] = a;
    }
}

template <typename Op>
__global__ __launch_bounds__(kOpsBlock) void unary_scalar_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 size_t n, Op op) {
    size_t i = (size_t)blockIdx.x * kOpsBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kOpsBlock;
    for (; i < n; i += stride) y[i] = op(x[i]);
}

constexpr size_t kStreamBytes = (size_t)256 << 20;   // Infinity Cache: beyond this a pass is pure streaming

inline unsigned grid_for(size_t work_items, int wg_per_cu = 16) {
    size_t g = (work_items + kOpsBlock - 1) / kOpsBlock;
    const size_t cap = (size_t)kCUs * wg_per_cu;  // grid-stride the rest
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (unsigned)g;
}

template <typename Op>
static int launch_unary(const float* x, float* y, size_t n, Op op, fq_stream_t stream) {
    if (n == 0) return FQ_OK;
    if (x == nullptr || y == nullptr) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
    if (aligned) {
        const size_t nvec = n >> 2;
        if (nvec) {
            const f4v* xv = reinterpret_cast<const f4v*>(x);
            f4v* yv = reinterpret_cast<f4v*>(y);
            if (n * 8 > kStreamBytes)
                hipLaunchKernelGGL((unary_vec_kernel<Op, true>), dim3(grid_for(nvec, 128)), dim3(kOpsBlock), 0, st, xv, yv, nvec, op);
            else
                hipLaunchKernelGGL((unary_vec_kernel<Op, false>), dim3(grid_for(nvec)), dim3(kOpsBlock), 0, st, xv, yv, nvec, op);
            FQ_LAUNCH_CHECK();
        }
        const size_t tail = n & 3u;
        if (tail) {
            hipLaunchKernelGGL(unary_scalar_kernel<Op>, dim3(1), dim3(kOpsBlock), 0, st, x + (nvec << 2),
                               y + (nvec << 2), tail, op);
            FQ_LAUNCH_CHECK();
        }
    } else {
        hipLaunchKernelGGL(unary_scalar_kernel<Op>, dim3(grid_for(n)), dim3(kOpsBlock), 0, st, x, y, n, op);
        FQ_LAUNCH_CHECK();
    }
    return FQ_OK;
}

// ---- NewAdd: clamp(a + b) -----------------------------------------------------------------------
template <bool kStream>
__global__ __launch_bounds__(kOpsBlock) void add_sat_vec_kernel(const f4v* __restrict__ a, const f4v* __restrict__ b,
                                                                f4v* __restrict__ y, size_t nvec, float lo, float hi) {
    size_t i = (size_t)blockIdx.x * kOpsBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kOpsBlock;
    for (; i < nvec; i += stride) {
        const f4v p = kStream ? __builtin_nontemporal_load(&a[i]) : a[i];
        const f4v q = kStream ? __builtin_nontemporal_load(&b[i]) : b[i];
        f4v r;
        r.x = clamp_nan(p.x + q.x, lo, hi); r.y = clamp_nan(p.y + q.y, lo, hi);
        r.z = clamp_nan(p.z + q.z, lo, hi); r.w = clamp_nan(p.w + q.w, lo, hi);
        if (kStream) __builtin_nontemporal_store(r, &y[i]); else y[i] = r;
    }
}
__global__ __launch_bounds__(kOpsBlock) void add_sat_scalar_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                   float* __restrict__ y, size_t n, float lo, float hi) {
    size_t i = (size_t)blockIdx.x * kOpsBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kOpsBlock;
    for (; i < n; i += stride) y[i] = clamp_nan(a[i] + b[i], lo, hi);
}

// ---- NewConv2d/NewLinear tail: acc[outer][C][inner] ---------------------------------------------
// one workgroup-row per (outer, c) plane so the bias is uniform; planes are contiguous runs of `inner`.
__global__ __launch_bounds__(kOpsBlock) void recon_epilogue_kernel(const float* __restrict__ acc, const float* __restrict__ qbias,
                                                                   float* __restrict__ y, size_t planes, size_t C, size_t inner,
                                                                   RightShiftOp rs, float lo, float hi, float oinv) {
    // flat grid-stride over elements; channel = (i / inner) % C
    size_t i = (size_t)blockIdx.x * kOpsBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kOpsBlock;
    const size_t n = planes * inner;
    for (; i < n; i += stride) {
        const size_t c = (i / inner) % C;
        const float v = rs(acc[i]) + qbias[c];
        y[i] = clamp_nan(v, lo, hi) * oinv;
    }
}
// inner % 4 == 0 and 16-B aligned: 4 consecutive elements share a channel
__global__ __launch_bounds__(kOpsBlock) void recon_epilogue_vec_kernel(const float4* __restrict__ acc, const float* __restrict__ qbias,
                                                                       float4* __restrict__ y, size_t nvec, size_t C, size_t inner4,
                                                                       RightShiftOp rs, float lo, float hi, float oinv) {
    size_t i = (size_t)blockIdx.x * kOpsBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kOpsBlock;
    for (; i < nvec; i += stride) {
        const float bq = qbias[(i / inner4) % C];
        float4 a = acc[i];
        a.x = clamp_nan(rs(a.x) + bq, lo, hi) * oinv;
        a.y = clamp_nan(rs(a.y) + bq, lo, hi) * oinv;
        a.z = clamp_nan(rs(a.z) + bq, lo, hi) * oinv;
        a.w = clamp_nan(rs(a.w) + bq, lo, hi) * oinv;
        y[i] = a;
    }
}

// ---- conv bias add / residual add of the float model with a calibration statistic taken on the way out --------------
// A float Conv2d on this stack is a MIOpen convolution followed by a separate broadcast add of the bias (8 B per
// element); the calibration then reads the result once more for the abs-max (pass 1) or the histogram (pass 2), 4 B.
// These kernels ARE that add -- y[n][c][hw] += bias[c], the same single fp32 rounding; z = x + y for the Eltwise module
// (fabu_layer.py:16-19) -- and fold the statistic in while the values are in registers:
//   Stat = MaxStat   pass 1 (distribution_collector.py:70-78): max |y| into the tensor's row of the collector;
//   Stat = HistStat  pass 2 (distribution_collector.py:127-135): every value binned (fq_hist_bin.h) into an 8 KB LDS
//                    histogram per workgroup, flushed with 64-bit atomics at the end.  A tensor that was just written
//                    is the worst input for the streaming histogram kernel (it shares HBM with the write-back of the
//                    very lines it reads) and would cost a second read; few, long-lived workgroups
//                    (2 per CU) keep the flush at 2048 bins x 512.
// relu_out != nullptr: the nn.ReLU that consumes the tensor is served in the same pass (r = y > 0 ? y : 0 with NaN kept,
// torch's clamp_min): one more 4-byte write instead of a separate 8-byte pass.
// kVec: HW % 4 == 0 and 16-byte aligned pointers.  kStream: the launch moves more than the 256 MB Infinity Cache holds --
// non-temporal loads and stores, as in unary_vec_kernel.  Two items per lane are in flight per iteration, and the
// channel of an item is carried along incrementally (plane = i / HW, c = plane % C advance by constants per grid stride:
// two adds and two selects instead of two integer divisions per item).
template <bool kStream> __device__ __forceinline__ f4v ld4(const f4v* p) { return kStream ? __builtin_nontemporal_load(p) : *p; }
template <bool kStream> __device__ __forceinline__ void st4(f4v v, f4v* p) { if (kStream) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ f4v relu4(f4v v) {
    f4v r;
    r.x = relu_like_torch(v.x); r.y = relu_like_torch(v.y); r.z = relu_like_torch(v.z); r.w = relu_like_torch(v.w);
    return r;
}

template <bool kVec, bool kStream, typename Stat>
__device__ __forceinline__ void bias_add_body(float* __restrict__ y, const float* __restrict__ bias, unsigned n_items,
                                              unsigned inner, unsigned C, float* __restrict__ relu_out, Stat& stat) {
    const size_t stride = (size_t)gridDim.x * kOpsBlock;
    size_t i = (size_t)blockIdx.x * kOpsBlock + threadIdx.x;
    // item i lies in plane i / inner (inner = HW, or HW / 4 in the vector form), channel plane % C
    unsigned plane = (unsigned)(i / inner), rem = (unsigned)(i - (size_t)plane * inner), c = plane % C;
    const unsigned dq = (unsigned)(stride / inner), dr = (unsigned)(stride - (size_t)dq * inner), dc = dq % C;
    auto next_channel = [&]() {                               // (rem, c) of item i + stride
        rem += dr;
        const unsigned carry = rem >= inner ? 1u : 0u;
        rem -= carry ? inner : 0u;
        c += dc + carry;
        c -= c >= C ? C : 0u;
    };
    for (; i < n_items; i += 2 * stride) {
        const size_t i1 = i + stride;
        const bool two = i1 < n_items;
        const float b0 = bias[c];
        next_channel();
        const float b1 = bias[c];
        next_channel();
        if (kVec) {
            f4v* y4 = reinterpret_cast<f4v*>(y);
            f4v* r4 = reinterpret_cast<f4v*>(relu_out);
            f4v v0 = ld4<kStream>(&y4[i]);
            f4v v1 = two ? ld4<kStream>(&y4[i1]) : f4v{0.f, 0.f, 0.f, 0.f};
            v0.x += b0; v0.y += b0; v0.z += b0; v0.w += b0;
            v1.x += b1; v1.y += b1; v1.z += b1; v1.w += b1;
            st4<kStream>(v0, &y4[i]);
            if (relu_out) st4<kStream>(relu4(v0), &r4[i]);
            stat.add(v0.x); stat.add(v0.y); stat.add(v0.z); stat.add(v0.w);
            if (two) {
                st4<kStream>(v1, &y4[i1]);
                if (relu_out) st4<kStream>(relu4(v1), &r4[i1]);
                stat.add(v1.x); stat.add(v1.y); stat.add(v1.z); stat.add(v1.w);
            }
        } else {
            const float v0 = y[i] + b0;
            y[i] = v0;
            if (relu_out) relu_out[i] = relu_like_torch(v0);
            stat.add(v0);
            if (two) {
                const float v1 = y[i1] + b1;
                y[i1] = v1;
                if (relu_out) relu_out[i1] = relu_like_torch(v1);
                stat.add(v1);
            }
        }
    }
}

template <bool kStream, typename Stat>
__device__ __forceinline__ void add_body(const f4v* __restrict__ x, const f4v* __restrict__ y, f4v* __restrict__ z, size_t nvec,
                                         const float* __restrict__ xs, const float* __restrict__ ys, float* __restrict__ zs,
                                         unsigned tail, float* __restrict__ relu_out, Stat& stat) {
    const size_t stride = (size_t)gridDim.x * kOpsBlock;
    f4v* r4 = reinterpret_cast<f4v*>(relu_out);
    for (size_t i = (size_t)blockIdx.x * kOpsBlock + threadIdx.x; i < nvec; i += 2 * stride) {
        const size_t i1 = i + stride;
        const bool two = i1 < nvec;
        const f4v a0 = ld4<kStream>(&x[i]), c0 = ld4<kStream>(&y[i]);
        const f4v a1 = two ? ld4<kStream>(&x[i1]) : f4v{0.f, 0.f, 0.f, 0.f};
        const f4v c1 = two ? ld4<kStream>(&y[i1]) : f4v{0.f, 0.f, 0.f, 0.f};
        const f4v v0 = a0 + c0, v1 = a1 + c1;
        st4<kStream>(v0, &z[i]);
        if (relu_out) st4<kStream>(relu4(v0), &r4[i]);
        stat.add(v0.x); stat.add(v0.y); stat.add(v0.z); stat.add(v0.w);
        if (two) {
            st4<kStream>(v1, &z[i1]);
            if (relu_out) st4<kStream>(relu4(v1), &r4[i1]);
            stat.add(v1.x); stat.add(v1.y); stat.add(v1.z); stat.add(v1.w);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < tail) {          // the last n % 4 elements
        const float v = xs[threadIdx.x] + ys[threadIdx.x];
        zs[threadIdx.x] = v;
        if (relu_out) relu_out[(nvec << 2) + threadIdx.x] = relu_like_torch(v);
        stat.add(v);
    }
}

// workgroup maximum -> the collector's row.  m >= 0: the bit pattern orders like an unsigned.  Thousands of workgroups
// publish into ONE word: only those that can still raise it pay the atomic (4 096 serialised atomics cost 40 us, more
// than the add itself)
template <bool kVec, bool kStream>
__global__ __launch_bounds__(kOpsBlock) void bias_add_absmax_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                                    unsigned n_items, unsigned inner, unsigned C,
                                                                    unsigned int* __restrict__ max_bits, float* __restrict__ relu_out) {
    MaxStat st;
    bias_add_body<kVec, kStream>(y, bias, n_items, inner, C, relu_out, st);
    publish_max<kOpsBlock>(st.m, max_bits);
}

__device__ __forceinline__ unsigned int* hist_lds_zeroed() {
    __shared__ unsigned int s_bins[FQ_BINS + kWave];
    for (int b = threadIdx.x; b < FQ_BINS + kWave; b += kOpsBlock) s_bins[b] = 0u;
    __syncthreads();
    return s_bins;
}

template <bool kVec, bool kStream>
__global__ __launch_bounds__(kOpsBlock) void bias_add_hist_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                                  unsigned n_items, unsigned inner, unsigned C,
                                                                  const float* __restrict__ interval,
                                                                  unsigned long long* __restrict__ hist_row,
                                                                  float* __restrict__ relu_out, const int allow_fast) {
    unsigned int* s_bins = hist_lds_zeroed();
    const float iv = *interval;
    unsigned int* park = s_bins + FQ_BINS + (threadIdx.x & (kWave - 1));
    if (allow_fast && fast_quotient_ok(iv)) {
        HistStat<true> st{s_bins, park, iv, 1.0f / iv};
        bias_add_body<kVec, kStream>(y, bias, n_items, inner, C, relu_out, st);
    } else {
        HistStat<false> st{s_bins, park, iv, 1.0f / iv};
        bias_add_body<kVec, kStream>(y, bias, n_items, inner, C, relu_out, st);
    }
    hist_flush<kOpsBlock>(s_bins, hist_row);
}

template <bool kStream>
__global__ __launch_bounds__(kOpsBlock) void add_absmax_kernel(const f4v* __restrict__ x, const f4v* __restrict__ y,
                                                               f4v* __restrict__ z, size_t nvec, const float* __restrict__ xs,
                                                               const float* __restrict__ ys, float* __restrict__ zs, unsigned tail,
                                                               unsigned int* __restrict__ max_bits, float* __restrict__ relu_out) {
    MaxStat st;
    add_body<kStream>(x, y, z, nvec, xs, ys, zs, tail, relu_out, st);
    publish_max<kOpsBlock>(st.m, max_bits);
}

template <bool kStream>
__global__ __launch_bounds__(kOpsBlock) void add_hist_kernel(const f4v* __restrict__ x, const f4v* __restrict__ y,
                                                             f4v* __restrict__ z, size_t nvec, const float* __restrict__ xs,
                                                             const float* __restrict__ ys, float* __restrict__ zs, unsigned tail,
                                                             const float* __restrict__ interval,
                                                             unsigned long long* __restrict__ hist_row,
                                                             float* __restrict__ relu_out, const int allow_fast) {
    unsigned int* s_bins = hist_lds_zeroed();
    const float iv = *interval;
    unsigned int* park = s_bins + FQ_BINS + (threadIdx.x & (kWave - 1));
    if (allow_fast && fast_quotient_ok(iv)) {
        HistStat<true> st{s_bins, park, iv, 1.0f / iv};
        add_body<kStream>(x, y, z, nvec, xs, ys, zs, tail, relu_out, st);
    } else {
        HistStat<false> st{s_bins, park, iv, 1.0f / iv};
        add_body<kStream>(x, y, z, nvec, xs, ys, zs, tail, relu_out, st);
    }
    hist_flush<kOpsBlock>(s_bins, hist_row);
}

// ---- weight quantiser ---------------------------------------------------------------------------
__global__ __launch_bounds__(kOpsBlock) void quantize_param_i32_kernel(const float* __restrict__ w, int32_t* __restrict__ q,
                                                                       size_t n, float scale) {
    size_t i = (size_t)blockIdx.x * kOpsBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kOpsBlock;
    for (; i < n; i += stride) {
        float r = rintf(w[i] * scale);
        r = fminf(fmaxf(r, -128.0f), 127.0f);             // np.clip
        q[i] = (int32_t)r;
    }
}

}  // namespace fq

using namespace fq;

extern "C" int fq_quandequan_f32(const float* x, float* y, size_t n, int bit, int bitwidth, fq_stream_t stream) {
    if (!valid_bitwidth(bitwidth) || bit < -120 || bit > 120) return FQ_ERR_INVALID_ARG;
    const Range r = range_of(bitwidth);
    return launch_unary(x, y, n, QuanDequanOp{ldexpf(1.0f, bit), ldexpf(1.0f, -bit), r.lo, r.hi}, stream);
}

extern "C" int fq_quantity_f32(const float* x, float* y, size_t n, int ib, int bitwidth, fq_stream_t stream) {
    if (!valid_bitwidth(bitwidth) || ib < -120 || ib > 120) return FQ_ERR_INVALID_ARG;
    const Range r = range_of(bitwidth);
    return launch_unary(x, y, n, QuantityOp{ldexpf(1.0f, ib), r.lo, r.hi}, stream);
}

extern "C" int fq_dequantity_f32(const float* x, float* y, size_t n, int ob, fq_stream_t stream) {
    if (ob < -120 || ob > 120) return FQ_ERR_INVALID_ARG;
    return launch_unary(x, y, n, DeQuantityOp{ldexpf(1.0f, -ob)}, stream);
}

extern "C" int fq_sp_f32(const float* x, float* y, size_t n, int bitwidth, fq_stream_t stream) {
    if (!valid_bitwidth(bitwidth)) return FQ_ERR_INVALID_ARG;
    const Range r = range_of(bitwidth);
    return launch_unary(x, y, n, SpOp{r.lo, r.hi}, stream);
}

extern "C" int fq_rightshift_f32(const float* x, float* y, size_t n, int rs, int bitwidth, fq_stream_t stream) {
    if (!valid_bitwidth(bitwidth) || rs < -120 || rs > 120) return FQ_ERR_INVALID_ARG;
    const Range r = range_of(bitwidth);
    return launch_unary(x, y, n, RightShiftOp{ldexpf(1.0f, -rs), (int)r.lo, (int)r.hi}, stream);
}

extern "C" int fq_add_sat_f32(const float* a, const float* b, float* y, size_t n, int bitwidth, fq_stream_t stream) {
    if (!valid_bitwidth(bitwidth)) return FQ_ERR_INVALID_ARG;
    if (n == 0) return FQ_OK;
    if (!a || !b || !y) return FQ_ERR_INVALID_ARG;
    const Range r = range_of(bitwidth);
    hipStream_t st = as_stream(stream);
    const bool aligned = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                           reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
    size_t done = 0;
    if (aligned && (n >> 2)) {
        const size_t nvec = n >> 2;
        const f4v* av = reinterpret_cast<const f4v*>(a);
        const f4v* bv = reinterpret_cast<const f4v*>(b);
        f4v* yv = reinterpret_cast<f4v*>(y);
        if (n * 12 > kStreamBytes)
            hipLaunchKernelGGL(add_sat_vec_kernel<true>, dim3(grid_for(nvec, 128)), dim3(kOpsBlock), 0, st, av, bv, yv, nvec, r.lo, r.hi);
        else
            hipLaunchKernelGGL(add_sat_vec_kernel<false>, dim3(grid_for(nvec)), dim3(kOpsBlock), 0, st, av, bv, yv, nvec, r.lo, r.hi);
        FQ_LAUNCH_CHECK();
        done = nvec << 2;
    }
    if (done < n) {
        hipLaunchKernelGGL(add_sat_scalar_kernel, dim3(grid_for(n - done)), dim3(kOpsBlock), 0, st, a + done,
                           b + done, y + done, n - done, r.lo, r.hi);
        FQ_LAUNCH_CHECK();
    }
    return FQ_OK;
}

extern "C" int fq_recon_epilogue_f32(const float* acc, const float* qbias, float* y, size_t outer, size_t C,
                                     size_t inner, int rs, int ob, int bitwidth, fq_stream_t stream) {
    if (!valid_bitwidth(bitwidth) || rs < -120 || rs > 120 || ob < -120 || ob > 120) return FQ_ERR_INVALID_ARG;
    const size_t n = outer * C * inner;
    if (n == 0) return FQ_OK;
    if (!acc || !qbias || !y) return FQ_ERR_INVALID_ARG;
    const Range r = range_of(bitwidth);
    const RightShiftOp rso{ldexpf(1.0f, -rs), (int)r.lo, (int)r.hi};
    const float oscale = ldexpf(1.0f, -ob);
    hipStream_t st = as_stream(stream);
    const bool aligned = ((reinterpret_cast<uintptr_t>(acc) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
    if (aligned && (inner & 3u) == 0) {
        const size_t nvec = n >> 2;
        hipLaunchKernelGGL(recon_epilogue_vec_kernel, dim3(grid_for(nvec)), dim3(kOpsBlock), 0, st,
                           reinterpret_cast<const float4*>(acc), qbias, reinterpret_cast<float4*>(y), nvec, C,
                           inner >> 2, rso, r.lo, r.hi, oscale);
    } else {
        hipLaunchKernelGGL(recon_epilogue_kernel, dim3(grid_for(n)), dim3(kOpsBlock), 0, st, acc, qbias, y,
                           outer * C, C, inner, rso, r.lo, r.hi, oscale);
    }
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

// Workgroups per CU of the streaming form, measured on the ResNet-50 calibration at batch 256 (aggregate TB/s over all
// launches of a forward, 8 / 16 / 32 / 64 / 128 / 256 per CU): bias add + abs-max 5.31 / 5.20 / 5.48 / 5.65 / 5.61 / 5.71, residual
// add + abs-max 5.29 / 5.11 / 5.33 / 5.60 / 5.82 / 5.98 (more is better: nothing is flushed), bias add + histogram (8 .. 64) 4.75 / 4.26 / 3.32 / 2.09, residual
// add + histogram 5.14 / 4.85 / 4.65 / 3.41 (every workgroup flushes up to 2048 bins).  FQ_PRODUCER_WG_PER_CU overrides all four.
static int producer_wg_per_cu(int dflt) {
    static const int v = [] { const char* e = getenv("FQ_PRODUCER_WG_PER_CU"); return e ? atoi(e) : 0; }();
    return v > 0 ? v : dflt;
}

extern "C" int fq_bias_add_absmax_f32(float* y, const float* bias, int N, int C, int HW, float* max_inout, float* relu_out,
                                      fq_stream_t stream) {
    using namespace fq;
    if (N < 0 || C <= 0 || HW <= 0) return FQ_ERR_INVALID_ARG;
    const size_t n = (size_t)N * C * HW;
    if (n == 0) return FQ_OK;
    if (!y || !bias || !max_inout) return FQ_ERR_INVALID_ARG;
    if (n >= 0xffffffffULL) return FQ_ERR_UNSUPPORTED;               // 32-bit element index inside the kernel
    hipStream_t st = as_stream(stream);
    unsigned int* bits = reinterpret_cast<unsigned int*>(max_inout);
    const bool stream_form = n * (relu_out ? 12 : 8) > kStreamBytes;
    if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(relu_out)) & 15u) == 0) {
        const unsigned nvec = (unsigned)(n >> 2);
        if (stream_form)
            hipLaunchKernelGGL((bias_add_absmax_kernel<true, true>), dim3(grid_for(nvec, producer_wg_per_cu(256))), dim3(kOpsBlock), 0, st,
                               y, bias, nvec, (unsigned)(HW >> 2), (unsigned)C, bits, relu_out);
        else
            hipLaunchKernelGGL((bias_add_absmax_kernel<true, false>), dim3(grid_for(nvec, 8)), dim3(kOpsBlock), 0, st, y, bias, nvec,
                               (unsigned)(HW >> 2), (unsigned)C, bits, relu_out);
    } else {
        hipLaunchKernelGGL((bias_add_absmax_kernel<false, false>), dim3(grid_for(n, 8)), dim3(kOpsBlock), 0, st, y, bias, (unsigned)n,
                           (unsigned)HW, (unsigned)C, bits, relu_out);
    }
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_add_absmax_f32(const float* x, const float* y, float* z, size_t n, float* max_inout, float* relu_out,
                                 fq_stream_t stream) {
    using namespace fq;
    if (n == 0) return FQ_OK;
    if (!x || !y || !z || !max_inout) return FQ_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(z) |
         reinterpret_cast<uintptr_t>(relu_out)) & 15u)
        return FQ_ERR_INVALID_ARG;
    const size_t nvec = n >> 2;
    const unsigned tail = (unsigned)(n & 3u);
    if (n * (relu_out ? 16 : 12) > kStreamBytes)
        hipLaunchKernelGGL(add_absmax_kernel<true>, dim3(grid_for(nvec, producer_wg_per_cu(256))), dim3(kOpsBlock), 0, as_stream(stream),
                           reinterpret_cast<const f4v*>(x), reinterpret_cast<const f4v*>(y), reinterpret_cast<f4v*>(z), nvec,
                           x + (nvec << 2), y + (nvec << 2), z + (nvec << 2), tail, reinterpret_cast<unsigned int*>(max_inout), relu_out);
    else
        hipLaunchKernelGGL(add_absmax_kernel<false>, dim3(grid_for(nvec ? nvec : 1, 8)), dim3(kOpsBlock), 0, as_stream(stream),
                           reinterpret_cast<const f4v*>(x), reinterpret_cast<const f4v*>(y), reinterpret_cast<f4v*>(z), nvec,
                           x + (nvec << 2), y + (nvec << 2), z + (nvec << 2), tail, reinterpret_cast<unsigned int*>(max_inout), relu_out);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

static int ops_hist_fast_quotient() {          // FQ_HIST_IEEE_DIV=1 forces the IEEE divide sequence (as in fq_calib.hip)
    static const int v = [] { const char* e = getenv("FQ_HIST_IEEE_DIV"); return (e && e[0] && e[0] != '0') ? 0 : 1; }();
    return v;
}

extern "C" int fq_bias_add_hist_f32(float* y, const float* bias, int N, int C, int HW, const float* interval,
                                    int64_t* hist_row, float* relu_out, fq_stream_t stream) {
    using namespace fq;
    if (N < 0 || C <= 0 || HW <= 0) return FQ_ERR_INVALID_ARG;
    const size_t n = (size_t)N * C * HW;
    if (n == 0) return FQ_OK;
    if (!y || !bias || !interval || !hist_row) return FQ_ERR_INVALID_ARG;
    if (n >= 0xffffffffULL) return FQ_ERR_UNSUPPORTED;               // 32-bit element index inside the kernel
    hipStream_t st = as_stream(stream);
    unsigned long long* h = reinterpret_cast<unsigned long long*>(hist_row);
    if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(relu_out)) & 15u) == 0) {
        const unsigned nvec = (unsigned)(n >> 2);
        if (n * (relu_out ? 12 : 8) > kStreamBytes)
            hipLaunchKernelGGL((bias_add_hist_kernel<true, true>), dim3(grid_for(nvec, producer_wg_per_cu(8))), dim3(kOpsBlock), 0, st, y, bias,
                               nvec, (unsigned)(HW >> 2), (unsigned)C, interval, h, relu_out, ops_hist_fast_quotient());
        else
            hipLaunchKernelGGL((bias_add_hist_kernel<true, false>), dim3(grid_for(nvec, 2)), dim3(kOpsBlock), 0, st, y, bias, nvec,
                               (unsigned)(HW >> 2), (unsigned)C, interval, h, relu_out, ops_hist_fast_quotient());
    } else {
        hipLaunchKernelGGL((bias_add_hist_kernel<false, false>), dim3(grid_for(n, 2)), dim3(kOpsBlock), 0, st, y, bias, (unsigned)n,
                           (unsigned)HW, (unsigned)C, interval, h, relu_out, ops_hist_fast_quotient());
    }
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_add_hist_f32(const float* x, const float* y, float* z, size_t n, const float* interval, int64_t* hist_row,
                               float* relu_out, fq_stream_t stream) {
    using namespace fq;
    if (n == 0) return FQ_OK;
    if (!x || !y || !z || !interval || !hist_row) return FQ_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(z) |
         reinterpret_cast<uintptr_t>(relu_out)) & 15u)
        return FQ_ERR_INVALID_ARG;
    const size_t nvec = n >> 2;
    const unsigned tail = (unsigned)(n & 3u);
    if (n * (relu_out ? 16 : 12) > kStreamBytes)
        hipLaunchKernelGGL(add_hist_kernel<true>, dim3(grid_for(nvec, producer_wg_per_cu(4))), dim3(kOpsBlock), 0, as_stream(stream),
                           reinterpret_cast<const f4v*>(x), reinterpret_cast<const f4v*>(y), reinterpret_cast<f4v*>(z), nvec,
                           x + (nvec << 2), y + (nvec << 2), z + (nvec << 2), tail, interval,
                           reinterpret_cast<unsigned long long*>(hist_row), relu_out, ops_hist_fast_quotient());
    else
        hipLaunchKernelGGL(add_hist_kernel<false>, dim3(grid_for(nvec ? nvec : 1, 2)), dim3(kOpsBlock), 0, as_stream(stream),
                           reinterpret_cast<const f4v*>(x), reinterpret_cast<const f4v*>(y), reinterpret_cast<f4v*>(z), nvec,
                           x + (nvec << 2), y + (nvec << 2), z + (nvec << 2), tail, interval,
                           reinterpret_cast<unsigned long long*>(hist_row), relu_out, ops_hist_fast_quotient());
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_quantize_param_i32(const float* w, int32_t* q, size_t n, int bit, fq_stream_t stream) {
    if (bit < -120 || bit > 120) return FQ_ERR_INVALID_ARG;
    if (n == 0) return FQ_OK;
    if (!w || !q) return FQ_ERR_INVALID_ARG;
    hipLaunchKernelGGL(quantize_param_i32_kernel, dim3(grid_for(n)), dim3(kOpsBlock), 0, as_stream(stream), w, q, n,
                       ldexpf(1.0f, bit));
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}
