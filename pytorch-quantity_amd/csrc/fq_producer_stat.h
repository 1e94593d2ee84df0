// fq_producer_stat.h -- what a producer kernel does with each output value while it is still in registers: the running
// abs-max of calibration pass 1 (distribution_collector.py:70-78) or the 2048-bin histogram of pass 2
// (distribution_collector.py:127-135).  Shared by the elementwise producers (fq_ops.hip) and the fp32 1x1 convolution
// (fq_conv1x1_f32.hip).
#pragma once
#include "fq_common.h"
#include "fq_hist_bin.h"

namespace fq {

// torch's clamp_min(x, 0): NaN stays NaN.  "not (v <= 0)" is true for v > 0 AND for NaN: one v_cmp_nle_f32 + one v_cndmask
// (the obvious  v > 0 ? v : (v != v ? v : 0)  is two compares, a scalar or and the select -- and every vector instruction of an
// epilogue is time the matrix pipe does not get)
__device__ __forceinline__ float relu_like_torch(float v) { return !(v <= 0.0f) ? v : 0.0f; }

struct MaxStat {
    float m = 0.0f;
    __device__ __forceinline__ void add(float v) { m = fmaxf(m, fabsf(v)); }                  // fmaxf drops NaN
};

// TestConv / TestLinear (new_quantity_op.py:283-292, :248-256): QuanDequan(bit) of the convolution's value on its way out
// of the accumulator -- not a statistic, a map: the stored value is map(v).  Bit for bit fq_quandequan_f32's expression.
struct QdStat {
    float scale, inv, lo, hi;                                 // 2^bit, 2^-bit, the integer range of the bit width
    __device__ __forceinline__ void add(float) {}
    __device__ __forceinline__ float map(float v) const {
        const float q = rintf(v * scale);
        return (q < lo ? lo : (q > hi ? hi : q)) * inv;       // NaN fails both compares and passes through
    }
};
template <typename S> __device__ __forceinline__ float stat_map(const S&, float v) { return v; }
__device__ __forceinline__ float stat_map(const QdStat& s, float v) { return s.map(v); }

template <bool kFast>
struct HistStat {
    unsigned int* bins;                                       // 2048 LDS counters of this workgroup
    unsigned int* park;                                       // exact zeros are not counted: a per-lane scratch slot
    float iv, yr;
    __device__ __forceinline__ void add(float v) { atomicAdd((v != 0.0f) ? (bins + bin_of<kFast>(v, iv, yr)) : park, 1u); }
};

// wave maximum -> LDS -> one atomicMax per workgroup on the non-negative float's bit pattern, and only when it can raise it
template <int kThreads>
__device__ __forceinline__ void publish_max(float m, unsigned int* __restrict__ max_bits) {
    __shared__ float s_wave[kThreads / kWave];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) s_wave[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kThreads / kWave; ++w) m = fmaxf(m, s_wave[w]);
        const unsigned int bits = __float_as_uint(m);
        if (bits > __hip_atomic_load(max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_bits, bits);
    }
}

template <int kThreads>
__device__ __forceinline__ void hist_flush(unsigned int* s_bins, unsigned long long* __restrict__ dst) {
    __syncthreads();
    for (int b = threadIdx.x; b < FQ_BINS; b += kThreads) {
        const unsigned int c = s_bins[b];
        if (c) atomicAdd(dst + b, (unsigned long long)c);
    }
}

}  // namespace fq
