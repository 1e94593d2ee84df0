// fq_int_tail.h -- device helpers shared by the integer convolution kernels (fq_conv_i8.hip, fq_stem.hip):
// Quantity on one element, the RightShift + BiasAdd + Sp tail in integer arithmetic, byte packing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Quantity(ib) of one value as an int8 bit pattern (new_quantity_op.py:52-58); scale = 2^ib
__device__ __forceinline__ unsigned q8(float v, float scale) {
    float q = rintf(v * scale);
    q = q < -128.0f ? -128.0f : (q > 127.0f ? 127.0f : q);          // NaN stays NaN; the cast gives 0
    return (unsigned)(uint8_t)(int8_t)(int)q;
}

// clamp(v, lo, hi), lo <= hi, in one instruction.  (hipcc checks an asm statement's outputs against the "16-byte store,
// then a write of its data registers" hazard like any other instruction's; what it does NOT pad on gfx950 is that hazard
// behind a buffer store whose scalar offset is a register -- see fq_conv1x1_i8.hip, which keeps that offset an immediate.)
__device__ __forceinline__ int med3_i32(int v, int lo, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

// RightShift(rs) -> BiasAdd -> Sp on an int32 accumulator, 6 vector instructions instead of the 11 of the fp32
// chain.  Valid for 1 <= rs <= 16 and |acc| + 2^15 < 2^31 (checked on the host):
//   trunc(v + copysign(0.5, v)), v = acc * 2^-rs, is round-half-away = (acc + 2^(rs-1) - (acc < 0)) >> rs
//   with an arithmetic shift; below |acc| < 2^24 every fp32 step of the reference is exact, and from
//   2^24 on both forms are far outside [-128, 127] (|v| >= 2^8) and saturate to the same bound.
// P supplies rs, half_rs = 2^(rs-1), the RightShift range [ilo, ihi] and the Sp range [slo, shi] (slo = 0 with a fused ReLU).
template <typename P>
__device__ __forceinline__ int conv_tail_i(int acc, int qb, const P& p) {
    const int r = (acc + p.half_rs + (acc >> 31)) >> p.rs;
    return med3_i32(med3_i32(r, p.ilo, p.ihi) + qb, p.slo, p.shi);
}

// bytes 0 of four registers -> one dword
__device__ __forceinline__ unsigned pack4(int b0, int b1, int b2, int b3) {
    const unsigned p01 = __builtin_amdgcn_perm((unsigned)b1, (unsigned)b0, 0x0c0c0400u);
    const unsigned p23 = __builtin_amdgcn_perm((unsigned)b3, (unsigned)b2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

}  // namespace fq
