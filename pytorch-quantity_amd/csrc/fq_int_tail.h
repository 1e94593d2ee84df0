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

// The same tail in FOUR vector instructions, from three per-channel constants instead of one (tail_consts below):
//   * floor commutes with adding a multiple of 2^rs:  ((acc + half + sign) >> rs) + qb = (acc + half + (qb << rs) + sign) >> rs,
//     so the bias rides in the rounding constant  B = half_rs + (qb << rs)   (qb clamped as described below: inside int32 with the
//     host's bound on |acc|);
//   * a clamp of a clamp is a clamp:  med3(med3(r, ilo, ihi) + qb, slo, shi) = med3(r + qb, lo, hi)  with
//     lo = med3(ilo + qb, slo, shi), hi = med3(ihi + qb, slo, shi)  (a bias beyond the output range makes lo = hi).
// Same integer for every acc the six-instruction form is valid for.  Where the tail is what a kernel's vector pipe is busy with
// (fq_block_tail_i8: two or three tails per output value) the two instructions are time; in the stem they were not.
// The bias is only "integer valued" by contract, not bounded: it is first brought into [slo - ihi, shi - ilo] -- beyond that range
// the output is slo or shi whatever the accumulator holds (the smallest r + qb is already above shi, or the largest below slo), so
// the clamped bias gives the same integer -- and then |qb| << rs stays below 2^(bits + 1 + rs), which the host adds to its bound
// on |acc| (fq_conv_i8.hip, int_tail).
struct TailK { int B, lo, hi; };
template <typename P>
__device__ __forceinline__ TailK tail_consts(int qb, const P& p) {
    TailK k;
    const int qlo = p.slo - p.ihi, qhi = p.shi - p.ilo;
    qb = qb < qlo ? qlo : (qb > qhi ? qhi : qb);
    k.B = p.half_rs + (qb << p.rs);
    const int a = p.ilo + qb, b = p.ihi + qb;
    k.lo = a < p.slo ? p.slo : (a > p.shi ? p.shi : a);
    k.hi = b < p.slo ? p.slo : (b > p.shi ? p.shi : b);
    return k;
}
__device__ __forceinline__ int conv_tail_k(int acc, int B, int lo, int hi, int rs) {
    return med3_i32((acc + B + (acc >> 31)) >> rs, lo, hi);
}

// bytes 0 of four registers -> one dword
__device__ __forceinline__ unsigned pack4(int b0, int b1, int b2, int b3) {
    const unsigned p01 = __builtin_amdgcn_perm((unsigned)b1, (unsigned)b0, 0x0c0c0400u);
    const unsigned p23 = __builtin_amdgcn_perm((unsigned)b3, (unsigned)b2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

}  // namespace fq
