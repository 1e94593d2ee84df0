// fq_hist_bin.h -- the bin arithmetic of the 2048-bin |x| histogram (distribution_collector.py:131), shared by the
// statistics kernels (fq_calib.hip) and by the producer kernels that histogram their own output (fq_ops.hip).
#pragma once
#include "fq_common.h"

namespace fq {

// Bin of one element.  kFast = false: the IEEE divide sequence.  kFast = true: the 3-instruction
// quotient  q0 = a*y, r = fma(-q0, iv, a), q = fma(r, y, q0)  with y = RN(1/iv); it equals the
// correctly rounded a/iv for every fp32 significand pair (checked exhaustively on the GPU,
// scripts/verify_fastdiv.hip, result under profiles/), and overflow / inf / nan fall through to the
// same "last bin" as the IEEE path because  q < 2048  is false for inf and nan.
template <bool kFast, int kBins = FQ_BINS>
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
    return (q < (float)kBins) ? (int)q : (kBins - 1);   // >= INTERVAL_NUM (2048), inf, nan -> last bin
}

// The exhaustive proof of the fast quotient covers every significand pair but assumes that no intermediate
// under/overflows.  For 2^-60 <= iv <= 2^60 that holds for every element that can land above bin 0 (|x| >= iv >= 2^-60
// keeps the fma residual normal; quotients that overflow become inf/nan and fall into the last bin exactly like the
// IEEE path).  Calibration intervals are max/2048 + 1e-12, far inside that range; anything else takes the IEEE divide.
__device__ __forceinline__ bool fast_quotient_ok(float iv) {
    const unsigned int ivb = __float_as_uint(iv);
    return ivb >= 0x21800000u && ivb <= 0x5d800000u;
}

}  // namespace fq
