// fq_resident.h -- pieces shared by the resident-activation kernels (fq_resident.hip) and the conv epilogue
// that fuses the residual add (fq_conv_i8.hip).
#pragma once
#include "fq_common.h"

namespace fq {

typedef int v4i_r __attribute__((ext_vector_type(4)));

struct AddResParams {
    float sx, sy;          // 2^-gx, 2^-gy: integer -> value
    float lo, hi;          // NewAdd's Sp range; lo = 0 when the following ReLU is fused
    float s_wide;          // 2^g_out   (exact value -> int16)
    float s_narrow;        // 2^ib      (next layers' Quantity)
};

template <typename T> struct Vec16;                       // 16 consecutive channels of one pixel
template <> struct Vec16<int8_t> {
    v4i_r a;
    __device__ __forceinline__ void load(const int8_t* p) { a = *reinterpret_cast<const v4i_r*>(p); }
    __device__ __forceinline__ float get(int i) const { return (float)(int)(int8_t)(((unsigned)a[i >> 2]) >> (8 * (i & 3))); }
};
template <> struct Vec16<int16_t> {
    v4i_r a, b;
    __device__ __forceinline__ void load(const int16_t* p) {
        a = *reinterpret_cast<const v4i_r*>(p);
        b = *reinterpret_cast<const v4i_r*>(p + 8);
    }
    __device__ __forceinline__ float get(int i) const {
        const unsigned d = (unsigned)(i < 8 ? a[(i & 7) >> 1] : b[(i & 7) >> 1]);
        return (float)(int)(int16_t)(d >> (16 * (i & 1)));
    }
};

// NewAdd on 16 resident elements (new_quantity_op.py:171-174 + the ReLU and Quantity that follow it):
//   s = clamp(x * 2^-gx + y * 2^-gy, lo, hi)            the reference's fp32 expression, exact here
//   wide[i]   = (int16) (s * 2^g_out)                    the exact sum, for the next residual add
//   narrow[i] = (int8) clamp(rint(s * 2^ib), -128, 127)  what the next conv's Quantity(ib) computes
template <typename VX, typename VY>
__device__ __forceinline__ void add_resident_16(const VX& vx, const VY& vy, int16_t* __restrict__ wide, int8_t* __restrict__ narrow,
                                                const AddResParams& p) {
    float s[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = vx.get(e) * p.sx + vy.get(e) * p.sy;
        s[e] = __builtin_amdgcn_fmed3f(v, p.lo, p.hi);                // integers scaled by 2^k: never NaN
    }
    if (wide) {
        v4i_r o0, o1;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const unsigned lo16 = (unsigned)(int)(s[2 * d] * p.s_wide) & 0xffffu;
            const unsigned hi16 = (unsigned)(int)(s[2 * d + 1] * p.s_wide) << 16;
            if (d < 4) o0[d] = (int)(lo16 | hi16); else o1[d - 4] = (int)(lo16 | hi16);
        }
        *reinterpret_cast<v4i_r*>(wide) = o0;
        *reinterpret_cast<v4i_r*>(wide + 8) = o1;
    }
    if (narrow) {
        v4i_r o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned w = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = __builtin_amdgcn_fmed3f(rintf(s[4 * d + e] * p.s_narrow), -128.0f, 127.0f);
                w |= ((unsigned)(int)q & 0xffu) << (8 * e);
            }
            o[d] = (int)w;
        }
        *reinterpret_cast<v4i_r*>(narrow) = o;
    }
}

// host: parameters of one resident add; FQ_OK or an error code (ranges, exact-sum grid)
inline int make_add_params(int gx, int gy, int g_wide, bool want_wide, int ib, int relu, AddResParams* p) {
    if (gx < -16 || gx > 16 || gy < -16 || gy > 16 || ib < -16 || ib > 16) return FQ_ERR_INVALID_ARG;
    if (want_wide) {
        // the exact sum must fit: grid max(0, gx, gy), |s| <= 128  =>  |S| <= 2^(7 + g) <= 2^15
        const int g_need = gx > gy ? (gx > 0 ? gx : 0) : (gy > 0 ? gy : 0);
        if (g_wide != g_need || g_wide > 8) return FQ_ERR_UNSUPPORTED;
    }
    p->sx = ldexpf(1.0f, -gx); p->sy = ldexpf(1.0f, -gy);
    p->lo = relu ? 0.0f : -128.0f; p->hi = 127.0f;
    p->s_wide = ldexpf(1.0f, want_wide ? g_wide : 0); p->s_narrow = ldexpf(1.0f, ib);
    return FQ_OK;
}

}  // namespace fq
