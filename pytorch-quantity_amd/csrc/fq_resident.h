// fq_resident.h -- pieces shared by the resident-activation kernels (fq_resident.hip) and the conv epilogue
// that fuses the residual add (fq_conv_i8.hip).
#pragma once
#include <cstdlib>
#include <type_traits>
#include "fq_common.h"
#include "fq_int_tail.h"

namespace fq {

typedef int v4i_r __attribute__((ext_vector_type(4)));

struct AddResParams {
    float sx, sy;          // 2^-gx, 2^-gy: integer -> value
    float lo, hi;          // NewAdd's Sp range; lo = 0 when the following ReLU is fused
    float s_wide;          // 2^g_out   (exact value -> int16)
    float s_narrow;        // 2^ib      (next layers' Quantity)
    // the same function in integer arithmetic on the grid g = max(0, gx, gy) (int_ok: the host proved the ranges)
    int int_ok;
    int shx, shy;          // g - gx, g - gy: operand -> grid g
    int ilo, ihi;          // Sp range on grid g
    int k, half_m1;        // narrow = S * 2^(ib - g): k = g - ib; k > 0: round-half-even right shift with 2^(k-1) - 1
    // the integer form on PAIRS of int16 (v_pk_*_i16: two elements per instruction) where every intermediate fits 16 bits:
    // pk_ok8 / pk_ok16 for an int8 / int16 operand y; the constants below are the ones above, replicated in both halves
    int pk_ok8, pk_ok16;
    unsigned shx2, shy2, k2, ilo2, ihi2, half2;
};

template <typename T> struct Vec16;                       // 16 consecutive channels of one pixel
template <> struct Vec16<int8_t> {
    v4i_r a;
    __device__ __forceinline__ void load(const int8_t* p) { a = *reinterpret_cast<const v4i_r*>(p); }
    __device__ __forceinline__ int geti(int i) const { return (int)(int8_t)(((unsigned)a[i >> 2]) >> (8 * (i & 3))); }
    __device__ __forceinline__ float get(int i) const { return (float)geti(i); }
};
template <> struct Vec16<int16_t> {
    v4i_r a, b;
    __device__ __forceinline__ void load(const int16_t* p) {
        a = *reinterpret_cast<const v4i_r*>(p);
        b = *reinterpret_cast<const v4i_r*>(p + 8);
    }
    __device__ __forceinline__ int geti(int i) const {
        const unsigned d = (unsigned)(i < 8 ? a[(i & 7) >> 1] : b[(i & 7) >> 1]);
        return (int)(int16_t)(d >> (16 * (i & 1)));
    }
    __device__ __forceinline__ float get(int i) const { return (float)geti(i); }
};

// NewAdd on 16 resident elements (new_quantity_op.py:171-174 + the ReLU and Quantity that follow it):
//   s = clamp(x * 2^-gx + y * 2^-gy, lo, hi)            the reference's fp32 expression, exact here
//   wide[i]   = (int16) (s * 2^g_out)                    the exact sum, for the next residual add
//   narrow[i] = (int8) clamp(rint(s * 2^ib), -128, 127)  what the next conv's Quantity(ib) computes
//
// Integer form (p.int_ok): every value above is an integer multiple of 2^-g, g = max(0, gx, gy), and small enough
// that the fp32 chain is exact, so with S = s * 2^g
//   S = med3((x << (g - gx)) + (y << (g - gy)), lo * 2^g, hi * 2^g),   wide = S,
//   narrow = med3(k > 0 ? (S + 2^(k-1) - 1 + ((S >> k) & 1)) >> k : S << -k, -128, 127),  k = g - ib
// (round-half-to-even of S / 2^k is exactly what rint does on the exact quotient) -- 10 vector instructions per
// element instead of 17; the fused conv + add epilogue of the 1x1 expand layers is bound by exactly these.
struct Add16Out {                                         // 16 elements: the exact sum as 16 int16, its re-quantisation as 16 int8
    v4i_r w0, w1, n;
};

template <bool kShiftRight, typename VX, typename VY>
__device__ __forceinline__ Add16Out add_resident_16_int(const VX& vx, const VY& vy, bool want_wide, bool want_narrow,
                                                        const AddResParams& p) {
    Add16Out o = {};
    int s[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = med3_i32((vx.geti(e) << p.shx) + (vy.geti(e) << p.shy), p.ilo, p.ihi);
    if (want_wide) {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const int w = (int)__builtin_amdgcn_perm((unsigned)s[2 * d + 1], (unsigned)s[2 * d], 0x05040100u);   // low halves
            if (d < 4) o.w0[d] = w; else o.w1[d - 4] = w;
        }
    }
    if (want_narrow) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            int q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int S = s[4 * d + e];
                const int r = kShiftRight ? (S + p.half_m1 + ((S >> p.k) & 1)) >> p.k : S << -p.k;
                q[e] = med3_i32(r, -128, 127);
            }
            o.n[d] = (int)pack4(q[0], q[1], q[2], q[3]);
        }
    }
    return o;
}

// ---- the same integers two at a time.  With g <= 8 the clamped sum S lies in [-128 * 2^g, 127 * 2^g], inside int16; with
// shx, shy <= 8 so do the shifted operands of an int8 x / y (|.| <= 128 * 256), and an int16 y is taken unshifted (shy = 0:
// the residual chain on one grid, the usual case); their sum is formed with a SATURATING add (v_pk_add_i16 clamp), and
// saturation to [-32768, 32767] followed by the clamp to [ilo, ihi] inside that range equals the clamp of the exact sum.
// For 1 <= k <= 8 the rounding term S + 2^(k-1) - 1 + bit stays below 32512 + 128 and above -32768: no wrap.  17 vector
// instructions per 4 elements (unpacking the int8 operand included) instead of 45; the int16 sum leaves as it stands.
typedef short v2s_r __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s_r pk_of(unsigned u) { return __builtin_bit_cast(v2s_r, u); }
__device__ __forceinline__ unsigned u_of(v2s_r v) { return __builtin_bit_cast(unsigned, v); }
// bytes (0, 1) / (2, 3) of a dword as two sign-extended int16
__device__ __forceinline__ v2s_r pk_bytes_lo(unsigned d) { return pk_of(__builtin_amdgcn_perm(0u, d, 0x010c000cu)) >> (short)8; }
__device__ __forceinline__ v2s_r pk_bytes_hi(unsigned d) { return pk_of(__builtin_amdgcn_perm(0u, d, 0x030c020cu)) >> (short)8; }

template <typename VY>
__device__ __forceinline__ Add16Out add_resident_16_pk(const Vec16<int8_t>& vx, const VY& vy, bool want_wide, bool want_narrow,
                                                       const AddResParams& p) {
    Add16Out o = {};
    const v2s_r sx = pk_of(p.shx2), sy = pk_of(p.shy2), kk = pk_of(p.k2), lo = pk_of(p.ilo2), hi = pk_of(p.ihi2);
    v2s_r s[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {                         // dword d of x: elements 4 d .. 4 d + 3
        const unsigned xb = (unsigned)vx.a[d];
        v2s_r y0, y1;
        if constexpr (std::is_same<VY, Vec16<int16_t>>::value) {
            y0 = pk_of((unsigned)(d < 2 ? vy.a[2 * d] : vy.b[2 * d - 4]));
            y1 = pk_of((unsigned)(d < 2 ? vy.a[2 * d + 1] : vy.b[2 * d - 3]));
        } else {
            y0 = pk_bytes_lo((unsigned)vy.a[d]) << sy;
            y1 = pk_bytes_hi((unsigned)vy.a[d]) << sy;
        }
        const v2s_r a0 = __builtin_elementwise_add_sat(pk_bytes_lo(xb) << sx, y0);
        const v2s_r a1 = __builtin_elementwise_add_sat(pk_bytes_hi(xb) << sx, y1);
        s[2 * d] = __builtin_elementwise_min(__builtin_elementwise_max(a0, lo), hi);
        s[2 * d + 1] = __builtin_elementwise_min(__builtin_elementwise_max(a1, lo), hi);
    }
    if (want_wide) {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            if (d < 4) o.w0[d] = (int)u_of(s[d]); else o.w1[d - 4] = (int)u_of(s[d]);
        }
    }
    if (want_narrow) {
        const v2s_r one = {1, 1}, m128 = {-128, -128}, p127 = {127, 127}, hm1 = pk_of(p.half2);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            v2s_r t0 = (s[2 * d] + hm1 + ((s[2 * d] >> kk) & one)) >> kk;
            v2s_r t1 = (s[2 * d + 1] + hm1 + ((s[2 * d + 1] >> kk) & one)) >> kk;
            t0 = __builtin_elementwise_min(__builtin_elementwise_max(t0, m128), p127);
            t1 = __builtin_elementwise_min(__builtin_elementwise_max(t1, m128), p127);
            o.n[d] = (int)__builtin_amdgcn_perm(u_of(t1), u_of(t0), 0x06040200u);      // the low bytes of the four int16
        }
    }
    return o;
}

// The sum of 16 elements in registers (the callers store it: plain pointers in fq_resident.hip / the general conv epilogue,
// buffer stores in fq_conv1x1_i8.hip)
template <typename VX, typename VY>
__device__ __forceinline__ Add16Out add_resident_16_regs(const VX& vx, const VY& vy, bool want_wide, bool want_narrow,
                                                         const AddResParams& p) {
    if constexpr (std::is_same<VX, Vec16<int8_t>>::value) {
        if (std::is_same<VY, Vec16<int16_t>>::value ? p.pk_ok16 : p.pk_ok8)     // uniform
            return add_resident_16_pk(vx, vy, want_wide, want_narrow, p);
    }
    if (p.int_ok) {                                       // uniform
        if (p.k > 0) return add_resident_16_int<true>(vx, vy, want_wide, want_narrow, p);
        return add_resident_16_int<false>(vx, vy, want_wide, want_narrow, p);
    }
    Add16Out o = {};
    float s[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = vx.get(e) * p.sx + vy.get(e) * p.sy;
        s[e] = __builtin_amdgcn_fmed3f(v, p.lo, p.hi);                // integers scaled by 2^k: never NaN
    }
    if (want_wide) {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const unsigned lo16 = (unsigned)(int)(s[2 * d] * p.s_wide) & 0xffffu;
            const unsigned hi16 = (unsigned)(int)(s[2 * d + 1] * p.s_wide) << 16;
            if (d < 4) o.w0[d] = (int)(lo16 | hi16); else o.w1[d - 4] = (int)(lo16 | hi16);
        }
    }
    if (want_narrow) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned w = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = __builtin_amdgcn_fmed3f(rintf(s[4 * d + e] * p.s_narrow), -128.0f, 127.0f);
                w |= ((unsigned)(int)q & 0xffu) << (8 * e);
            }
            o.n[d] = (int)w;
        }
    }
    return o;
}

template <typename VX, typename VY>
__device__ __forceinline__ void add_resident_16(const VX& vx, const VY& vy, int16_t* __restrict__ wide, int8_t* __restrict__ narrow,
                                                const AddResParams& p) {
    const Add16Out o = add_resident_16_regs(vx, vy, wide != nullptr, narrow != nullptr, p);
    if (wide) {
        *reinterpret_cast<v4i_r*>(wide) = o.w0;
        *reinterpret_cast<v4i_r*>(wide + 8) = o.w1;
    }
    if (narrow) *reinterpret_cast<v4i_r*>(narrow) = o.n;
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
    // integer form: operands |x|, |y| <= 2^15 shifted by at most 12, the clamped sum |S| <= 2^(7+g) <= 2^15 shifted left by
    // at most 15 -- nothing leaves int32.  FQ_ADD_FLOAT=1 keeps the fp32 chain (A/B timing).
    static const bool force_float = [] { const char* e = getenv("FQ_ADD_FLOAT"); return e && e[0] && e[0] != '0'; }();
    const int g = gx > gy ? (gx > 0 ? gx : 0) : (gy > 0 ? gy : 0);
    p->shx = g - gx; p->shy = g - gy;
    p->k = g - ib;
    p->int_ok = !force_float && g <= 8 && p->shx <= 12 && p->shy <= 12 && p->k >= -15 && p->k <= 24;
    p->half_m1 = p->k > 0 ? (1 << (p->k - 1)) - 1 : 0;
    p->ilo = relu ? 0 : -(128 << g); p->ihi = 127 << g;
    if (!p->int_ok) { p->shx = p->shy = 0; p->k = 0; }
    // packed int16 form: every intermediate inside 16 bits (see add_resident_16_pk).  FQ_ADD_PACKED=0 keeps the 32-bit form.
    static const bool no_pk = [] { const char* e = getenv("FQ_ADD_PACKED"); return e && e[0] == '0'; }();
    const bool pk = p->int_ok && !no_pk && p->k >= 1 && p->k <= 8 && p->shx >= 0 && p->shx <= 8;
    p->pk_ok8 = pk && p->shy >= 0 && p->shy <= 8;
    p->pk_ok16 = pk && p->shy == 0;
    auto rep = [](int v) { return ((unsigned)v & 0xffffu) * 0x10001u; };
    p->shx2 = rep(p->shx); p->shy2 = rep(p->shy); p->k2 = rep(p->k); p->ilo2 = rep(p->ilo); p->ihi2 = rep(p->ihi);
    p->half2 = rep(p->half_m1);
    return FQ_OK;
}

}  // namespace fq
