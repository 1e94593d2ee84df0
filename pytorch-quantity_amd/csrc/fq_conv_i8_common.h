// fq_conv_i8_common.h -- what the integer convolution kernels share (fq_conv_i8.hip: the general implicit GEMM;
// fq_conv1x1_i8.hip: the streaming 1x1 form with stationary weights): launch parameters, buffer descriptors, the LDS-DMA
// instruction, the LDS swizzle.
#pragma once
#include <cstdlib>

#include "fq_resident.h"
#include "fq_int_tail.h"

namespace fq {

constexpr int kConvBlock = 256;
constexpr int kTP = 128;                 // pixels per workgroup tile

struct ConvParams {
    int N, H, W, C;                      // input NHWC, C % 16 == 0
    int K, R, S;                         // weights [K][R][S][C]
    int P, Q;                            // output spatial
    int stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int M;                               // N * P * Q
    int chunks;                          // R * S * C / 16   (16-byte units of the reduction axis)
    int c16;                             // C / 16
    float inv_rs, inv_ob, lo, hi;        // 2^-rs, 2^-ob, clamp range (lo = 0 when a ReLU is fused)
    int ilo, ihi;
    int Kpad;                            // channel stride of the int8 NHWC output (>= K, multiple of 16)
    unsigned out_elems;                  // M * Kpad: elements of the int8 output / the fused add's residual and sum tensors
    unsigned x_bytes;                    // N * H * W * C: num_records of the activation buffer descriptor
    int rs, half_rs, slo, shi;           // integer tail: shift, 2^(rs-1), Sp range; rs = 0 selects the fp32 tail
    unsigned w_bytes;                    // K * R * S * C: num_records of the weight buffer descriptor
    // fused residual add (kOutAdd): the conv output is operand x of NewAdd, `res` is operand y
    const void* res;                     // int8 / int16 NHWC [N][P][Q][Kpad], same layout as the int8 output
    int res_bytes;
    int16_t* wide;                       // exact int16 sum (may be null); the int8 output pointer receives `narrow`
    AddResParams ap;
    // XCD-aware workgroup order (0 = plain 2-D grid): see conv_tile_of()
    int xcd_kt, tiles_m;
};

// Activation loads are buffer loads: an out-of-image tap (zero padding) or a chunk past the end of the
// reduction axis gets the offset kOutOfRange, which is beyond num_records of the buffer descriptor, and the
// hardware returns zeros -- no select, no branch, no zero page.  (A first version selected between the
// activation pointer and a __device__ const zero page: the const object lives in the constant address
// space, every operand load degraded to flat_load, and the compiler then waited for vmcnt(0) -- the whole
// weight-tile latency -- before it issued the activation loads of each K-step.)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned kOutOfRange = 0x80000000u;             // tensors on this path are < 2^31 bytes
// LDS-DMA: 64 lanes x 16 bytes from a buffer straight into LDS at lds_base + 16 * lane.  Deliberately inline
// asm rather than __builtin_amdgcn_raw_ptr_buffer_load_lds: hipcc treats the builtin as an LDS store that may
// alias every later ds_read and inserts s_waitcnt vmcnt(0) right behind it -- in the K loop that serialised
// the next step's loads with this step's MFMAs completely (found in the ISA, not in the timings of a
// trace build whose stamps perturb the schedule).  The asm has no memory clobber; ordering is explicit:
// every wave waits vmcnt(0) and passes a workgroup barrier before anyone reads the slot that was filled.
typedef int rsrc_words __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_words make_rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    rsrc_words r = {(int)(unsigned)(a & 0xffffffffu), (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    return r;
}
// (m0 is what the instruction reads its LDS base from; naming it in the clobber list is the point, and clang's
// "clobber list contains reserved registers" note about it is silenced here only)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma_to_lds(rsrc_words rsrc, unsigned lds_base, unsigned voffset, int soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_base), "v"(voffset), "s"(rsrc), "s"(soffset) : "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ unsigned lds_offset(const void* shared_ptr) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)shared_ptr;
}

__device__ __forceinline__ v4i load_act(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return (v4i)__builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}

// 128-byte LDS rows hold 8 16-byte chunks; chunk ^= (row >> 1) & 7 makes every ds_read_b128 lane
// group ({0-3,12-15,20-27}, ...) touch 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// kOut: bit 0 = fp32 NCHW output y (the module-boundary format), bit 1 = int8 NHWC output q (the
// resident hand-off to the next integer layer: 1 byte per element instead of 4 written + 4 read + 1).
constexpr int kOutF32 = 1, kOutI8 = 2, kOutAdd = 4;      // kOutAdd: with kOutI8, NewAdd fused into the store
constexpr int kOutResEarly = 8;                          // with kOutAdd (register-staged kernel): the residual is requested behind the first K-step's operand loads

// fq_conv1x1_i8.hip: true when the streaming kernel took the launch (p is complete except xcd_kt / tiles_m)
bool launch_conv1x1_stream(hipStream_t st, const int8_t* x, const int8_t* w, const float* qbias, float* y, int8_t* q,
                           const ConvParams& p);

}  // namespace fq
