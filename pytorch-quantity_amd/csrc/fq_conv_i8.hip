// fq_conv_i8.hip -- the integer contraction of NewConv2d / NewLinear on the gfx950 matrix cores.
//
// Reference: quantity/common/quantity/new_quantity_op.py:124-133 (NewConv2d.forward) and :197-205
// (NewLinear.forward) run Quantity -> fp32 conv over integer-valued tensors -> RightShift -> BiasAdd
// -> Sp -> DeQuantity as 7+ separate passes.  Here:
//
//   quantize_i8_nhwc_kernel   x fp32 NCHW -> int8 NHWC (clamp(rint(x * 2^ib))), channels zero-padded
//                             to a multiple of 16.  HBM bound: 4 B read + 1 B written per element.
//   conv2d_i8_kernel          implicit GEMM  D[k_out][pixel] = sum_{r,s,c} W[k][r][s][c] * X[n][ih][iw][c]
//                             on v_mfma_i32_32x32x32_i8 (int32 accumulation: exact, where the
//                             reference's fp32 conv is exact only below 2^24), with the whole tail
//                             (shift, round half away, bias, saturate, dequantise) fused into the
//                             epilogue, written straight to fp32 NCHW.  This is the one place on
//                             the path that is a dense contraction, hence the one MFMA kernel.
//
// Tiling (64-wide wavefronts): workgroup = 256 threads = 4 waves side by side on the pixel axis,
// tile = TK output channels x 128 pixels, K-step = 128 bytes of the (r, s, c) reduction axis = 4 MFMA
// k-sub-steps.  Pixels sit on the MFMA lane index, so each accumulator register is 32 consecutive
// pixels of one output channel (128-byte coalesced NCHW stores, no shuffle) and the activation
// operand is loaded by each lane straight from the NHWC tensor (16 contiguous bytes = 16 channels
// of one tap): the streamed operand never touches LDS.  Only the weight tile, shared by the four
// waves, is staged global -> registers -> LDS (double buffered); its 128-byte rows are XOR-swizzled
// (chunk ^= (row >> 1) & 7) so that ds_read_b128 fragment reads are bank-conflict free.
//
//   conv2d_i8_dma_kernel      the same contraction for deep reductions (C % 128 == 0, >= 8 K-steps): both operand
//                             tiles arrive by buffer_load ... lds (LDS-DMA), ring of 2 or 3 K-steps.
//   epilogues                 fp32 NCHW, int8 NHWC (the resident hand-off), both, or the residual add fused in
//                             (fq_resident.h); integer tail where the host proved it equal to the fp32 chain.
//   workgroup order           1-D grid decoded so that all output-channel tiles of a pixel tile run on one XCD
//                             (conv_tile_of): the activation tile is fetched into that XCD's L2 once.
//   quantize_i8_unfold_w*     stem layers outside fq_conv2d_i8_stem's limits (fq_stem.hip): kernel width folded
//                             into the channel axis.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "fq_conv_i8_common.h"

namespace fq {

// Debug build only (-DFQ_CONV_TRACE, libfq_hip_trace.so): s_memtime stamps of one wave of a few workgroups,
// read back with fq_debug_read_trace (scripts/conv_trace.py).  Never part of libfq_hip.so.
#ifdef FQ_CONV_TRACE
__device__ unsigned long long g_trace[8192];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define TR(slot) do { if (trace_on) { unsigned long long t__ = stamp(); if (lane == 0) g_trace[trace_base + (slot)] = t__; } } while (0)
#else
#define TR(slot) do {} while (0)
#endif


// Which (pixel tile, output-channel tile) this workgroup computes.  Workgroups are dealt to the 8 XCDs round robin
// by linear id, and each XCD has its own L2.  With the plain grid (pixel tiles on x) the k-tiles of one pixel tile run
// gridDim.x workgroups apart on whatever XCD that lands on: every one of them fetches the activation tile again from
// beyond its L2.  With xcd_kt = number of k-tiles the 1-D grid is decoded so that all k-tiles of a pixel tile are
// consecutive workgroups of ONE XCD (pixel tile = 8 * group + xcd): the tile is fetched into that L2 once.
__device__ __forceinline__ bool conv_tile_of(const ConvParams& p, int& bx, int& by) {
    if (p.xcd_kt == 0) { bx = blockIdx.x; by = blockIdx.y; return true; }
    const unsigned id = blockIdx.x, slot = id >> 3;
    by = (int)(slot % (unsigned)p.xcd_kt);
    bx = (int)(slot / (unsigned)p.xcd_kt) * 8 + (int)(id & 7u);
    return bx < p.tiles_m;                                // the grid is padded to a multiple of 8 pixel tiles
}

// RightShift -> BiasAdd -> Sp -> DeQuantity on one accumulator (new_quantity_op.py:127-132)
// The values are integers (never NaN), so the clamps are single v_med3 instructions; v == 0 may round
// with either sign of 0.5 (both truncate to 0), so the half is attached with a sign copy.
// Returns the saturated integer (as a float) that DeQuantity then scales by 2^-ob; that integer is
// also exactly what the NEXT layer's Quantity(ib = ob) would recover from the fp32 value.
__device__ __forceinline__ float conv_tail_int(int acc, float qb, const ConvParams& p) {
    const float v = (float)acc * p.inv_rs;
    const float w = v + __builtin_copysignf(0.5f, v);
    int r = (int)w;                                       // truncates toward zero, saturates
    r = min(max(r, p.ilo), p.ihi);
    return __builtin_amdgcn_fmed3f((float)r + qb, p.lo, p.hi);
}
__device__ __forceinline__ float conv_tail(int acc, float qb, const ConvParams& p) {
    return conv_tail_int(acc, qb, p) * p.inv_ob;
}
// The same tail in integer arithmetic: conv_tail_i (fq_int_tail.h), selected when the host proved it equivalent.

// Position of a 16-byte chunk on the reduction axis: tap (r, s) and 16-channel group cc.
struct RedPos { int cc, fs, fr; };
// delta is 2 (chunks of one lane are two apart): at most two wraps (c16 == 1), done with selects so the
// K loop carries no divergent branch.
__device__ __forceinline__ void red_advance(RedPos& q, int delta, int c16, int S) {
    q.cc += delta;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const bool wrap = q.cc >= c16;
        q.cc -= wrap ? c16 : 0;
        const int fs1 = q.fs + (wrap ? 1 : 0);
        const bool row = fs1 == S;
        q.fs = row ? 0 : fs1;
        q.fr += row ? 1 : 0;
    }
}


// Epilogue shared by the conv kernels.  Accumulator layout (v_mfma_i32_32x32x32_i8): D row = k_out =
// (r&3) + 8*(r>>2) + 4*half, D col = this lane's pixel (wave*32 + lane&31 of the 128-pixel tile).  The tail
// runs in integer arithmetic when the host proved it equivalent (kIntTail), else as the reference's fp32
// chain.  sO: >= kTP * (TK + 16) bytes of LDS that no wave reads any more (the weight buffers).
// The residual operand of a fused NewAdd for one thread of the store layout (16 channels of NJ pixels).  Plain registers handed
// around BY REFERENCE (a first attempt at requesting them early passed a pointer chosen at run time between two such structs,
// which put both into scratch memory and doubled the launch times).
template <int TK>
struct ResRegs {
    static constexpr int NJ = (kTP * (TK / 16)) / kConvBlock;
    v4i_r lo[NJ], hi[NJ];
};
template <int TK>
__device__ __forceinline__ void load_residual(ResRegs<TK>& r, const ConvParams& p, int lane, int wave, int m0, int k0) {
    constexpr int CPP = TK / 16;
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.out_elems * (unsigned)p.res_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < ResRegs<TK>::NJ; ++j) {
        const int idx = lane + 64 * j;
        const int pix = wave * 32 + idx / CPP, ch = idx % CPP;
        const int mm = m0 + pix, kk = k0 + 16 * ch;
        // (element offset in the [M][Kpad] tensors, 32 bits -- M x Kpad < 2^30, host check; 0x40000000 is out of range for the
        //  int8 descriptor and, doubled, for the int16 ones: such loads give zeros, such stores are dropped)
        const unsigned o = (mm < p.M && kk < p.Kpad) ? (unsigned)mm * (unsigned)p.Kpad + (unsigned)kk : 0x40000000u;
        // (both halves always: for an int8 shortcut the second request is out of range -- zeros, no traffic -- so that the struct is
        //  fully defined on every path and stays in registers)
        const bool wide_res = p.res_bytes != 1;
        r.lo[j] = (v4i_r)load_act(rr, wide_res ? 2u * o : o);
        r.hi[j] = (v4i_r)load_act(rr, wide_res ? 2u * o + 16u : kOutOfRange);
    }
}

template <int TK, int kOut, bool kIntTail, bool kStageAliased = true>
__device__ __forceinline__ void conv_epilogue(v16i (&acc)[TK / 32], const ConvParams& p, float* __restrict__ y,
                                              int8_t* __restrict__ q, int8_t* sO, const float* sBias, const int* sBiasI,
                                              const int (&sLH)[2][TK], int m0,
                                              int k0, int n_img, int pq, bool m_ok, ResRegs<TK>& res, int tid_base = 0) {
    constexpr int MT = TK / 32;
    // (tid_base: a 512-thread workgroup runs this once per 256-thread half, each on its own 128-pixel tile and its own sO)
    const int tid = (int)threadIdx.x - tid_base, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int PQ = p.P * p.Q;
    if ((kOut & kOutF32) && m_ok) {
        // For a fixed register the 32 lanes of a half-wave write 32 consecutive pixels of one channel
        // (128 bytes).  One 64-bit base per lane; the 16*MT rows are 32-bit element offsets from it (a
        // K-tile of one image spans at most TK*PQ floats, far below 2^31)
        float* __restrict__ out = y + ((long)n_img * p.K + k0 + 4 * half) * PQ + pq;
        const int kmax = p.K - k0 - 4 * half;             // rows of this lane that exist
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = a * 32 + (r & 3) + 8 * (r >> 2);          // compile-time constant
                if (kl < kmax)
                    out[(unsigned)(kl * PQ)] = kIntTail ? (float)conv_tail_k(acc[a][r], sBiasI[kl + 4 * half], sLH[0][kl + 4 * half],
                                                                             sLH[1][kl + 4 * half], p.rs) * p.inv_ob
                                                        : conv_tail(acc[a][r], sBias[kl + 4 * half], p);
            }
        }
    }
    if (kOut & kOutI8) {
        // int8 NHWC: registers 4g..4g+3 of a tile are 4 consecutive channels = one dword of this lane's
        // pixel.  The dwords go through LDS as [pixel][TK + 16 bytes] and leave as 16-byte stores, TK
        // contiguous bytes per pixel.  Rows k >= K carry zero weights and zero bias, so the channel
        // padding [K, Kpad) is written as zeros.
        constexpr int OS = TK + 16;                       // LDS row stride in bytes (16-byte aligned rows)
        constexpr int CPP = TK / 16;                      // 16-byte chunks per pixel
        constexpr int NJ = (kTP * CPP) / kConvBlock;      // 16-channel groups each thread stores
        // fused NewAdd: the residual groups this thread will need are requested FIRST, so that their latency
        // hides under the tail arithmetic and the LDS transpose below instead of sitting in front of the stores
        // (the operand-fragment registers are dead here, so this costs no occupancy)
        // Store layout: WAVE-LOCAL since round 4 -- item j of a lane is (pixel wave * 32 + (lane + 64 j) / CPP, 16-channel group
        // (lane + 64 j) % CPP), i.e. a wave reads back only the 32 rows of sO it wrote itself.  LDS instructions of one wave execute
        // in order, so the barrier that used to separate the tail's writes from these reads is gone.  (Measured neutral here:
        // 2.831 vs 2.843 ms per 256-image forward, profiles/r04_int8_layer_table_b256*.txt -- the barrier in front of the tail,
        // which the aliasing of sO with the operand tiles needs, still keeps a workgroup's waves in step.)
        static_assert(NJ == ResRegs<TK>::NJ, "store layout");
        if constexpr ((kOut & kOutAdd) != 0 && (kOut & kOutResEarly) == 0) load_residual<TK>(res, p, lane, wave, m0, k0);
        // sO aliases the operand tiles in every kernel but the stationary-weight one: every wave must be done reading them.
        // (kStageAliased = false: sO is the caller's own region, a wave only ever touches its own 32 rows of it -- no barrier)
        if constexpr (kStageAliased) __syncthreads();
        const int prow = (wave * 32 + (lane & 31)) * OS + 4 * half;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int v[4];
                if constexpr (kIntTail) {
                    // the four-instruction tail (fq_int_tail.h): rounding constant with the bias in it + the merged clamp's bounds of
                    // this lane's four consecutive channels, three 16-byte LDS reads
                    const int kl = a * 32 + 8 * g + 4 * half;
                    const v4i cB = *reinterpret_cast<const v4i*>(&sBiasI[kl]), cL = *reinterpret_cast<const v4i*>(&sLH[0][kl]),
                              cH = *reinterpret_cast<const v4i*>(&sLH[1][kl]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = conv_tail_k(acc[a][4 * g + e], cB[e], cL[e], cH[e], p.rs);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (int)conv_tail_int(acc[a][4 * g + e], sBias[a * 32 + e + 8 * g + 4 * half], p);
                }
                *reinterpret_cast<unsigned*>(&sO[prow + a * 32 + 8 * g]) = pack4(v[0], v[1], v[2], v[3]);
                if (kOut & kOutAdd) __builtin_amdgcn_sched_barrier(0);   // four tails at a time beside the residual registers (fq_block_tail_i8.hip)
            }
        }
        // outputs through buffer descriptors (an output nobody wants: zero records, its stores are dropped); the 16-byte stores keep
        // the scalar offset the immediate 0 (the store hazard hipcc does not pad on gfx950)
        const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(q, 0, q ? p.out_elems : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(p.wide, 0, ((kOut & kOutAdd) && p.wide) ? 2u * p.out_elems : 0u, 0x00020000);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int idx = lane + 64 * j;
            const int pix = wave * 32 + idx / CPP, ch = idx % CPP;
            const int mm = m0 + pix, kk = k0 + 16 * ch;
            const unsigned o = (mm < p.M && kk < p.Kpad) ? (unsigned)mm * (unsigned)p.Kpad + (unsigned)kk : 0x40000000u;
            if (kOut & kOutAdd) {
                // NewAdd (+ ReLU + the consumers' Quantity) on the tile while it is in flight: the conv's
                // int8 result never reaches HBM; 16 channels per thread
                Vec16<int8_t> cv;
                cv.a = *reinterpret_cast<const v4i_r*>(&sO[pix * OS + 16 * ch]);
                auto emit = [&](const Add16Out& out) {
                    __builtin_amdgcn_raw_buffer_store_b128((v4u)out.w0, wr, (int)(2u * o), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128((v4u)out.w1, wr, (int)(2u * o + 16u), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128((v4u)out.n, qr, (int)o, 0, 0);
                };
                if (p.res_bytes == 1) {
                    Vec16<int8_t> rv;
                    rv.a = res.lo[j];
                    emit(add_resident_16_regs(cv, rv, p.wide != nullptr, q != nullptr, p.ap));
                } else {
                    Vec16<int16_t> rv;
                    rv.a = res.lo[j]; rv.b = res.hi[j];
                    emit(add_resident_16_regs(cv, rv, p.wide != nullptr, q != nullptr, p.ap));
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4u*>(&sO[pix * OS + 16 * ch]), qr, (int)o, 0, 0);
            }
        }
    }
}

// Workgroup = 4 waves side by side along the pixel axis: wave w owns pixels [32w, 32w+32) of the
// 128-pixel tile and ALL TK output channels (MT = TK/32 accumulator tiles).
//   * activations (MFMA B operand) never touch LDS: lane l holds pixel (l & 31) and loads, per 32-deep
//     sub-step, the 16 contiguous bytes (16 channels of one tap) at chunk 2*ks + (l >> 5) straight
//     from the int8 NHWC tensor into the operand registers -- each byte is fetched once per workgroup;
//   * weights (A operand) are shared by the 4 waves and staged through a double-buffered, XOR-swizzled
//     LDS tile of TK rows x 128 bytes per K-step.
// kPath 1 (C % 128 == 0 and K % TK == 0): every K-step lies inside one tap and every weight row / chunk exists,
// so the per-step index arithmetic collapses to one tap update and pointer increments (the general path, 0,
// spends ~260 VALU instructions per K-step on it, against 16 MFMAs).  kPath 2 (C == 64, the first stage of a
// ResNet): a K-step is exactly two taps of 64 bytes, again with uniform tap arithmetic; a missing second tap
// (R*S odd) is an out-of-range activation offset, i.e. zeros.
constexpr int kPathGeneral = 0, kPathC128 = 1, kPathC64 = 2;
// Register budget of the register-staged kernel, as waves per SIMD the compiler must make room for.  Left alone,
// hipcc keeps the accumulators in AGPRs next to 85-164 VGPRs: 2 waves per SIMD for the fused-add epilogue (the
// residual stream of the 1x1 expand layers then ran at 4.6-5.1 TB/s).  3 waves fit without spilling (resident
// ResNet-50 forward at batch 128: 1.96 -> 1.83 ms); asking for 4 on the 128-row tiles spills 6-19 dwords and
// costs more than it gains, the 64-row tiles fit 4.
#ifndef FQ_WAVES_TK64
#define FQ_WAVES_TK64 4
#endif
#ifndef FQ_WAVES_TK128
#define FQ_WAVES_TK128 3
#endif
template <int TK, int kPath, int kOut>
__global__ __launch_bounds__(kConvBlock) __attribute__((amdgpu_waves_per_eu(TK == 64 ? FQ_WAVES_TK64 : FQ_WAVES_TK128))) void conv2d_i8_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                               const float* __restrict__ qbias, float* __restrict__ y,
                                                               int8_t* __restrict__ q, const ConvParams p) {
    constexpr int BKB = 128;              // bytes of the reduction axis per K-step (8 chunks, 4 MFMA sub-steps)
    constexpr int MT = TK / 32;           // 32-row MFMA tiles per wave along k_out
    constexpr int A_LOADS = TK / 32;      // 16-byte chunks per thread per K-step for the weight tile
    __shared__ __attribute__((aligned(16))) int8_t sA[2][TK * BKB];
    __shared__ float sBias[TK];
    __shared__ __attribute__((aligned(16))) int sBiasI[TK];     // (integer tail: the rounding constant with the bias in it, tail_consts)
    __shared__ __attribute__((aligned(16))) int sLH[2][TK];     //  ... and the merged clamp's bounds

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5;
    int tile_x, tile_y;
    if (!conv_tile_of(p, tile_x, tile_y)) return;
    const int m0 = tile_x * kTP;
    const int k0 = tile_y * TK;
    const int PQ = p.P * p.Q;
    if (tid < TK) {                                       // visible after the first barrier
        const float b = (k0 + tid < p.K) ? qbias[k0 + tid] : 0.0f;
        sBias[tid] = b;
        const TailK tk = tail_consts((int)b, p);           // (integer valued by contract)
        sBiasI[tid] = tk.B; sLH[0][tid] = tk.lo; sLH[1][tid] = tk.hi;
    }

    // this lane's output pixel
    const int m = m0 + wave * 32 + (lane & 31);
    const bool m_ok = m < p.M;
    int ih0, iw0, n_img = 0, pq = 0;
    {
        const int mm = m_ok ? m : 0;
        n_img = mm / PQ; pq = mm - n_img * PQ;
        const int op = pq / p.Q, oq = pq - op * p.Q;
        ih0 = m_ok ? op * p.stride_h - p.pad_h : -(1 << 28);          // out-of-range pixel: every tap misses
        iw0 = oq * p.stride_w - p.pad_w;
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(x), 0, p.x_bytes, 0x00020000);
    const unsigned img_off = (unsigned)n_img * (unsigned)(p.H * p.W * p.C);

    // weight staging: thread -> (row = tid >> 3 (+32 per load), chunk-in-step = tid & 7)
    const int ld_row = tid >> 3, ld_chunk = tid & 7;
    const long wrow_bytes = (long)p.chunks * 16;
    int ga = ld_chunk;                                   // this thread's weight chunk on the reduction axis
    const int8_t* wrow[A_LOADS];                         // row bases: 64-bit once, 32-bit offsets per step
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) wrow[j] = w + (long)(k0 + ld_row + 32 * j) * wrow_bytes;

    // activation chunk of sub-step 0 for this lane: g = step*8 + half; sub-step ks adds 2*ks
    int gb = half;
    RedPos pb;
    pb.cc = gb % p.c16;
    { const int rs0 = gb / p.c16; pb.fr = rs0 / p.S; pb.fs = rs0 - pb.fr * p.S; }

    v4i ra[A_LOADS], rb[4];
    // fast-path state: current tap (uniform), steps left inside it, per-lane byte offset of the tap
    int tap_r = 0, tap_s = 0, c_step = 0;
    const int steps_per_tap = p.C >> 7;
    const int8_t* wp[A_LOADS];
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) wp[j] = wrow[j] + ld_chunk * 16;
    // kPathC64: weight offsets for buffer loads (rows k >= K and the overrun of an odd tap count read as zeros
    // or meet zero activations)
    const __amdgpu_buffer_rsrc_t wr64 = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(w), 0, p.w_bytes, 0x00020000);
    unsigned wo64[A_LOADS];
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
        const int k = k0 + ld_row + 32 * j;
        wo64[j] = k < p.K ? (unsigned)k * (unsigned)(p.chunks * 16) + (unsigned)(ld_chunk * 16) : kOutOfRange;
    }
    auto load_step = [&]() {
        if (kPath == kPathC64) {
#pragma unroll
            for (int j = 0; j < A_LOADS; ++j) {
                ra[j] = load_act(wr64, wo64[j]);
                wo64[j] += BKB;
            }
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {                // the two taps of this K-step
                const int ih = ih0 + tap_r * p.dil_h, iw = iw0 + tap_s * p.dil_w;
                const bool ok = tap_r < p.R && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const unsigned off = ok ? img_off + (unsigned)((ih * p.W + iw) * 64 + half * 16) : kOutOfRange;
                rb[2 * tp] = load_act(xr, off);
                rb[2 * tp + 1] = load_act(xr, off + 32);
                if (++tap_s == p.S) { tap_s = 0; ++tap_r; }
            }
            return;
        }
        if (kPath == kPathC128) {
#pragma unroll
            for (int j = 0; j < A_LOADS; ++j) {
                ra[j] = *reinterpret_cast<const v4i*>(wp[j]);
                wp[j] += BKB;
            }
            const int ih = ih0 + tap_r * p.dil_h, iw = iw0 + tap_s * p.dil_w;
            const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const unsigned off = ok ? img_off + (unsigned)((ih * p.W + iw) * p.C + c_step * BKB + half * 16) : kOutOfRange;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) rb[ks] = load_act(xr, off + ks * 32);       // chunks of one lane are 2 apart
            if (++c_step == steps_per_tap) {
                c_step = 0;
                if (++tap_s == p.S) { tap_s = 0; ++tap_r; }
            }
            return;
        }
        // branch-free: an out-of-range chunk reads a valid dummy address and is zeroed by a select,
        // so all loads of a step issue back to back
        const bool a_live = ga < p.chunks;
        const v4i zero = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < A_LOADS; ++j) {
            const int k = k0 + ld_row + 32 * j;
            const bool ok = a_live && k < p.K;
            const v4i v = *reinterpret_cast<const v4i*>(ok ? wrow[j] + (unsigned)(ga * 16) : w);
            ra[j] = ok ? v : zero;
        }
        ga += 8;
        RedPos q = pb;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ih = ih0 + q.fr * p.dil_h, iw = iw0 + q.fs * p.dil_w;
            const bool ok = gb + 2 * ks < p.chunks && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            rb[ks] = load_act(xr, ok ? img_off + (unsigned)((ih * p.W + iw) * p.C + q.cc * 16) : kOutOfRange);
            red_advance(q, 2, p.c16, p.S);
        }
        pb = q;                                           // 4 x (+2) = +8: first chunk of the next K-step
        gb += 8;
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int j = 0; j < A_LOADS; ++j) {
            const int row = ld_row + 32 * j;
            *reinterpret_cast<v4i*>(&sA[buf][row * BKB + swz(row, ld_chunk) * 16]) = ra[j];
        }
    };

    v16i acc[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0;

    // LDS byte offsets of this lane's A fragments: row = a*32 + (lane&31); the XOR term (row>>1)&7 does
    // not depend on a (a*16 is a multiple of 8), so the swizzled chunk offset is shared by all tiles
    int a_off[MT], swz_off[4];
#pragma unroll
    for (int a = 0; a < MT; ++a) a_off[a] = (a * 32 + (lane & 31)) * BKB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) swz_off[ks] = ((ks * 2 + half) ^ (((lane & 31) >> 1) & 7)) * 16;

    const int nsteps = (p.chunks + 7) >> 3;
    load_step();
    ResRegs<TK> res;
    if constexpr ((kOut & kOutResEarly) != 0) {
        // the fused NewAdd's residual tile (32 KB per workgroup) requested behind the first K-step's operand loads -- vmcnt retires
        // in issue order, so the operands are not held up -- its latency then runs beside the LDS staging and the MFMAs
        load_residual<TK>(res, p, lane, wave, m0, k0);
        __builtin_amdgcn_sched_barrier(0);
    }
    store_a(0);
    v4i fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = rb[ks];
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        if (step + 1 < nsteps) load_step();              // next step's global loads fly under the MFMAs
        // A fragments are read one sub-step ahead into their own registers: every fragment feeds a
        // single MFMA (each wave owns all TK rows), so without this the ds_read latency sits between
        // every pair of MFMAs
        v4i fa[2][MT];
#pragma unroll
        for (int a = 0; a < MT; ++a)
            fa[0][a] = *reinterpret_cast<const v4i*>(&sA[cur][a_off[a] + swz_off[0]]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
#pragma unroll
                for (int a = 0; a < MT; ++a)
                    fa[(ks + 1) & 1][a] = *reinterpret_cast<const v4i*>(&sA[cur][a_off[a] + swz_off[ks + 1]]);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
                acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks & 1][a], fb[ks], acc[a], 0, 0, 0);
        }
        if (step + 1 < nsteps) {
            store_a(cur ^ 1);                             // the other buffer was last read one barrier ago
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb[ks] = rb[ks];
            __syncthreads();
        }
    }

    static_assert(kTP * (TK + 16) <= 2 * TK * BKB, "the int8 output tile is staged in the weight buffers");
    if (p.rs) conv_epilogue<TK, kOut, true>(acc, p, y, q, &sA[0][0], sBias, sBiasI, sLH, m0, k0, n_img, pq, m_ok, res);
    else conv_epilogue<TK, kOut, false>(acc, p, y, q, &sA[0][0], sBias, sBiasI, sLH, m0, k0, n_img, pq, m_ok, res);
}

// ---- C % 128 == 0, K % TK == 0: both operands by LDS-DMA ----------------------------------------------
// The kernel above loads the activation operand straight into MFMA layout: lane = pixel, 16 bytes per
// lane, so the 64 lanes of one load touch 64 different 64-byte sectors and the vector L1 spends one tag
// lookup per lane (rocprofv3, 3x3 256->256 14x14 layer at batch 128: 37 L1 accesses per wave-load, L1 busy
// 73 % of the kernel).  An s_memtime trace of one wave of that kernel (FQ_CONV_TRACE build,
// scripts/conv_trace.py) showed the K-step as a serial chain at one wave per SIMD: ~700 cycles issuing 8
// loads and their address arithmetic, ~900 in the MFMA phase (512 of matrix work plus exposed ds_read
// latency), ~330 waiting for the loads, ~280 staging the weight tile registers -> LDS, ~360 in the barrier,
// ~220 re-reading the activation fragments: 2 800 cycles per 512 cycles of MFMA.  This variant removes the
// links one by one:
//   * both tiles arrive by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write): 8 adjacent
//     lanes fetch the 8 chunks (128 contiguous bytes) of one row, so a wave-load touches 8 full lines, and
//     lane l lands at LDS base + 16*l.  The XOR swizzle that makes the MFMA-layout ds_read_b128 conflict
//     free is applied on the global side: LDS position c of row r receives chunk c ^ ((r >> 1) & 7);
//   * the K-step advance is the scalar offset of the buffer instruction; the per-lane tap offsets are a
//     handful of branch-free vector ops per step.  Out-of-image taps (and the step past the last one) use an
//     offset beyond num_records and arrive as zeros without touching memory;
//   * the loop body is one basic block in which the next step's 8 DMA issues and the second half of the
//     fragment reads are written, and pinned with sched_group_barrier, one behind each of the first 2*MT
//     MFMAs, so issue cost and ds_read latency hide under the matrix pipe (dma_to_lds explains why the DMA
//     is inline asm).
// Each wave stages the activation rows it multiplies itself and a quarter of the weight tile; one
// vmcnt(0) + workgroup barrier per K-step orders the DMA writes before the next step's reads.
// What bounds it now is operand delivery: ~16 bytes per clock per CU at 128 ops per loaded byte.
// (accumulators in VGPRs: this file is compiled with -mllvm -amdgpu-mfma-vgpr-form=1, csrc/Makefile -- left alone hipcc keeps
//  them in AGPRs and every one crosses v_accvgpr_write at tile start and v_accvgpr_read on its way into the tail: 192 vector
//  instructions per wave and tile)
template <int TK, int kOut, int STAGES>
__global__ __launch_bounds__(kConvBlock) void conv2d_i8_dma_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                   const float* __restrict__ qbias, float* __restrict__ y,
                                                                   int8_t* __restrict__ q, const ConvParams p) {
    constexpr int BKB = 128;
    constexpr int MT = TK / 32;
    constexpr int A_LOADS = TK / 32;
    static_assert(STAGES == 2 || STAGES == 3, "LDS ring of 2 or 3 K-steps");
    __shared__ __attribute__((aligned(16))) int8_t sA[STAGES][TK * BKB];
    __shared__ __attribute__((aligned(16))) int8_t sB[STAGES][kTP * BKB];
    __shared__ float sBias[TK];
    __shared__ __attribute__((aligned(16))) int sBiasI[TK];     // (integer tail: the rounding constant with the bias in it, tail_consts)
    __shared__ __attribute__((aligned(16))) int sLH[2][TK];     //  ... and the merged clamp's bounds

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: LDS-DMA bases live in SGPRs (M0)
    const int half = lane >> 5;
    int tile_x, tile_y;
    if (!conv_tile_of(p, tile_x, tile_y)) return;
    const int m0 = tile_x * kTP;
    const int k0 = tile_y * TK;
    const int PQ = p.P * p.Q;
#ifdef FQ_CONV_TRACE
    const bool trace_on = (tile_x % 37 == 5) && tile_y == 0 && wave == 1 && tile_x < 37 * 8;
    const int trace_base = (tile_x / 37) * 512;
#endif
    TR(0);
    if (tid < TK) {                                       // visible after the first barrier
        const float b = qbias[k0 + tid];
        sBias[tid] = b;
        const TailK tk = tail_consts((int)b, p);
        sBiasI[tid] = tk.B; sLH[0][tid] = tk.lo; sLH[1][tid] = tk.hi;
    }

    // output pixel of this lane in MFMA layout (used by the fp32 epilogue)
    const int m = m0 + wave * 32 + (lane & 31);
    const bool m_ok = m < p.M;
    int n_img = 0, pq = 0;
    if (kOut & kOutF32) {
        const int mm = m_ok ? m : 0;
        n_img = mm / PQ; pq = mm - n_img * PQ;
    }

    // activation staging: in load j this lane fetches pixel row pj = 8j + (lane >> 3) of the wave's 32 and
    // LDS position lane & 7, i.e. global chunk (lane & 7) ^ swz(pj)
    const rsrc_words xr = make_rsrc_words(x, p.x_bytes);
    int ih0[4], iw0[4];
    unsigned boff[4];                                     // image base + chunk byte offset inside a K-step
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pj = 8 * j + (lane >> 3);
        const int mj = m0 + wave * 32 + pj;
        const bool ok = mj < p.M;
        const int mm = ok ? mj : 0;
        const int nj = mm / PQ, pqj = mm - nj * PQ;
        const int op = pqj / p.Q, oq = pqj - op * p.Q;
        ih0[j] = ok ? op * p.stride_h - p.pad_h : -(1 << 28);         // out-of-range pixel: every tap misses
        iw0[j] = oq * p.stride_w - p.pad_w;
        boff[j] = (unsigned)nj * (unsigned)(p.H * p.W * p.C) + (unsigned)(((lane & 7) ^ swz(pj, 0)) * 16);
    }
    // weight staging: load j of wave v covers tile rows 32j + 8v .. +7 (one 128-byte K-step row per 8 lanes)
    const rsrc_words wr = make_rsrc_words(w, p.w_bytes);
    unsigned aoff[A_LOADS];
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
        const int row = 32 * j + 8 * wave + (lane >> 3);
        aoff[j] = (unsigned)(k0 + row) * ((unsigned)p.chunks * 16u) + (unsigned)(((lane & 7) ^ swz(row, 0)) * 16);
    }

    int tap_r = 0, tap_s = 0, c_step = 0;                 // filter tap (uniform) and K-step inside it
    const int steps_per_tap = p.C >> 7;
    unsigned bvo[4];                                      // activation offsets of the tap being loaded
    // offset of tap (r, s) = base0[j] + (r * dil_h * W + s * dil_w) * C: the second term is wave-uniform (scalar
    // ALU), so a step costs each lane one add and two range checks per row group -- no vector multiplies (the
    // first version recomputed (ih * W + iw) * C per lane: ~500 cycles of quarter-rate v_mul_lo_u32 per step)
    unsigned base0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) base0[j] = boff[j] + (unsigned)((ih0[j] * p.W + iw0[j]) * p.C);
    auto tap_offsets = [&](bool live) {                   // branch-free; `live` = false: every lane out of range
        const int dh = tap_r * p.dil_h, dw = tap_s * p.dil_w;
        const unsigned tap_delta = (unsigned)((dh * p.W + dw) * p.C);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = live && (unsigned)(ih0[j] + dh) < (unsigned)p.H && (unsigned)(iw0[j] + dw) < (unsigned)p.W;
            bvo[j] = ok ? base0[j] + tap_delta : kOutOfRange;
        }
    };
    // DMA instruction k of a K-step (k < A_LOADS: weight rows, then the 4 activation row groups) into ring
    // slot `buf`; a_live = false sends the weight lanes out of range too (a step past the end loads nothing)
    auto dma = [&](auto k_tag, int buf, int step, bool a_live) {
        constexpr int k = decltype(k_tag)::value;
        if constexpr (k < A_LOADS)
            dma_to_lds(wr, lds_offset(&sA[buf][(32 * k + 8 * wave) * BKB]), a_live ? aoff[k] : kOutOfRange, step * BKB);
        else
            dma_to_lds(xr, lds_offset(&sB[buf][(wave * 32 + 8 * (k - A_LOADS)) * BKB]), bvo[k - A_LOADS], c_step * BKB);
    };
    auto advance_tap = [&]() {
        if (++c_step == steps_per_tap) {
            c_step = 0;
            if (++tap_s == p.S) { tap_s = 0; ++tap_r; }
        }
    };

    v16i acc[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0;

    int a_off[MT], swz_off[4];
#pragma unroll
    for (int a = 0; a < MT; ++a) a_off[a] = (a * 32 + (lane & 31)) * BKB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) swz_off[ks] = ((ks * 2 + half) ^ (((lane & 31) >> 1) & 7)) * 16;
    const int b_row = (wave * 32 + (lane & 31)) * BKB;    // this lane's pixel row in the activation tile

    const int nsteps = p.chunks >> 3;
    constexpr int kLoads = A_LOADS + 4;                   // DMA instructions one wave issues per K-step
    auto for_each_dma = [&](auto&& f) {                   // f(integral_constant<k>) for k = 0 .. kLoads-1
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) { (f(std::integral_constant<int, Ks>{}), ...); }
        (std::make_integer_sequence<int, kLoads>{});
    };
    TR(1);
    // STAGES - 1 K-steps are in flight.  With a ring of 3 the wait at the end of step s is for step s + 1 only
    // (vmcnt counts in issue order; the kLoads newest -- step s + 2 -- may stay outstanding), which takes the
    // ~400 cycles of DMA latency that a ring of 2 exposes on every step off the critical path, at the price of
    // 1.5x the LDS (96 / 72 KB: one / two workgroups per CU).
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0) {
        const bool live0 = s0 < nsteps;
        tap_offsets(live0);
        for_each_dma([&](auto k) { dma(k, s0, s0, live0); });
        advance_tap();
    }
    if (STAGES == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's DMA rows of step 0 have landed ...
    __syncthreads();                                      // ... and so have everybody else's
    TR(2);
    // The loop body is one basic block: the 8 DMA issues of the step being prefetched (~65 cycles each for the
    // issuing wave) are written -- and pinned with sched_group_barrier -- BETWEEN this step's MFMAs, one behind
    // each of the first 2*MT, together with the second half of the fragment reads, so that issue cost and
    // ds_read latency hide under the matrix pipe instead of preceding it.  A step past the end sends every lane
    // out of range (no traffic) rather than branching, which would split the block.
    int cur = 0, nxt = STAGES - 1;                        // ring slots of this step and of the step being prefetched
    for (int step = 0; step < nsteps; ++step) {
        const int pre = step + STAGES - 1;                // the K-step whose tiles are requested during this one
        const bool more = pre < nsteps;
        TR(8 + step * 8 + 0);
        tap_offsets(more);
        TR(8 + step * 8 + 1);
        v4i fb[4], fa[4][MT];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const v4i*>(&sB[cur][b_row + swz_off[ks]]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < MT; ++a)
                fa[ks][a] = *reinterpret_cast<const v4i*>(&sA[cur][a_off[a] + swz_off[ks]]);
        [&]<int... Is>(std::integer_sequence<int, Is...>) {
            ([&] {
                constexpr int i = Is, ks = i / MT, a = i % MT;
                acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks][a], fb[ks], acc[a], 0, 0, 0);
                if constexpr (i < 2 * MT) {
                    fa[2 + ks][a] = *reinterpret_cast<const v4i*>(&sA[cur][a_off[a] + swz_off[2 + ks]]);
                    constexpr int k0_ = kLoads * i / (2 * MT), k1_ = kLoads * (i + 1) / (2 * MT);
                    if constexpr (k1_ > k0_) dma(std::integral_constant<int, k0_>{}, nxt, pre, more);
                    if constexpr (k1_ > k0_ + 1) dma(std::integral_constant<int, k0_ + 1>{}, nxt, pre, more);
                }
            }(), ...);
        }(std::make_integer_sequence<int, 4 * MT>{});
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * MT, 0);
        [&]<int... Is>(std::integer_sequence<int, Is...>) {
            ([&] {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, kLoads * (Is + 1) / (2 * MT) - kLoads * Is / (2 * MT), 0);
            }(), ...);
        }(std::make_integer_sequence<int, 2 * MT>{});
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT, 0);
        advance_tap();
        TR(8 + step * 8 + 2);
        // the next step's tiles must have landed; with a ring of 3 the one just requested may still be in flight
        if (STAGES == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TR(8 + step * 8 + 3);
        __syncthreads();
        TR(8 + step * 8 + 4);
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of this wave may still land in LDS: the epilogue reuses it
    TR(3);

    static_assert(kTP * (TK + 16) <= STAGES * TK * BKB, "the int8 output tile is staged in the weight buffers");
    ResRegs<TK> res;
    if (p.rs) conv_epilogue<TK, kOut, true>(acc, p, y, q, &sA[0][0], sBias, sBiasI, sLH, m0, k0, n_img, pq, m_ok, res);
    else conv_epilogue<TK, kOut, false>(acc, p, y, q, &sA[0][0], sBias, sBiasI, sLH, m0, k0, n_img, pq, m_ok, res);
    TR(4);
}

// ---- 3 x 3, stride 1, padding 1, C % 128 == 0: the activation operand as a RESIDENT HALO ---------------------------
// In the kernel above every K-step (one filter tap of one 128-channel slice) fetches its activation tile again: 128 pixel
// rows x 128 bytes by LDS-DMA, gathered by tap -- 9 fetches of (almost) the same pixels per slice, half of the 32 KB a
// step moves into LDS, and with them half of the DMA issues (~45 cycles each for the issuing wave) and of the L2 -> LDS
// traffic that, at 64 bytes per clock and CU, costs as many cycles as the step's 64 MFMAs.
// With stride 1 and padding 1 the input pixel of (output pixel m, tap (r, s)) is pixel m + (r - 1) W + (s - 1) of the SAME
// flat N x H x W order (where it exists): the 128 output pixels of a tile need the contiguous run of input pixels
// [m0 - W - 1, m0 + 128 + W] -- 130 + 2 W rows of 128 bytes per channel slice.  That slab is fetched ONCE per slice,
// a tap is a row offset r W + s into it, and taps that fall outside
// the image (padding; the slab holds a neighbouring row's pixel there) read a 128-byte zero row instead.  Per step: the
// weight tile only (16 KB, half the DMA issues), 12 vector instructions of tap addressing per lane, same epilogue.
// Reduction order: slices outer, taps inner -- integer accumulation, any order gives the same bits.
struct HaloParams {
    int slab_rows;                         // rows of one slab buffer (130 + 2 W rounded up to 8)
    int total_pixels;                      // N * H * W
};

// NP: pixel sub-tiles of 32 per wave (1: 128-pixel tile as above; 2: 256-pixel tile, each wave multiplies two sub-tiles
// against every weight fragment it reads -- 6 LDS fragment reads per 8 MFMAs instead of 5 per 4, and 32 MFMAs per wave
// between two barriers instead of 16, which is what the counters asked for: at NP = 1 the matrix pipe is 27 % busy while
// no other unit passes 45 %, the waves simply spend two thirds of their time in waits and issue stalls around 16 MFMAs).
// Sub-tile u of wave w is pixels m0 + 128 u + 32 w + lane: each 128-pixel half of the tile is a tile of the shared epilogue.
template <int TK, int kOut, int ST, int NP>
__global__ __launch_bounds__(kConvBlock) void conv3x3_i8_halo_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                     const float* __restrict__ qbias, float* __restrict__ y,
                                                                     int8_t* __restrict__ q, const ConvParams p, const HaloParams hp) {
    constexpr int BKB = 128;
    constexpr int MT = TK / 32;
    constexpr int A_LOADS = TK / 32;
    constexpr int TPX = kTP * NP;                             // pixels per workgroup tile
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    // [weights: ST x TK x 128][bias f32: TK][bias i32: TK][zero row: 128][slab: slab_rows x 128]
    static_assert(ST == 2 || ST == 3, "weight ring of two or three steps");
    int8_t* const sA = smem;
    float* const sBias = reinterpret_cast<float*>(smem + ST * TK * BKB);
    int* const sBiasI = reinterpret_cast<int*>(smem + ST * TK * BKB + TK * 4);
    __shared__ __attribute__((aligned(16))) int sLH[2][TK];     // the merged clamp's bounds of the integer tail (tail_consts)
    int8_t* const sZero = smem + ST * TK * BKB + TK * 8;
    int8_t* const sSlab = sZero + BKB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    int tile_x, tile_y;
    if (!conv_tile_of(p, tile_x, tile_y)) return;
    const int m0 = tile_x * TPX;
    const int k0 = tile_y * TK;
    const int PQ = p.P * p.Q;
    if (tid < TK) {
        const float b = qbias[k0 + tid];
        sBias[tid] = b;
        const TailK tk = tail_consts((int)b, p);
        sBiasI[tid] = tk.B; sLH[0][tid] = tk.lo; sLH[1][tid] = tk.hi;
    }
    if (tid < BKB / 4) reinterpret_cast<int*>(sZero)[tid] = 0;

    // this lane's output pixels (one per sub-tile) and which of their nine taps exist
    bool m_ok[NP];
    int n_img[NP], pq[NP];
    unsigned tap_mask[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int m = m0 + kTP * u + wave * 32 + (lane & 31);
        m_ok[u] = m < p.M;
        const int mm = m_ok[u] ? m : 0;
        n_img[u] = mm / PQ; pq[u] = mm - n_img[u] * PQ;
        const int oh = pq[u] / p.Q, ow = pq[u] - oh * p.Q;
        tap_mask[u] = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const bool ok = m_ok[u] && (unsigned)(oh + r - 1) < (unsigned)p.H && (unsigned)(ow + s_ - 1) < (unsigned)p.W;
                tap_mask[u] |= ok ? 1u << (3 * r + s_) : 0u;
            }
    }

    // slab fetch: DMA instruction i covers slab rows 8 i .. 8 i + 7 (input pixels m0 - W - 1 + row), one 128-byte row per 8 lanes
    const rsrc_words xr = make_rsrc_words(x, p.x_bytes);
    const int n_slab_dma = hp.slab_rows >> 3;
    auto slab_dma = [&](int slice) {
        for (int i = wave; i < n_slab_dma; i += 4) {
            const int row = 8 * i + (lane >> 3);
            const int g = m0 - p.W - 1 + row;                 // input pixel (flat N x H x W index); outside the tensor: zeros
            const unsigned vo = (unsigned)g < (unsigned)hp.total_pixels
                                    ? (unsigned)g * (unsigned)p.C + (unsigned)(((lane & 7) ^ swz(row, 0)) * 16) : kOutOfRange;
            dma_to_lds(xr, lds_offset(sSlab + i * 1024), vo, slice * BKB);
        }
    };
    // weight fetch, as in the kernel above: load j of wave v covers tile rows 32 j + 8 v .. + 7
    const rsrc_words wr = make_rsrc_words(w, p.w_bytes);
    unsigned aoff[A_LOADS];
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
        const int row = 32 * j + 8 * wave + (lane >> 3);
        aoff[j] = (unsigned)(k0 + row) * ((unsigned)p.chunks * 16u) + (unsigned)(((lane & 7) ^ swz(row, 0)) * 16);
    }
    auto weight_dma = [&](int slice, int tap, int buf, bool live) {
        const int so = tap * p.C + slice * BKB;               // [K][3][3][C]: tap (r, s) of channel slice `slice`
#pragma unroll
        for (int j = 0; j < A_LOADS; ++j)
            dma_to_lds(wr, lds_offset(sA + buf * TK * BKB + (32 * j + 8 * wave) * BKB), live ? aoff[j] : kOutOfRange, so);
    };

    v16i acc[NP][MT];
#pragma unroll
    for (int u = 0; u < NP; ++u)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][a][r] = 0;
    int a_off[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) a_off[a] = (a * 32 + (lane & 31)) * BKB;
    int swz_a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) swz_a[ks] = ((ks * 2 + half) ^ (((lane & 31) >> 1) & 7)) * 16;
    const int p_row = wave * 32 + (lane & 31);                // this lane's pixel inside a 128-pixel half of the tile
    const unsigned zero_off = (unsigned)(sZero - smem);
    const unsigned slab_off0 = (unsigned)(sSlab - smem);

    // Weights: ring of ST steps.  With three, the tile requested during step s is the one of step s + 2, so the wait at the
    // end of a step (vmcnt(A_LOADS): everything but the newest request) is for data that had a whole step to arrive.  The
    // slab has ONE buffer: between two channel slices every wave finishes the last tap, the slab is fetched again and
    // waited for in the open -- once per nine steps.
    const int nslices = p.C >> 7;
    const int nsteps = 9 * nslices;
    auto step_of = [&](int st, int& sl, int& tp) { sl = st / 9; tp = st - 9 * sl; };
    slab_dma(0);
    weight_dma(0, 0, 0, true);
    if (ST == 3) weight_dma(0, 1, 1, nsteps > 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int slice = 0, tap = 0, cur = 0, nxt2 = ST - 1;
    for (int step = 0; step < nsteps; ++step) {
        // this tap's activation fragments out of the resident slab: row = pixel + r W + s, or the zero row
        const int r = tap / 3, s_ = tap - 3 * r;
        v4i fb[NP][4], fa[2][MT];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int j = kTP * u + p_row + r * p.W + s_;
            const bool ok = (tap_mask[u] >> tap) & 1u;
            const unsigned rowb = ok ? slab_off0 + (unsigned)(j * BKB) : zero_off;
            const int sw = ok ? (j >> 1) & 7 : 0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fb[u][ks] = *reinterpret_cast<const v4i*>(smem + rowb + (unsigned)((((ks * 2 + half) ^ sw)) * 16));
        }
        const int8_t* const sAc = sA + cur * TK * BKB;
#pragma unroll
        for (int a = 0; a < MT; ++a) fa[0][a] = *reinterpret_cast<const v4i*>(sAc + a_off[a] + swz_a[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
#pragma unroll
                for (int a = 0; a < MT; ++a) fa[(ks + 1) & 1][a] = *reinterpret_cast<const v4i*>(sAc + a_off[a] + swz_a[ks + 1]);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int u = 0; u < NP; ++u)
                    acc[u][a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks & 1][a], fb[u][ks], acc[u][a], 0, 0, 0);
            if (ks == 0) {
                // the weight tile of step + ST - 1, requested behind the first MFMAs of this step (the matrix pipe has work
                // queued while the wave sits in the DMA issues); it goes where step - 1 was read from
                int sl2, tp2;
                step_of(step + ST - 1, sl2, tp2);
                __builtin_amdgcn_sched_barrier(0);
                weight_dma(sl2, tp2, nxt2, step + ST - 1 < nsteps);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LOADS) : "memory");     // step + 1's weights have landed (mine) ...
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                     // ... and everybody's; everybody is done with this step
        cur = cur + 1 == ST ? 0 : cur + 1;
        nxt2 = nxt2 + 1 == ST ? 0 : nxt2 + 1;
        if (++tap == 9) {
            tap = 0;
            if (++slice < nslices) {                          // next channel slice: the slab again, in the open
                slab_dma(slice);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing of this wave may still land in LDS: the epilogue reuses it
    __syncthreads();

#pragma unroll
    for (int u = 0; u < NP; ++u) {
        if (u) __syncthreads();                               // the epilogue stages its int8 tile in sA: one half after the other
        ResRegs<TK> res;
        if (p.rs) conv_epilogue<TK, kOut, true>(acc[u], p, y, q, sA, sBias, sBiasI, sLH, m0 + kTP * u, k0, n_img[u], pq[u], m_ok[u], res);
        else conv_epilogue<TK, kOut, false>(acc[u], p, y, q, sA, sBias, sBiasI, sLH, m0 + kTP * u, k0, n_img[u], pq[u], m_ok[u], res);
    }
}

// ---- the same with 256-pixel tiles and EIGHT waves ---------------------------------------------------------------------
// What bounds the kernel above is operand delivery into LDS (a round of tiles moves its operands at 11-13 bytes
// per clock and CU, the LDS-DMA fill rate of the chip, and a 128 x 128 tile needs a byte per 128 MACs).  Here one weight tile
// feeds 256 pixels: waves 0-3 own the first 128 pixels of the tile, waves 4-7 the second 128, every wave as in the kernel
// above (32 pixels x TK channels, the same fragments, the same 16 MFMAs per step) -- so the bytes per MAC halve while the
// waves per SIMD do not (the four-wave NP = 2 form paid for the same halving with half the resident waves and lost).
// Weights in a ring of two (32 KB) + one slab of 258 + 2 W rows: 70 KB at W = 14, two workgroups = 16 waves per CU.
template <int TK, int kOut, int ST>
__global__ __launch_bounds__(2 * kConvBlock) void conv3x3_i8_halo8_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                          const float* __restrict__ qbias, float* __restrict__ y,
                                                                          int8_t* __restrict__ q, const ConvParams p, const HaloParams hp) {
    constexpr int BKB = 128;
    constexpr int MT = TK / 32;
    constexpr int A_LOADS = TK / 64;                          // 8 waves x 8 rows per DMA instruction
    constexpr int TPX = 2 * kTP;
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    // [weights: ST x TK x 128][bias f32: TK][bias i32: TK][zero row: 128][slab: slab_rows x 128]
    static_assert(ST == 2 || ST == 3, "weight ring of two or three steps");
    static_assert(TK == 64 || TK == 128, "one or two DMA instructions per wave and weight tile");
    int8_t* const sA = smem;
    float* const sBias = reinterpret_cast<float*>(smem + ST * TK * BKB);
    int* const sBiasI = reinterpret_cast<int*>(smem + ST * TK * BKB + TK * 4);
    __shared__ __attribute__((aligned(16))) int sLH[2][TK];     // the merged clamp's bounds of the integer tail (tail_consts)
    int8_t* const sZero = smem + ST * TK * BKB + TK * 8;
    int8_t* const sSlab = sZero + BKB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0 .. 7
    const int u = wave >> 2, wq = wave & 3;                   // which 128-pixel half, which 32 pixels of it
    const int half = lane >> 5;
    int tile_x, tile_y;
    if (!conv_tile_of(p, tile_x, tile_y)) return;
    const int m0 = tile_x * TPX;
    const int k0 = tile_y * TK;
    const int PQ = p.P * p.Q;
    if (tid < TK) {
        const float b = qbias[k0 + tid];
        sBias[tid] = b;
        const TailK tk = tail_consts((int)b, p);
        sBiasI[tid] = tk.B; sLH[0][tid] = tk.lo; sLH[1][tid] = tk.hi;
    }
    if (tid < BKB / 4) reinterpret_cast<int*>(sZero)[tid] = 0;

    const int m = m0 + kTP * u + wq * 32 + (lane & 31);
    const bool m_ok = m < p.M;
    const int mm = m_ok ? m : 0;
    const int n_img = mm / PQ, pq = mm - n_img * PQ;
    unsigned tap_mask = 0;
    {
        const int oh = pq / p.Q, ow = pq - oh * p.Q;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const bool ok = m_ok && (unsigned)(oh + r - 1) < (unsigned)p.H && (unsigned)(ow + s_ - 1) < (unsigned)p.W;
                tap_mask |= ok ? 1u << (3 * r + s_) : 0u;
            }
    }

    const rsrc_words xr = make_rsrc_words(x, p.x_bytes);
    const int n_slab_dma = hp.slab_rows >> 3;
    auto slab_dma = [&](int slice) {
        for (int i = wave; i < n_slab_dma; i += 8) {
            const int row = 8 * i + (lane >> 3);
            const int g = m0 - p.W - 1 + row;                 // input pixel (flat N x H x W index); outside the tensor: zeros
            const unsigned vo = (unsigned)g < (unsigned)hp.total_pixels
                                    ? (unsigned)g * (unsigned)p.C + (unsigned)(((lane & 7) ^ swz(row, 0)) * 16) : kOutOfRange;
            dma_to_lds(xr, lds_offset(sSlab + i * 1024), vo, slice * BKB);
        }
    };
    // weight fetch: load j of wave v covers tile rows 64 j + 8 v .. + 7
    const rsrc_words wr = make_rsrc_words(w, p.w_bytes);
    unsigned aoff[A_LOADS];
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
        const int row = 64 * j + 8 * wave + (lane >> 3);
        aoff[j] = (unsigned)(k0 + row) * ((unsigned)p.chunks * 16u) + (unsigned)(((lane & 7) ^ swz(row, 0)) * 16);
    }
    auto weight_dma = [&](int slice, int tap, int buf, bool live) {
        const int so = tap * p.C + slice * BKB;
#pragma unroll
        for (int j = 0; j < A_LOADS; ++j)
            dma_to_lds(wr, lds_offset(sA + buf * TK * BKB + (64 * j + 8 * wave) * BKB), live ? aoff[j] : kOutOfRange, so);
    };

    v16i acc[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    int a_off[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) a_off[a] = (a * 32 + (lane & 31)) * BKB;
    int swz_a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) swz_a[ks] = ((ks * 2 + half) ^ (((lane & 31) >> 1) & 7)) * 16;
    const int p_row = kTP * u + wq * 32 + (lane & 31);        // this lane's pixel inside the 256-pixel tile
    const unsigned zero_off = (unsigned)(sZero - smem);
    const unsigned slab_off0 = (unsigned)(sSlab - smem);

    const int nslices = p.C >> 7;
    const int nsteps = 9 * nslices;
    auto step_of = [&](int st, int& sl, int& tp) { sl = st / 9; tp = st - 9 * sl; };
    slab_dma(0);
    weight_dma(0, 0, 0, true);
    if (ST == 3) weight_dma(0, 1, 1, nsteps > 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int slice = 0, tap = 0, cur = 0, nxt2 = ST - 1;
    for (int step = 0; step < nsteps; ++step) {
        const int r = tap / 3, s_ = tap - 3 * r;
        v4i fb[4], fa[2][MT];
        {
            const int j = p_row + r * p.W + s_;
            const bool ok = (tap_mask >> tap) & 1u;
            const unsigned rowb = ok ? slab_off0 + (unsigned)(j * BKB) : zero_off;
            const int sw = ok ? (j >> 1) & 7 : 0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fb[ks] = *reinterpret_cast<const v4i*>(smem + rowb + (unsigned)((((ks * 2 + half) ^ sw)) * 16));
        }
        const int8_t* const sAc = sA + cur * TK * BKB;
#pragma unroll
        for (int a = 0; a < MT; ++a) fa[0][a] = *reinterpret_cast<const v4i*>(sAc + a_off[a] + swz_a[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
#pragma unroll
                for (int a = 0; a < MT; ++a) fa[(ks + 1) & 1][a] = *reinterpret_cast<const v4i*>(sAc + a_off[a] + swz_a[ks + 1]);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a) acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks & 1][a], fb[ks], acc[a], 0, 0, 0);
            if (ks == 0) {
                // the weight tile of step + ST - 1, requested behind the first MFMAs of this step; it goes where step - 1 was
                // read from (free since the barrier that ended that step)
                int sl2, tp2;
                step_of(step + ST - 1, sl2, tp2);
                __builtin_amdgcn_sched_barrier(0);
                weight_dma(sl2, tp2, nxt2, step + ST - 1 < nsteps);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LOADS) : "memory");     // step + 1's weights have landed (mine) ...
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                     // ... and everybody's; everybody is done with this step
        cur = cur + 1 == ST ? 0 : cur + 1;
        nxt2 = nxt2 + 1 == ST ? 0 : nxt2 + 1;
        if (++tap == 9) {
            tap = 0;
            if (++slice < nslices) {                          // next channel slice: the slab again, in the open
                slab_dma(slice);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // each half stages its int8 tile in its own LDS: the first in the weight buffers, the second in the slab
    static_assert(kTP * (TK + 16) <= 2 * TK * BKB, "the first half's int8 tile is staged in the weight buffers");
    int8_t* const sO = u ? sSlab : sA;
    ResRegs<TK> res;
    if (p.rs) conv_epilogue<TK, kOut, true>(acc, p, y, q, sO, sBias, sBiasI, sLH, m0 + kTP * u, k0, n_img, pq, m_ok, res, u * kConvBlock);
    else conv_epilogue<TK, kOut, false>(acc, p, y, q, sO, sBias, sBiasI, sLH, m0 + kTP * u, k0, n_img, pq, m_ok, res, u * kConvBlock);
}

// ---- 3 x 3, stride 1, padding 1, C == 64, K <= 64 (the first stage of a ResNet): weights stationary, persistent -------
// 64 x 3 x 3 x 64 weights are 36 KB: they are fetched ONCE per workgroup and stay in LDS; the workgroup then walks over
// 128-pixel tiles (tile t, t + grid, ...), each of which needs one slab of 130 + 2 W input pixels x 64 bytes -- requested
// for the NEXT tile while this one is multiplied (two slab buffers).  A tile is 9 taps x 2 sub-steps x 2 MFMAs per wave with
// no barrier and no global request in between; at 256 images these layers carry as many bytes as matrix cycles (12.8 us of
// HBM, 11.8 us of MFMA each), so the general kernel's per-tile prologue, weight re-fetch and six barriers were most of it.
struct C64Params {
    int slab_rows;                         // 130 + 2 W rounded up to 16
    int total_pixels;                      // N * H * W
    int tiles;                             // 128-pixel tiles
    int xcd_chunk;                         // tiles per XCD (ceil(tiles / 8)); 0: tiles dealt round robin (FQ_C64_XCD=0)
};

template <int kOut>
__global__ __launch_bounds__(kConvBlock) void conv3x3_i8_c64_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                    const float* __restrict__ qbias, float* __restrict__ y,
                                                                    int8_t* __restrict__ q, const ConvParams p, const C64Params cp) {
    constexpr int TK = 64, MT = 2, RB = 64;                   // output channels per tile, 32-row MFMA tiles, bytes per LDS row
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    // [weights: 9 taps x 64 rows x 64][bias f32: 64][bias i32: 64][zero row: 64][epilogue staging: 128 x 80][slabs: 2 x rows x 64]
    int8_t* const sW = smem;
    float* const sBias = reinterpret_cast<float*>(smem + 9 * TK * RB);
    int* const sBiasI = reinterpret_cast<int*>(smem + 9 * TK * RB + TK * 4);
    __shared__ __attribute__((aligned(16))) int sLH[2][TK];     // the merged clamp's bounds of the integer tail (tail_consts)
    int8_t* const sZero = smem + 9 * TK * RB + TK * 8;
    int8_t* const sO = sZero + RB;
    int8_t* const sSlab = sO + kTP * (TK + 16);
    const int slab_bytes = cp.slab_rows * RB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int PQ = p.P * p.Q;
    if (tid < TK) {
        const float b = tid < p.K ? qbias[tid] : 0.0f;
        sBias[tid] = b;
        const TailK tk = tail_consts((int)b, p);
        sBiasI[tid] = tk.B; sLH[0][tid] = tk.lo; sLH[1][tid] = tk.hi;
    }
    if (tid < RB / 4) reinterpret_cast<int*>(sZero)[tid] = 0;

    // 64-byte rows hold 4 chunks of 16 bytes; position c of row r holds chunk c ^ ((r >> 2) & 3) (conflict-free ds_read_b128
    // of 32 consecutive rows, as for the 128-byte rows: the lane groups of a read cover all 16 slots of the bank row)
    auto swz64 = [](int row) { return (row >> 2) & 3; };
    const rsrc_words wr = make_rsrc_words(w, p.w_bytes);
    for (int i = wave; i < 9 * (TK / 16); i += 4) {           // one DMA instruction: 16 weight rows x 64 bytes of one tap
        const int tap = i / (TK / 16), blk = i - tap * (TK / 16), row = blk * 16 + (lane >> 2);
        const unsigned vo = row < p.K ? (unsigned)row * (unsigned)(9 * RB) + (unsigned)(((lane & 3) ^ swz64(row)) * 16) : kOutOfRange;
        dma_to_lds(wr, lds_offset(sW + (tap * TK + blk * 16) * RB), vo, tap * RB);
    }
    const rsrc_words xr = make_rsrc_words(x, p.x_bytes);
    const int n_slab_dma = cp.slab_rows >> 4;
    // which tiles this workgroup takes: an XCD (workgroup g runs on XCD g % 8: observed placement, for speed only) owns a contiguous
    // eighth of the tile sequence, so that the rows a tile's slab shares with its neighbours' -- 130 + 2 W rows for 128 fresh ones --
    // are hits in that XCD's L2 (dealt g, g + grid, ... the neighbours ran on other XCDs: 104 MB fetched for a 51 MB tensor; round 5)
    int tile, tile_end, tile_step;
    if (cp.xcd_chunk != 0 && (gridDim.x & 7u) == 0) {
        const int xcd = (int)(blockIdx.x & 7u), local = (int)(blockIdx.x >> 3);
        tile = xcd * cp.xcd_chunk + local;
        tile_end = min((xcd + 1) * cp.xcd_chunk, cp.tiles);
        tile_step = (int)(gridDim.x >> 3);
    } else {
        tile = (int)blockIdx.x; tile_end = cp.tiles; tile_step = (int)gridDim.x;
    }
    auto slab_dma = [&](int tile, int buf) {                  // 16 slab rows per instruction
        const int m0 = tile * kTP;
        const bool live = tile < tile_end;
        for (int i = wave; i < n_slab_dma; i += 4) {
            const int row = 16 * i + (lane >> 2);
            const int g = m0 - p.W - 1 + row;
            const unsigned vo = live && (unsigned)g < (unsigned)cp.total_pixels
                                    ? (unsigned)g * (unsigned)RB + (unsigned)(((lane & 3) ^ swz64(row)) * 16) : kOutOfRange;
            dma_to_lds(xr, lds_offset(sSlab + buf * slab_bytes + i * 1024), vo, 0);
        }
    };
    int a_off[MT], swz_a[2];
#pragma unroll
    for (int a = 0; a < MT; ++a) a_off[a] = (a * 32 + (lane & 31)) * RB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) swz_a[ks] = ((ks * 2 + half) ^ swz64(lane & 31)) * 16;
    const int p_row = wave * 32 + (lane & 31);
    const unsigned zero_off = (unsigned)(sZero - smem), slab_off0 = (unsigned)(sSlab - smem);

    int buf = 0;
    slab_dma(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; tile < tile_end; tile += tile_step) {
        slab_dma(tile + tile_step, buf ^ 1);                  // the next tile's slab flies under this tile's arithmetic
        const int m0 = tile * kTP;
        const int m = m0 + p_row;
        const bool m_ok = m < p.M;
        const int mm = m_ok ? m : 0;
        const int n_img = mm / PQ, pq = mm - n_img * PQ;
        const int oh = pq / p.Q, ow = pq - oh * p.Q;
        v16i acc[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0;
        // The operands of a tap -- two 16-byte reads of this lane's input pixel, four of the weights -- are requested kAhead taps
        // before the tap's four MFMAs (round 5).  Left to the compiler the loop was "read, s_waitcnt lgkmcnt(0), MFMA" 18 times per
        // tile on 90 registers: it schedules for an occupancy this kernel's 80 KB of LDS never reaches (two workgroups per CU), and
        // every wait exposed the LDS latency -- 47 waits for 36 MFMAs.  The fences keep the order written here; the waits the
        // compiler derives from it are counted ones that leave the younger taps' reads in flight.  144 registers; 41.5 -> 39.6 us
        // per launch at 256 images (51.4 -> 47.1 stand-alone): the latency was a part of the tile's time, not most of it.
        constexpr int kAhead = 2;
        struct TapOps { v4i fb[2]; v4i fa[2][MT]; };
        TapOps ops[kAhead + 1];
        auto load_tap = [&](int tap, TapOps& o) {
            const int r = tap / 3, s_ = tap - 3 * r;          // compile-time after unrolling
            const int j = p_row + r * p.W + s_;
            const bool ok = m_ok && (unsigned)(oh + r - 1) < (unsigned)p.H && (unsigned)(ow + s_ - 1) < (unsigned)p.W;
            const unsigned rowb = ok ? slab_off0 + (unsigned)(buf * slab_bytes + j * RB) : zero_off;
            const int sw = ok ? swz64(j) : 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                o.fb[ks] = *reinterpret_cast<const v4i*>(smem + rowb + (unsigned)((((ks * 2 + half) ^ sw)) * 16));
#pragma unroll
                for (int a = 0; a < MT; ++a) o.fa[ks][a] = *reinterpret_cast<const v4i*>(sW + tap * TK * RB + a_off[a] + swz_a[ks]);
            }
        };
#pragma unroll
        for (int t = 0; t < kAhead; ++t) load_tap(t, ops[t]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + kAhead < 9) load_tap(tap + kAhead, ops[(tap + kAhead) % (kAhead + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            const TapOps& o = ops[tap % (kAhead + 1)];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < MT; ++a) acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.fa[ks][a], o.fb[ks], acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ResRegs<TK> res;
        if (p.rs) conv_epilogue<TK, kOut, true, false>(acc, p, y, q, sO, sBias, sBiasI, sLH, m0, 0, n_img, pq, m_ok, res);
        else conv_epilogue<TK, kOut, false, false>(acc, p, y, q, sO, sBias, sBiasI, sLH, m0, 0, n_img, pq, m_ok, res);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next slab has landed (mine) ...
        __syncthreads();                                      // ... and everybody's; everybody is done with this tile's slab and staging
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- fp32 NCHW -> int8 NHWC with Quantity fused (new_quantity_op.py:52-58) -----------------------
// q8 (one element): fq_int_tail.h

// block = (64 hw) x (64 c) tile of one image.  Each thread loads a 4(c) x 4(hw) patch with four
// 16-byte loads along hw, quantises, and writes the patch transposed (4 dwords of 4 channels) into an
// LDS tile [hw][c]; the tile is then read back 16 channels at a time and stored with 16-byte stores.
// kVec = false: scalar loads for planes whose rows are not 16-byte aligned (HW % 4 != 0).
template <bool kVec>
__global__ __launch_bounds__(256) void quantize_i8_nhwc_kernel(const float* __restrict__ x, int8_t* __restrict__ y, int C, int HW,
                                                               int Cpad, float scale) {
    constexpr int RS = 17;                                // LDS row stride in dwords (64 c bytes + 4 pad)
    __shared__ unsigned tile[64 * RS];                    // [hw][c / 4]
    const int n = blockIdx.z, hw0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const float* __restrict__ src = x + (long)n * C * HW;
    const int hwg = tid & 15, cg = tid >> 4;              // 16 hw groups of 4, 16 channel groups of 4
    const int hw = hw0 + 4 * hwg;
    float v[4][4];                                        // [channel i][hw e]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + 4 * cg + i;
        const bool c_ok = c < C;
        if (kVec) {
            const bool ok = c_ok && hw < HW;              // HW % 4 == 0: a group is all in or all out
            const float4 f = *reinterpret_cast<const float4*>(ok ? src + (long)c * HW + hw : src);
            v[i][0] = ok ? f.x : 0.0f; v[i][1] = ok ? f.y : 0.0f; v[i][2] = ok ? f.z : 0.0f; v[i][3] = ok ? f.w : 0.0f;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = (c_ok && hw + e < HW) ? src[(long)c * HW + hw + e] : 0.0f;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned packed = q8(v[0][e], scale) | (q8(v[1][e], scale) << 8) | (q8(v[2][e], scale) << 16) |
                                (q8(v[3][e], scale) << 24);
        tile[(4 * hwg + e) * RS + cg] = packed;
    }
    __syncthreads();
    // store: thread -> (hw = tid >> 2, 16 channels = tid & 3)
    const int shw = tid >> 2, cq = tid & 3;
    const int ohw = hw0 + shw, oc = c0 + 16 * cq;
    if (ohw < HW && oc < Cpad) {
        const unsigned* row = &tile[shw * RS + 4 * cq];
        int8_t* dst = y + ((long)n * HW + ohw) * Cpad + oc;
        if (oc + 16 <= Cpad && (Cpad & 15) == 0) {
            uint4 o; o.x = row[0]; o.y = row[1]; o.z = row[2]; o.w = row[3];
            *reinterpret_cast<uint4*>(dst) = o;
        } else {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (oc + 4 * d < Cpad) *reinterpret_cast<unsigned*>(dst + 4 * d) = row[d];
        }
    }
}

// C <= 4 (the image going into the stem convolution), Cpad == 16: one pixel per thread -- C coalesced
// plane reads, one 16-byte store of [c0 c1 c2 c3 0 ... 0].
__global__ __launch_bounds__(256) void quantize_i8_nhwc_smallc_kernel(const float* __restrict__ x, int8_t* __restrict__ y, int C,
                                                                      int HW, long total_pixels, float scale) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < total_pixels; i += stride) {
        const long n = i / HW;
        const int hw = (int)(i - n * HW);
        const float* __restrict__ src = x + n * C * HW + hw;
        unsigned packed = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) packed |= q8(src[(long)c * HW], scale) << (8 * c);
        uint4 o; o.x = packed; o.y = 0; o.z = 0; o.w = 0;
        *reinterpret_cast<uint4*>(y + i * 16) = o;
    }
}

// Stem layers (C <= 4, e.g. 7x7 stride-2 on RGB): channel padding to 16 would waste 13/16 of every
// MFMA.  Instead the kernel width is folded into the channel axis: y[n][ih][q][s*C + c] =
// Quantity(x[n][c][ih][q*stride_w - pad_w + s*dil_w]) (zero outside the image / beyond S*C), Cpad2 bytes
// per output column q.  The convolution then runs with S = 1, stride_w = 1, pad_w = 0 over width Q and
// "channels" Cpad2: an R x 1 kernel with R * Cpad2 reduction bytes instead of R * S * 16.
__global__ __launch_bounds__(256) void quantize_i8_unfold_w_kernel(const float* __restrict__ x, int8_t* __restrict__ y, int C, int H,
                                                                   int W, int S, int stride_w, int pad_w, int dil_w, int Q,
                                                                   int Cpad2, long total, float scale) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < total; i += stride) {                      // i = (n*H + ih)*Q + q
        const int q = (int)(i % Q);
        const long nih = i / Q;
        const int ih = (int)(nih % H);
        const long n = nih / H;
        const float* __restrict__ src = x + (n * C * H + ih) * (long)W;     // + c*H*W + iw
        int8_t* dst = y + i * Cpad2;
        const int iw0 = q * stride_w - pad_w;
        for (int b0 = 0; b0 < Cpad2; b0 += 4) {           // one dword (4 folded channels) at a time
            unsigned packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = b0 + e;                      // folded channel = s*C + c
                const int s_ = f / C, c = f - s_ * C;
                const int iw = iw0 + s_ * dil_w;
                if (s_ < S && (unsigned)iw < (unsigned)W) packed |= q8(src[(long)c * H * W + iw], scale) << (8 * e);
            }
            *reinterpret_cast<unsigned*>(dst + b0) = packed;
        }
    }
}

// The common stem (C <= 4 channels, S <= 8 taps, S*C <= 32 folded channels): C is a template parameter and
// the tap loop is fully unrolled, so every folded-channel index is a compile-time constant -- no division per
// element, the 32 output bytes are assembled in registers and leave as two 16-byte stores.  Neighbouring
// threads (q, q+1) read overlapping input columns, which the L1 absorbs.
template <int C>
__global__ __launch_bounds__(256) void quantize_i8_unfold_w_small_kernel(const float* __restrict__ x, int8_t* __restrict__ y, int H, int W,
                                                                         int S, int stride_w, int pad_w, int dil_w, int Q, int Cpad2,
                                                                         long total, float scale) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const long plane = (long)H * W;
    for (; i < total; i += stride) {                      // i = (n*H + ih)*Q + q
        const int q = (int)(i % Q);
        const long nih = i / Q;
        const int ih = (int)(nih % H);
        const long n = nih / H;
        const float* __restrict__ src = x + (n * C * H + ih) * (long)W;     // + c*H*W + iw
        const int iw0 = q * stride_w - pad_w;
        unsigned out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            const int iw = iw0 + s_ * dil_w;
            const bool ok = s_ < S && (unsigned)iw < (unsigned)W;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int f = s_ * C + c;                  // compile-time after unrolling
                const float v = ok ? src[c * plane + iw] : 0.0f;
                out[f >> 2] |= q8(v, scale) << (8 * (f & 3));
            }
        }
        int8_t* dst = y + i * Cpad2;
        uint4 o0; o0.x = out[0]; o0.y = out[1]; o0.z = out[2]; o0.w = out[3];
        *reinterpret_cast<uint4*>(dst) = o0;
        if (Cpad2 > 16) {
            uint4 o1; o1.x = out[4]; o1.y = out[5]; o1.z = out[6]; o1.w = out[7];
            *reinterpret_cast<uint4*>(dst + 16) = o1;
        }
    }
}

// HW == 1 (Linear input [N][F]): layouts coincide, plain element-wise with channel padding
__global__ __launch_bounds__(256) void quantize_i8_rows_kernel(const float* __restrict__ x, int8_t* __restrict__ y, int C, int Cpad,
                                                               long rows, float scale) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = rows * Cpad, stride = (long)gridDim.x * 256;
    for (; i < total; i += stride) {
        const long rrow = i / Cpad;
        const int c = (int)(i - rrow * Cpad);
        float v = c < C ? x[rrow * C + c] : 0.0f;
        float q = rintf(v * scale);
        q = q < -128.0f ? -128.0f : (q > 127.0f ? 127.0f : q);
        y[i] = (int8_t)(int)q;
    }
}

}  // namespace fq

using namespace fq;

extern "C" int fq_quantize_i8_nhwc(const float* x_nchw, int8_t* y_nhwc, int N, int C, int HW, int Cpad, int ib,
                                   fq_stream_t stream) {
    if (N < 0 || C <= 0 || HW < 0 || Cpad < C || (Cpad & 3) || ib < -120 || ib > 120) return FQ_ERR_INVALID_ARG;
    if (N == 0 || HW == 0) return FQ_OK;
    if (!x_nchw || !y_nhwc) return FQ_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(y_nhwc) & 3u) return FQ_ERR_INVALID_ARG;
    hipStream_t st = as_stream(stream);
    const float scale = ldexpf(1.0f, ib);
    if (HW == 1) {
        const long total = (long)N * Cpad;
        long g = (total + 255) / 256;
        if (g > kCUs * 16) g = kCUs * 16;
        hipLaunchKernelGGL(quantize_i8_rows_kernel, dim3((unsigned)g), dim3(256), 0, st, x_nchw, y_nhwc, C, Cpad, (long)N, scale);
    } else if (C <= 4 && Cpad == 16 && (reinterpret_cast<uintptr_t>(y_nhwc) & 15u) == 0) {
        const long total = (long)N * HW;
        long g = (total + 255) / 256;
        if (g > kCUs * 32) g = kCUs * 32;
        hipLaunchKernelGGL(quantize_i8_nhwc_smallc_kernel, dim3((unsigned)g), dim3(256), 0, st, x_nchw, y_nhwc, C, HW, total, scale);
    } else {
        if (N > 65535) return FQ_ERR_UNSUPPORTED;
        dim3 grid((HW + 63) / 64, (Cpad + 63) / 64, N);
        const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(x_nchw) & 15u) == 0);
        if (vec)
            hipLaunchKernelGGL(quantize_i8_nhwc_kernel<true>, grid, dim3(256), 0, st, x_nchw, y_nhwc, C, HW, Cpad, scale);
        else
            hipLaunchKernelGGL(quantize_i8_nhwc_kernel<false>, grid, dim3(256), 0, st, x_nchw, y_nhwc, C, HW, Cpad, scale);
    }
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_quantize_i8_unfold_w(const float* x_nchw, int8_t* y, int N, int C, int H, int W, int S, int stride_w,
                                       int pad_w, int dil_w, int Cpad2, int ib, fq_stream_t stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || S <= 0 || stride_w <= 0 || pad_w < 0 || dil_w <= 0 || ib < -120 || ib > 120)
        return FQ_ERR_INVALID_ARG;
    if (Cpad2 < S * C || (Cpad2 & 15)) return FQ_ERR_INVALID_ARG;
    const int Q = (W + 2 * pad_w - dil_w * (S - 1) - 1) / stride_w + 1;
    if (Q <= 0) return FQ_ERR_INVALID_ARG;
    if (N == 0) return FQ_OK;
    if (!x_nchw || !y || (reinterpret_cast<uintptr_t>(y) & 15u)) return FQ_ERR_INVALID_ARG;
    const long total = (long)N * H * Q;
    long g = (total + 255) / 256;
    if (g > kCUs * 32) g = kCUs * 32;
    const float scale = ldexpf(1.0f, ib);
    hipStream_t st = as_stream(stream);
    if (C <= 4 && S <= 8 && S * C <= 32 && Cpad2 <= 32) {
        switch (C) {
            case 1: hipLaunchKernelGGL(quantize_i8_unfold_w_small_kernel<1>, dim3((unsigned)g), dim3(256), 0, st, x_nchw, y, H, W, S,
                                       stride_w, pad_w, dil_w, Q, Cpad2, total, scale); break;
            case 2: hipLaunchKernelGGL(quantize_i8_unfold_w_small_kernel<2>, dim3((unsigned)g), dim3(256), 0, st, x_nchw, y, H, W, S,
                                       stride_w, pad_w, dil_w, Q, Cpad2, total, scale); break;
            case 3: hipLaunchKernelGGL(quantize_i8_unfold_w_small_kernel<3>, dim3((unsigned)g), dim3(256), 0, st, x_nchw, y, H, W, S,
                                       stride_w, pad_w, dil_w, Q, Cpad2, total, scale); break;
            default: hipLaunchKernelGGL(quantize_i8_unfold_w_small_kernel<4>, dim3((unsigned)g), dim3(256), 0, st, x_nchw, y, H, W, S,
                                        stride_w, pad_w, dil_w, Q, Cpad2, total, scale); break;
        }
    } else {
        hipLaunchKernelGGL(quantize_i8_unfold_w_kernel, dim3((unsigned)g), dim3(256), 0, st, x_nchw, y, C, H, W, S, stride_w, pad_w,
                           dil_w, Q, Cpad2, total, scale);
    }
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

namespace fq {

// FQ_CONV_XCD=0 keeps the plain 2-D grid (A/B timing)
static ConvParams xcd_order(dim3& grid, const ConvParams& p0) {
    static const bool on = [] { const char* e = getenv("FQ_CONV_XCD"); return !(e && e[0] == '0'); }();
    ConvParams p = p0;
    p.xcd_kt = 0;
    p.tiles_m = (int)grid.x;
    if (on && grid.y > 1 && (long)((grid.x + 7) / 8) * 8 * grid.y < 0x7fffffffL) {
        p.xcd_kt = (int)grid.y;
        grid = dim3(((grid.x + 7) / 8) * 8 * grid.y, 1);
    }
    return p;
}

template <int TK, int STAGES>
static void launch_conv_dma_stages(dim3 grid, hipStream_t st, const int8_t* x, const int8_t* w, const float* qbias, float* y,
                                   int8_t* q, const ConvParams& p0) {
    const ConvParams p = xcd_order(grid, p0);
    if (p.res)
        hipLaunchKernelGGL((conv2d_i8_dma_kernel<TK, kOutI8 | kOutAdd, STAGES>), grid, dim3(kConvBlock), 0, st, x, w, qbias, y, q, p);
    else if (y && q)
        hipLaunchKernelGGL((conv2d_i8_dma_kernel<TK, kOutF32 | kOutI8, STAGES>), grid, dim3(kConvBlock), 0, st, x, w, qbias, y, q, p);
    else if (q)
        hipLaunchKernelGGL((conv2d_i8_dma_kernel<TK, kOutI8, STAGES>), grid, dim3(kConvBlock), 0, st, x, w, qbias, y, q, p);
    else
        hipLaunchKernelGGL((conv2d_i8_dma_kernel<TK, kOutF32, STAGES>), grid, dim3(kConvBlock), 0, st, x, w, qbias, y, q, p);
}

template <int TK>
static void launch_conv_dma(dim3 grid, hipStream_t st, const int8_t* x, const int8_t* w, const float* qbias, float* y,
                            int8_t* q, const ConvParams& p) {
    // A ring of 3 takes the DMA latency off the per-step critical path (one wave's step: 1 960 -> 1 700 cycles) but
    // needs 96 / 72 KB of LDS, i.e. one workgroup fewer per CU: it pays exactly when the launch cannot put a
    // second workgroup on the CUs anyway (the 7x7 layers at batch 128, most layers at small batch) and costs
    // 20-50 % otherwise (measured per layer, scripts/conv_bench.py with FQ_CONV_STAGES=2/3).
    static const int stages_env = [] { const char* e = getenv("FQ_CONV_STAGES"); return e ? atoi(e) : 0; }();
    const bool three = stages_env ? stages_env == 3 : (long)grid.x * grid.y <= kCUs;
    if (three) launch_conv_dma_stages<TK, 3>(grid, st, x, w, qbias, y, q, p);
    else launch_conv_dma_stages<TK, 2>(grid, st, x, w, qbias, y, q, p);
    note_conv_variant(three ? kVarDma3 : kVarDma2, TK);
}

template <int TK, int kPath>
static void launch_conv_tile(dim3 grid, hipStream_t st, const int8_t* x, const int8_t* w, const float* qbias, float* y,
                             int8_t* q, const ConvParams& p0) {
    const ConvParams p = xcd_order(grid, p0);
    note_conv_variant(kPath == kPathC128 ? kVarTileC128 : (kPath == kPathC64 ? kVarTileC64 : kVarTileGeneral), TK);
    // (kOutResEarly -- the residual requested behind the first K-step's operand loads -- is kept in the kernel as a template flag
    //  and NOT instantiated: measured twice in round 4, the second time without the scratch traffic that spoilt the first, it is
    //  slower everywhere inside the network at 256 images: 86 -> 97 us on the 28 x 28 tail, 63.6 -> 74.5 on the 14 x 14 ones,
    //  38.8 -> 41.6 at 7 x 7; 333 -> 374 us on the 56 x 56 tail from HBM.  The burst of 32 KB per workgroup ahead of the other
    //  resident workgroups' operand requests delays THEIR matrix work by more than it saves this one.)
    if (p.res)
        hipLaunchKernelGGL((conv2d_i8_kernel<TK, kPath, kOutI8 | kOutAdd>), grid, dim3(kConvBlock), 0, st, x, w, qbias, y, q, p);
    else if (y && q)
        hipLaunchKernelGGL((conv2d_i8_kernel<TK, kPath, kOutF32 | kOutI8>), grid, dim3(kConvBlock), 0, st, x, w, qbias, y, q, p);
    else if (q)
        hipLaunchKernelGGL((conv2d_i8_kernel<TK, kPath, kOutI8>), grid, dim3(kConvBlock), 0, st, x, w, qbias, y, q, p);
    else
        hipLaunchKernelGGL((conv2d_i8_kernel<TK, kPath, kOutF32>), grid, dim3(kConvBlock), 0, st, x, w, qbias, y, q, p);
}

// Dynamic LDS a kernel of the halo family may ask for and still run two workgroups per CU (160 KB): half the CU's LDS minus what
// the kernel declares STATICALLY beside its dynamic carve-out -- sLH[2][TK] ints, the merged clamp bounds of the integer tail --
// and the allocation granule.  (The budget used to ignore the static part; no shape fell into the gap, but one layout change
// could have halved the occupancy silently.)
static constexpr size_t two_per_cu_lds(int tk) { return (size_t)80 * 1024 - 64 - (size_t)8 * tk; }

// the halo form of the 3 x 3 layers; false when the layer is not of that shape
template <int TK>
static bool launch_conv_halo(hipStream_t st, const int8_t* x, const int8_t* w, const float* qbias, float* y, int8_t* q,
                             const ConvParams& p0) {
    static const bool on = [] { const char* e = getenv("FQ_CONV_HALO"); return !(e && e[0] == '0'); }();
    if (!on || p0.R != 3 || p0.S != 3 || p0.stride_h != 1 || p0.stride_w != 1 || p0.pad_h != 1 || p0.pad_w != 1 || p0.dil_h != 1 ||
        p0.dil_w != 1 || (p0.C & 127) || (p0.K % TK) || p0.res)
        return false;
    static const int st_env = [] { const char* e = getenv("FQ_HALO_STAGES"); return e ? atoi(e) : 0; }();
    static const int np_env = [] { const char* e = getenv("FQ_HALO_NP"); return e ? atoi(e) : 0; }();
    // 256-pixel tiles on eight waves (conv3x3_i8_halo8_kernel): FQ_HALO8=0 keeps the 128-pixel form
    static const int eight = [] { const char* e = getenv("FQ_HALO8"); return e ? atoi(e) : 1; }();
    if (eight && !np_env && (eight > 1 || (long)((p0.M + 255) / 256) * (p0.K / TK) >= kCUs)) {
        HaloParams hp8;
        hp8.slab_rows = (2 * 128 + 2 + 2 * p0.W + 7) & ~7;
        hp8.total_pixels = p0.N * p0.H * p0.W;
        const size_t fixed8 = (size_t)TK * 8 + 128 + (size_t)hp8.slab_rows * 128;
        const int stages8 = st_env ? st_env : 2;
        const size_t lds8 = (size_t)stages8 * TK * 128 + fixed8;
        if (lds8 <= two_per_cu_lds(TK) && (stages8 == 2 || stages8 == 3)) {
            dim3 grid8((unsigned)(((long)p0.M + 255) / 256), (unsigned)(p0.K / TK));
            const ConvParams p8 = xcd_order(grid8, p0);
#define FQ_HALO8_K(OUT, STG)                                                                                             \
    do {                                                                                                                 \
        auto k = conv3x3_i8_halo8_kernel<TK, OUT, STG>;                                                                  \
        static bool lds_ok[kMaxDevices] = {};                                                                            \
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), 80 * 1024, lds_ok)) return false;                      \
        hipLaunchKernelGGL(k, grid8, dim3(2 * kConvBlock), lds8, st, x, w, qbias, y, q, p8, hp8);                        \
    } while (0)
#define FQ_HALO8(OUT) do { if (stages8 == 3) FQ_HALO8_K(OUT, 3); else FQ_HALO8_K(OUT, 2); } while (0)
            if (y && q) FQ_HALO8(kOutF32 | kOutI8);
            else if (q) FQ_HALO8(kOutI8);
            else FQ_HALO8(kOutF32);
#undef FQ_HALO8
#undef FQ_HALO8_K
            note_conv_variant(kVarHalo8, TK);
            return true;
        }
    }
    // 128-pixel tiles.  (FQ_HALO_NP=2: 256-pixel tiles, two sub-tiles per wave -- bit-exact, fewer LDS reads and barriers per
    // MFMA, and 20-35 % SLOWER on ResNet-50's layers at 256 images: 48.7 vs 39.9 us on 256 -> 256 @14x14.)
    const int np = np_env ? np_env : 1;
    HaloParams hp;
    hp.slab_rows = (128 * np + 2 + 2 * p0.W + 7) & ~7;
    hp.total_pixels = p0.N * p0.H * p0.W;
    const size_t fixed = (size_t)TK * 8 + 128 + (size_t)hp.slab_rows * 128;
    const int stages = st_env ? st_env : ((size_t)3 * TK * 128 + fixed <= two_per_cu_lds(TK) ? 3 : 2);
    const size_t lds = (size_t)stages * TK * 128 + fixed;
    if (lds > two_per_cu_lds(TK) || (stages != 2 && stages != 3) || (np != 1 && np != 2)) return false;   // two workgroups per CU
    dim3 grid((unsigned)(((long)p0.M + 128 * np - 1) / (128 * np)), (unsigned)(p0.K / TK));
    const ConvParams p = xcd_order(grid, p0);
#define FQ_HALO_K(OUT, STG, NPX)                                                                                         \
    do {                                                                                                                 \
        auto k = conv3x3_i8_halo_kernel<TK, OUT, STG, NPX>;                                                              \
        static bool lds_ok[kMaxDevices] = {};                                                                            \
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), 80 * 1024, lds_ok)) return false;                      \
        hipLaunchKernelGGL(k, grid, dim3(kConvBlock), lds, st, x, w, qbias, y, q, p, hp);                                \
    } while (0)
#define FQ_HALO(OUT)                                                                                                     \
    do {                                                                                                                 \
        if (np == 2) { if (stages == 3) FQ_HALO_K(OUT, 3, 2); else FQ_HALO_K(OUT, 2, 2); }                               \
        else { if (stages == 3) FQ_HALO_K(OUT, 3, 1); else FQ_HALO_K(OUT, 2, 1); }                                      \
    } while (0)
    if (y && q) FQ_HALO(kOutF32 | kOutI8);
    else if (q) FQ_HALO(kOutI8);
    else FQ_HALO(kOutF32);
#undef FQ_HALO
#undef FQ_HALO_K
    note_conv_variant(kVarHalo, TK);
    return true;
}

static bool launch_conv_c64(hipStream_t st, const int8_t* x, const int8_t* w, const float* qbias, float* y, int8_t* q,
                            const ConvParams& p0) {
    static const bool on = [] { const char* e = getenv("FQ_CONV_C64_HALO"); return !(e && e[0] == '0'); }();
    if (!on || p0.R != 3 || p0.S != 3 || p0.stride_h != 1 || p0.stride_w != 1 || p0.pad_h != 1 || p0.pad_w != 1 || p0.dil_h != 1 ||
        p0.dil_w != 1 || p0.C != 64 || p0.K > 64 || p0.res)
        return false;
    C64Params cp;
    cp.slab_rows = (130 + 2 * p0.W + 15) & ~15;
    cp.total_pixels = p0.N * p0.H * p0.W;
    cp.tiles = (p0.M + kTP - 1) / kTP;
    static const bool xcd_order = [] { const char* e = getenv("FQ_C64_XCD"); return !(e && e[0] == '0'); }();
    cp.xcd_chunk = xcd_order ? (cp.tiles + 7) / 8 : 0;
    const size_t lds = (size_t)9 * 64 * 64 + 64 * 8 + 64 + (size_t)kTP * 80 + (size_t)2 * cp.slab_rows * 64;
    if (lds > two_per_cu_lds(64)) return false;            // two workgroups per CU
    ConvParams p = p0;
    p.xcd_kt = 0; p.tiles_m = cp.tiles;
    static const int per_cu = [] { const char* e = getenv("FQ_C64_WG_PER_CU"); return e ? atoi(e) : 2; }();
    unsigned grid = (unsigned)(kCUs * per_cu);
    if ((long)grid > cp.tiles) grid = (unsigned)cp.tiles;
#define FQ_C64(OUT)                                                                                                      \
    do {                                                                                                                 \
        auto k = conv3x3_i8_c64_kernel<OUT>;                                                                             \
        static bool lds_ok[kMaxDevices] = {};                                                                            \
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), 80 * 1024, lds_ok)) return false;                      \
        hipLaunchKernelGGL(k, dim3(grid), dim3(kConvBlock), lds, st, x, w, qbias, y, q, p, cp);                          \
    } while (0)
    if (y && q) FQ_C64(kOutF32 | kOutI8);
    else if (q) FQ_C64(kOutI8);
    else FQ_C64(kOutF32);
#undef FQ_C64
    note_conv_variant(kVarC64Halo, 64);
    return true;
}

struct FusedAdd {                        // residual operand and outputs of a fused NewAdd (res == nullptr: none)
    const void* res = nullptr;
    int res_bytes = 0;
    int16_t* wide = nullptr;
    AddResParams ap = {};
};

// A linear layer (NewLinear: N x C activations, K x C weights, fp32 [N][K] out) as one small workgroup per 32 x 32 output tile
// (round 5).  The tiled kernels give a 256 x 1000 classifier 32 workgroups that each walk the whole reduction through LDS and
// barriers: 24 us for 0.5 GMAC.  Here both operand fragments come straight from L2 into the MFMA's registers (a row of either
// matrix is one lane's 16 bytes per sub-step), eight sub-steps of loads in flight per wave, and the four waves of the workgroup
// take every fourth group of eight sub-steps; their int32 partial tiles meet in LDS (exact: integer sums in any order).
constexpr int kLinWaves = 4;
__global__ __launch_bounds__(64 * kLinWaves) void linear_i8_wave_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                        const float* __restrict__ qbias, float* __restrict__ y,
                                                                        const ConvParams p) {
    __shared__ int part[kLinWaves - 1][16][64];
    const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int k0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(w), 0, p.w_bytes, 0x00020000);
    const unsigned a0 = k0 + r < p.K ? (unsigned)(k0 + r) * (unsigned)p.C + (unsigned)(16 * half) : kOutOfRange;
    const unsigned b0 = m0 + r < p.M ? (unsigned)(m0 + r) * (unsigned)p.C + (unsigned)(16 * half) : kOutOfRange;
    const int steps = (p.C + 31) / 32;
    // (C % 32 == 16: the upper half of the last sub-step lies behind the row -- in the next row, not out of range: request nothing)
    const bool odd = (p.C & 31) != 0;
    v16i acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    for (int g0 = 8 * wave; g0 < steps; g0 += 8 * kLinWaves) {       // groups of eight sub-steps, round robin over the waves
        v4i fa[8], fb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ks = g0 + u;
            const bool dead = ks >= steps || (odd && ks == steps - 1 && half == 1);
            fa[u] = load_act(wr, dead ? kOutOfRange : a0 + (unsigned)(ks * 32));
            fb[u] = load_act(xr, dead ? kOutOfRange : b0 + (unsigned)(ks * 32));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u], fb[u], acc, 0, 0, 0);
    }
    if (wave != 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) part[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int v = 0; v < kLinWaves - 1; ++v)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += part[v][i][lane];
    const int n = m0 + r;
    if (n >= p.M) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = k0 + (i & 3) + 8 * (i >> 2) + 4 * half;
        if (k < p.K) {
            const float b = qbias[k];
            // (a bias of ANY magnitude, as fq.h promises: tail_consts brings it into the range beyond which the output is a bound
            //  whatever the accumulator holds; the unclamped six-instruction form would wrap on a bias near INT_MAX)
            float out;
            if (p.rs) {
                const TailK tk = tail_consts((int)b, p);               // (v_cvt_i32_f32 saturates)
                out = (float)conv_tail_k(acc[i], tk.B, tk.lo, tk.hi, p.rs) * p.inv_ob;
            } else {
                out = conv_tail(acc[i], b, p);
            }
            y[(size_t)n * p.K + k] = out;
        }
    }
}

static int conv2d_i8_dispatch(const int8_t* x_nhwc, const int8_t* w_krsc, const float* qbias, float* y_nchw, int8_t* q_nhwc,
                              int Kpad, int relu, const FusedAdd& fa, int N, int H, int W, int C, int K, int R, int S, int stride_h, int stride_w,
                              int pad_h, int pad_w, int dil_h, int dil_w, int rs, int ob, int bitwidth, fq_stream_t stream) {
    if (!valid_bitwidth(bitwidth) || rs < -120 || rs > 120 || ob < -120 || ob > 120) return FQ_ERR_INVALID_ARG;
    if (N < 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride_h <= 0 || stride_w <= 0 ||
        pad_h < 0 || pad_w < 0 || dil_h <= 0 || dil_w <= 0)
        return FQ_ERR_INVALID_ARG;
    if (C % 16) return FQ_ERR_UNSUPPORTED;                // pad channels to 16 in fq_quantize_i8_nhwc
    if ((q_nhwc || fa.res) && (bitwidth != 8 || Kpad < K || (Kpad & 15))) return FQ_ERR_INVALID_ARG;
    if (fa.res && (y_nchw || relu || (!q_nhwc && !fa.wide))) return FQ_ERR_INVALID_ARG;
    const int P = (H + 2 * pad_h - dil_h * (R - 1) - 1) / stride_h + 1;
    const int Q = (W + 2 * pad_w - dil_w * (S - 1) - 1) / stride_w + 1;
    if (P <= 0 || Q <= 0) return FQ_ERR_INVALID_ARG;
    g_last_conv_variant = kVarNone;
    if (N == 0) return FQ_OK;
    if (!x_nhwc || !w_krsc || !qbias || (!y_nchw && !q_nhwc && !fa.res)) return FQ_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x_nhwc) | reinterpret_cast<uintptr_t>(w_krsc) | reinterpret_cast<uintptr_t>(q_nhwc) |
         reinterpret_cast<uintptr_t>(fa.res) | reinterpret_cast<uintptr_t>(fa.wide)) & 15u)
        return FQ_ERR_INVALID_ARG;
    const long M = (long)N * P * Q;
    if (M > 0x7fffffffL || (long)R * S * C / 16 > 0x7fffffffL) return FQ_ERR_UNSUPPORTED;
    if ((long)N * H * W * C >= 0x7fffffffL || (long)K * R * S * C >= 0x7fffffffL || (long)K * P * Q > 0x1fffffffL)
        return FQ_ERR_UNSUPPORTED;                        // 32-bit buffer / per-row offsets inside the kernel
    ConvParams p;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.P = P; p.Q = Q;
    p.stride_h = stride_h; p.stride_w = stride_w; p.pad_h = pad_h; p.pad_w = pad_w; p.dil_h = dil_h; p.dil_w = dil_w;
    p.M = (int)M; p.c16 = C / 16; p.chunks = R * S * p.c16;
    p.inv_rs = ldexpf(1.0f, -rs); p.inv_ob = ldexpf(1.0f, -ob);
    if (bitwidth == 8) { p.lo = -128.0f; p.hi = 127.0f; p.ilo = -128; p.ihi = 127; }
    else { p.lo = -32768.0f; p.hi = 32767.0f; p.ilo = -32768; p.ihi = 32767; }
    if (relu) p.lo = 0.0f;                                // ReLU commutes with the positive scale 2^-ob
    // integer tail where it is provably the same function (see conv_tail_i)
    // (... and with the bias folded into the rounding constant -- tail_consts: |qb| <= shi - ilo after its clamp, shifted by rs)
    const long qmax = bitwidth == 8 ? 255 : 65535;
    const bool int_tail = rs >= 1 && rs <= 16 && (long)R * S * C * 16384 + 65536 + (qmax << rs) < 0x7fffffffL;
    p.rs = int_tail ? rs : 0;
    p.half_rs = int_tail ? 1 << (rs - 1) : 0;
    p.slo = (int)p.lo; p.shi = (int)p.hi;
    p.Kpad = (q_nhwc || fa.res) ? Kpad : 0;
    p.res = fa.res; p.res_bytes = fa.res_bytes; p.wide = fa.wide; p.ap = fa.ap;
    if ((q_nhwc || fa.res) && (long)M * Kpad >= 0x3fffffffL) return FQ_ERR_UNSUPPORTED;     // 32-bit element offsets into the [M][Kpad] tensors
    p.out_elems = (q_nhwc || fa.res) ? (unsigned)((long)M * Kpad) : 0u;
    p.x_bytes = (unsigned)((long)N * H * W * C);
    p.w_bytes = (unsigned)((long)K * R * S * C);
    hipStream_t st = as_stream(stream);
    // the 64 -> 64 3x3 layers: stationary weights, persistent workgroups
    if (launch_conv_c64(st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p)) {
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    }
    // 1x1 layers with an int8 output: the streaming kernel with stationary weights (fq_conv1x1_i8.hip) where it applies
    if (launch_conv1x1_stream(st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p)) {
        note_conv_variant(kVarStream, 0);
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    }
    // a linear layer with an fp32 output: one wave per 32 x 32 tile, operands straight from L2 (FQ_LINEAR_WAVE=0: the tiled kernels)
    static const bool thin = [] { const char* e = getenv("FQ_LINEAR_WAVE"); return !(e && e[0] == '0'); }();
    // (pad 0, dilation 1: with padding and a stride >= 3 the one output pixel would sample the zero border, not x)
    if (thin && H == 1 && W == 1 && R == 1 && S == 1 && P == 1 && Q == 1 && pad_h == 0 && pad_w == 0 && dil_h == 1 && dil_w == 1 && y_nchw && !q_nhwc && !fa.res && (M + 31) / 32 <= 65535) {
        hipLaunchKernelGGL(linear_i8_wave_kernel, dim3((unsigned)((K + 31) / 32), (unsigned)((M + 31) / 32)), dim3(64 * kLinWaves), 0, st, x_nhwc, w_krsc,
                           qbias, y_nchw, p);
        note_conv_variant(kVarLinearWave, 32);
        FQ_LAUNCH_CHECK();
        return FQ_OK;
    }
    const unsigned gx = (unsigned)((M + kTP - 1) / kTP);
    // 64-row tiles when the output is narrow, or when 128-row tiles would not even give one workgroup per CU
    const long wg128 = (long)gx * ((K + 127) / 128);
    // Deep reductions (>= 8 K-steps: the 3x3 layers and the wide 1x1 reductions) are bound by L1/L2 request
    // throughput and run faster with coalesced LDS-DMA staging; shallow ones are output-bound, and there the
    // register-staged kernel's smaller LDS footprint (3 workgroups per CU instead of 2) wins.  Measured per
    // layer on ResNet-50 at batch 128 (scripts/conv_bench.py); FQ_CONV_DMA=0/1 forces one of them.
    static const int dma_env = [] { const char* e = getenv("FQ_CONV_DMA"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    const bool use_dma = dma_env < 0 ? (p.chunks >> 3) >= 8 : dma_env == 1;
    static const int force_tk = [] { const char* e = getenv("FQ_CONV_TK"); return e ? atoi(e) : 0; }();
    static const bool use_c64 = [] { const char* e = getenv("FQ_CONV_C64"); return !(e && e[0] == '0'); }();
    if ((K <= 64 || wg128 < kCUs || force_tk == 64) && force_tk != 128) {
        if (C % 128 == 0 && K % 64 == 0 && use_dma && launch_conv_halo<64>(st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p))
            ;
        else if (C % 128 == 0 && K % 64 == 0 && use_dma)
            launch_conv_dma<64>(dim3(gx, K / 64), st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p);
        else if (C % 128 == 0 && K % 64 == 0)
            launch_conv_tile<64, kPathC128>(dim3(gx, K / 64), st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p);
        else if (C == 64 && use_c64)
            launch_conv_tile<64, kPathC64>(dim3(gx, (K + 63) / 64), st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p);
        else
            launch_conv_tile<64, kPathGeneral>(dim3(gx, (K + 63) / 64), st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p);
    } else {
        if (C % 128 == 0 && K % 128 == 0 && use_dma && launch_conv_halo<128>(st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p))
            ;
        else if (C % 128 == 0 && K % 128 == 0 && use_dma)
            launch_conv_dma<128>(dim3(gx, K / 128), st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p);
        else if (C % 128 == 0 && K % 128 == 0)
            launch_conv_tile<128, kPathC128>(dim3(gx, K / 128), st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p);
        else if (C == 64 && use_c64)
            launch_conv_tile<128, kPathC64>(dim3(gx, (K + 127) / 128), st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p);
        else
            launch_conv_tile<128, kPathGeneral>(dim3(gx, (K + 127) / 128), st, x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, p);
    }
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

}  // namespace fq

extern "C" int fq_conv2d_i8(const int8_t* x_nhwc, const int8_t* w_krsc, const float* qbias, float* y_nchw, int N, int H,
                            int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                            int dil_h, int dil_w, int rs, int ob, int bitwidth, fq_stream_t stream) {
    if (!y_nchw && N > 0) return FQ_ERR_INVALID_ARG;
    return conv2d_i8_dispatch(x_nhwc, w_krsc, qbias, y_nchw, nullptr, 0, 0, FusedAdd{}, N, H, W, C, K, R, S, stride_h, stride_w, pad_h,
                              pad_w, dil_h, dil_w, rs, ob, bitwidth, stream);
}

extern "C" int fq_conv2d_i8_resident(const int8_t* x_nhwc, const int8_t* w_krsc, const float* qbias, float* y_nchw,
                                     int8_t* q_nhwc, int Kpad, int relu, int N, int H, int W, int C, int K, int R, int S,
                                     int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int rs, int ob,
                                     fq_stream_t stream) {
    return conv2d_i8_dispatch(x_nhwc, w_krsc, qbias, y_nchw, q_nhwc, Kpad, relu, FusedAdd{}, N, H, W, C, K, R, S, stride_h, stride_w, pad_h,
                              pad_w, dil_h, dil_w, rs, ob, 8, stream);
}

extern "C" int fq_conv2d_i8_add_resident(const int8_t* x_nhwc, const int8_t* w_krsc, const float* qbias, const void* res,
                                         int res_bytes, int g_res, int16_t* wide, int g_wide, int8_t* narrow, int ib, int relu,
                                         int Kpad, int N, int H, int W, int C, int K, int R, int S, int stride_h, int stride_w,
                                         int pad_h, int pad_w, int dil_h, int dil_w, int rs, int ob, fq_stream_t stream) {
    if (!res || (res_bytes != 1 && res_bytes != 2)) return FQ_ERR_INVALID_ARG;
    FusedAdd fa;
    fa.res = res; fa.res_bytes = res_bytes; fa.wide = wide;
    const int rc = make_add_params(ob, g_res, g_wide, wide != nullptr, ib, relu, &fa.ap);
    if (rc != FQ_OK) return rc;
    return conv2d_i8_dispatch(x_nhwc, w_krsc, qbias, nullptr, narrow, Kpad, 0, fa, N, H, W, C, K, R, S, stride_h, stride_w, pad_h,
                              pad_w, dil_h, dil_w, rs, ob, 8, stream);
}

#ifdef FQ_CONV_TRACE
extern "C" int fq_debug_read_trace(unsigned long long* host_out) {
    FQ_HIP_CHECK(hipDeviceSynchronize());
    FQ_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(fq::g_trace), sizeof(unsigned long long) * 8192));
    return FQ_OK;
}
#endif
