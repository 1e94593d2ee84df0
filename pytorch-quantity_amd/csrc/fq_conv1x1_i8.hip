// fq_conv1x1_i8.hip -- the 1x1 integer convolutions of NewConv2d (new_quantity_op.py:124-133) as a STREAMING kernel:
// weights stationary in LDS, persistent workgroups, waves that never meet at a barrier, outputs stored straight from the
// accumulator registers.
//
// Why a second kernel.  36 of ResNet-50's 53 convolutions are 1x1, and at 256 images per forward 28 of them are bound by
// their operand / result bytes, not by the matrix pipe (bench.py roofline_int8_conv; scripts/int8_layer_table.py): the 16
// conv3 + NewAdd launches alone move 7 of the forward's 10 GB (the exact int16 residual stream).  The general implicit-GEMM
// kernel (fq_conv_i8.hip) runs them as one workgroup per 128 x 128 tile: load operands -> a K loop of one or two steps ->
// epilogue through LDS -> stores, with two workgroup barriers in the epilogue and the weight tile fetched again for every
// pixel tile.  Its phases follow each other inside a workgroup, so a CU has few bytes in flight (4.1-5.8 TB/s on the
// residual stream, 0.6-1.2 POP/s on the reductions).  Here:
//
//   * a workgroup owns ONE output-channel tile (TK = 32 MT channels) for its whole life and walks over pixel tiles; its
//     weights (TK x C bytes, <= 64 KB) are fetched once, by LDS-DMA, and stay in LDS (read-only after one barrier);
//   * each wave owns 32 pixels of the workgroup's pixel tile and everything that belongs to them: it requests its own
//     activation rows D steps ahead (buffer_load ... lds, full 64 / 128-byte rows, ring of D + 1 slots per wave), waits
//     with a counted vmcnt, multiplies, and stores.  No other wave ever reads those rows, so the loop has NO barrier;
//   * MFMA row r of a 32-row tile holds output channel 16 ((r >> 2) & 1) + 4 (r >> 3) + (r & 3) of that tile (a permutation
//     applied where the weight rows are fetched), so that accumulator register i of lane (pixel p, half h) is channel
//     16 h + i: a lane ends up with 16 CONSECUTIVE channels of its pixel -- one 16-byte int8 store (two for the int16 sum,
//     two 16-byte loads of the int16 residual) straight from registers: no LDS transpose, no barrier in the epilogue;
//   * the residual of the NEXT pixel tile is requested before this tile's epilogue, and all addresses are a per-lane
//     32-bit offset fixed for the kernel's life plus a scalar tile offset: no vector address arithmetic per tile.
//
// Taken for: R = S = 1, no padding, stride 1 or 2, C = 64 or a multiple of 128, K = Kpad a multiple of TK, int8 NHWC output
// (with or without the fused NewAdd), integer tail.  Everything else keeps fq_conv_i8.hip.  Same integers, bit for bit
// (tests/test_gpu_conv_i8.py, scripts/conv_fuzz.py run both kernels against the CPU oracle).
#include <utility>

#include "fq_conv_i8_common.h"

namespace fq {
namespace {

struct StreamParams {
    int nsteps;                          // C / SB
    int kt;                              // K / TK output-channel tiles
    int ns;                              // pixel-tile streams (workgroups per channel tile)
    int npt;                             // pixel tiles of 32 * NW pixels
    unsigned out_bytes;                  // M * Kpad
    float inv_pq, inv_q;                 // stride 2: pixel -> (image, row, column) by reciprocal + fix-up (M < 2^24)
#ifdef FQ_STREAM_ABLATE
    int ablate;                          // debug build only (timing; wrong results): 1 stores dropped, 2 residual not loaded, 4 no activation DMA, 8 no add arithmetic
#endif
};
#ifdef FQ_STREAM_ABLATE
#define FQ_SA(bit) (sp.ablate & (bit))
#else
#define FQ_SA(bit) false
#endif

// s_waitcnt vmcnt(BASE + cnt * OPE) for a wave-uniform cnt in 0 .. D (immediates; vmcnt is a 6-bit field: a smaller
// count than the true one only waits longer)
template <int BASE, int OPE, int D, int I = 0>
__device__ __forceinline__ void wait_vmcnt_steps(int cnt) {
    constexpr int n = BASE + I * OPE < 63 ? BASE + I * OPE : 63;
    if constexpr (I == D) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory");
    } else {
        if (cnt == I) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory");
        else wait_vmcnt_steps<BASE, OPE, D, I + 1>(cnt);
    }
}

// 16 conv results of one lane, still one per register (operand x of the fused NewAdd: no pack / unpack round trip)
struct Regs16 {
    int v[16];
    __device__ __forceinline__ int geti(int i) const { return v[i]; }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
};

// LDS-DMA with a memory clobber: the compiler keeps its own LDS reads and global stores on their side of it, so the
// counted waits below can rely on program order.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma_to_lds_ordered(rsrc_words rsrc, unsigned lds_base, unsigned voffset, unsigned soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_base), "v"(voffset), "s"(rsrc), "s"(soffset) : "m0", "memory");
}
#pragma clang diagnostic pop

// A 16-byte buffer load the compiler does not count: its own s_waitcnt bookkeeping cannot see the LDS-DMA requests above,
// so for a load it DOES see it waits with too small a count -- and in this kernel that meant "until the stores of the tile
// before are acknowledged", once per tile (vmcnt retires loads and stores in issue order).  The destination is valid only
// behind res_wait() below; nothing may read or copy it earlier (the tile loop ping-pongs two register sets instead of
// rotating one).
__device__ __forceinline__ void load16_uncounted(v4i_r& dst, rsrc_words rsrc, unsigned voffset, unsigned soffset) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}

template <int SB> __device__ __forceinline__ int swz_row(int row) { return SB == 128 ? (row >> 1) & 7 : (row >> 2) & 3; }

// SB: bytes of the reduction axis per step (64: the C = 64 layers, one step; 128 otherwise).  MT: 32-row output-channel
// tiles per wave.  NW: waves per workgroup.  D: steps requested ahead (ring of D + 1 slots per wave).
// kAdd / kRes16: the fused NewAdd and the width of its residual operand.  kGather: stride 2 (per-pixel input offsets).
#ifndef FQ_STREAM_WAVES
#define FQ_STREAM_WAVES 2
#endif
template <int SB, int MT, int NW, int D, bool kAdd, bool kRes16, bool kGather>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(FQ_STREAM_WAVES, FQ_STREAM_WAVES))) void conv1x1_i8_stream_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                   const float* __restrict__ qbias, int8_t* __restrict__ q,
                                                                   const ConvParams p, const StreamParams sp) {
    constexpr int TK = 32 * MT, S = D + 1;
    constexpr int KS = SB / 32;                           // MFMA sub-steps (32 bytes of the reduction axis each) per step
    constexpr int CPR = SB / 16;                          // 16-byte chunks per LDS row
    constexpr int RPI = 64 / CPR;                         // rows one DMA instruction (64 lanes x 16 bytes) covers
    constexpr int L = 32 / RPI;                           // DMA instructions per wave and step (32 pixel rows)
    // vector-memory operations one wave issues per pixel tile besides the DMA: residual loads, and stores -- always all three
    // of the fused add (an output the launch does not want gets an out-of-range offset: dropped by the hardware, counted
    // by vmcnt), so that every wait below is an immediate
    constexpr int R_OPS = kAdd ? MT * (kRes16 ? 2 : 1) : 0, ST_OPS = kAdd ? 3 * MT : MT;
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    // [weights: nsteps x TK rows x SB][activation rings: NW waves x S slots x 32 rows x SB][bias: TK ints]
    // [output transposition: NW waves x 32 pixel rows x (TK + 16) bytes]
    constexpr int OP = TK + 16;                           // pitch of a pixel row of the int8 conv result (16-byte aligned, conflict-free)
    const int a_bytes = sp.nsteps * TK * SB;
    int* const sBias = reinterpret_cast<int*>(smem + a_bytes + NW * S * 32 * SB);
    int8_t* const sOut = smem + a_bytes + NW * S * 32 * SB + TK * 4 ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, prow = lane & 31;
    // Workgroups are dealt to the 8 XCDs round robin (g & 7): the kt channel-tile workgroups of one pixel stream are
    // consecutive on ONE XCD and walk the same pixel tiles at the same pace, so that XCD's L2 serves the activation tile
    // to all of them from one HBM fetch.
    const int g = blockIdx.x, u = g >> 3;
    const int kti = u % sp.kt;
    const int stream = (u / sp.kt) * 8 + (g & 7);
    const int k0 = kti * TK;
    const int ntiles = stream < sp.npt ? (sp.npt - stream + sp.ns - 1) / sp.ns : 0;

    // ---- weights: rows of this channel tile, permuted (header), all C bytes, once ----
    const rsrc_words wr = make_rsrc_words(w, p.w_bytes);
    {
        constexpr int IPS = TK / RPI;                     // DMA instructions per step of the weight tile
        const int total = sp.nsteps * IPS;
        for (int idx = wave; idx < total; idx += NW) {
            const int t = idx / IPS, j = idx - t * IPS;
            const int row = j * RPI + lane / CPR;         // MFMA row of the tile
            const int r5 = row & 31;
            const int ch = k0 + (row & ~31) + 16 * ((r5 >> 2) & 1) + 4 * (r5 >> 3) + (r5 & 3);
            const unsigned vo = (unsigned)ch * (unsigned)p.C + (unsigned)(((lane % CPR) ^ swz_row<SB>(row)) * 16);
            dma_to_lds_ordered(wr, lds_offset(smem + (t * TK + j * RPI) * SB), vo, (unsigned)(t * SB));
        }
        if (tid < TK) sBias[tid] = (int)qbias[k0 + tid];  // integer valued by contract
    }

    // ---- per-lane constants of the activation stream ----
    const rsrc_words xr = make_rsrc_words(x, p.x_bytes);
    int xrow[L];                                          // pixel row (0 .. 32 NW - 1) this lane fetches in DMA j
    unsigned xchunk[L];                                   // byte offset of its (swizzled) 16-byte chunk inside the step
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int row = j * RPI + lane / CPR;
        xrow[j] = 32 * wave + row;
        xchunk[j] = (unsigned)(((lane % CPR) ^ swz_row<SB>(row)) * 16);
    }
    // LDS byte offsets (from smem): the weights at 0, this wave's ring behind them
    const unsigned b_off = (unsigned)(a_bytes + wave * S * 32 * SB);
    const unsigned lds0 = lds_offset(smem);
    // operand fragments: lane (row = prow, chunk 2 ks + half) of a 32-row block
    unsigned frag_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) frag_off[ks] = (unsigned)(prow * SB + (((2 * ks + half) ^ swz_row<SB>(prow)) * 16));

    // ---- outputs / residual: per-lane offset of (pixel row of the tile, first channel of the lane) ----
    const __amdgpu_buffer_rsrc_t nr = __builtin_amdgcn_make_buffer_rsrc(q, 0, sp.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wdr = __builtin_amdgcn_make_buffer_rsrc(p.wide, 0, 2u * sp.out_bytes, 0x00020000);
    const rsrc_words rr = make_rsrc_words(p.res, (kRes16 ? 2u : 1u) * sp.out_bytes);
    // Store layout: lane = (pixel row, 16-channel group) so that the CPP lanes of a pixel write TK contiguous bytes (2 TK of
    // the int16 sum): whole 128-byte lines per instruction.  (Stores straight from the accumulator layout -- lane = pixel,
    // 16 consecutive channels per tile -- were tried first: bit-exact and 15-25 % slower than the general kernel, four
    // pixels = four different lines per quad of lanes make the address unit spend four cycles per quad.)
    constexpr int CPP = TK / 16;                          // 16-channel groups per pixel row of this channel tile
    constexpr int PPI = 64 / CPP;                         // pixel rows one instruction covers
    constexpr int NJ = 32 / PPI;                          // = MT items of 16 channels per lane and pixel tile
    const int ochunk = lane % CPP;
    int orow[NJ];
    unsigned ovo[NJ];                                     // int8 byte offset of (pixel row, first channel) inside a pixel tile
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        orow[j] = 32 * wave + lane / CPP + PPI * j;
        ovo[j] = (unsigned)orow[j] * (unsigned)p.Kpad + (unsigned)(k0 + 16 * ochunk);
    }
    int8_t* const my_out = sOut + wave * 32 * OP;
    const bool want_wide = kAdd && p.wide != nullptr, want_narrow = q != nullptr;

    // ---- request side: (tile, step) of the next DMA group, D steps ahead of the arithmetic ----
    int it = 0, istep = 0, islot = 0;
    unsigned xv[L];                                       // this lane's offsets for the tile being requested
    auto tile_offsets = [&]() {                           // per tile: the range check (and the pixel mapping of a stride)
        const int px0 = (stream + it * sp.ns) * (32 * NW);
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int m = px0 + xrow[j];
            const bool live = it < ntiles && m < p.M && !FQ_SA(4);
            if (kGather) {
                const int mm = live ? m : 0, PQ = p.P * p.Q;
                int n = (int)((float)mm * sp.inv_pq);     // reciprocal + fix-up: exact for M < 2^24 (host check)
                int pq = mm - n * PQ;
                if (pq < 0) { pq += PQ; --n; }
                if (pq >= PQ) { pq -= PQ; ++n; }
                int oh = (int)((float)pq * sp.inv_q);
                int ow = pq - oh * p.Q;
                if (ow < 0) { ow += p.Q; --oh; }
                if (ow >= p.Q) { ow -= p.Q; ++oh; }
                const unsigned pix = (unsigned)((n * p.H + oh * p.stride_h) * p.W + ow * p.stride_w);
                xv[j] = live ? pix * (unsigned)p.C + xchunk[j] : kOutOfRange;
            } else {
                xv[j] = live ? (unsigned)xrow[j] * (unsigned)p.C + xchunk[j] : kOutOfRange;
            }
        }
    };
    auto issue = [&]() {                                  // DMA of (tile it, step istep) into slot islot; then advance
        if (istep == 0) tile_offsets();
        unsigned so = (unsigned)(istep * SB);             // (the scalar offset is not range checked: dead lanes carry kOutOfRange)
        if (!kGather && it < ntiles) so += (unsigned)((stream + it * sp.ns) * (32 * NW)) * (unsigned)p.C;
#pragma unroll
        for (int j = 0; j < L; ++j)
            dma_to_lds_ordered(xr, lds0 + b_off + (unsigned)(islot * 32 * SB + j * RPI * SB), xv[j], so);
        if (++istep == sp.nsteps) { istep = 0; ++it; }
        islot = islot + 1 == S ? 0 : islot + 1;
    };

    // residual of one pixel tile: MT groups of 16 consecutive channels per lane
    struct Res {
        v4i_r lo[kAdd ? MT : 1], hi[kAdd && kRes16 ? MT : 1];
    };
    auto load_res = [&](Res& r, int tile) {
        if constexpr (kAdd) {
            const int px0 = (stream + tile * sp.ns) * (32 * NW);
            const unsigned so = tile < ntiles ? (unsigned)px0 * (unsigned)p.Kpad : 0u;     // wave-uniform: an SGPR operand
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool live = tile < ntiles && px0 + orow[j] < p.M && !FQ_SA(2);
                if constexpr (kRes16) {
                    const unsigned vo2 = live ? 2u * ovo[j] : kOutOfRange;
                    load16_uncounted(r.lo[j], rr, vo2, 2u * so);
                    load16_uncounted(r.hi[j], rr, vo2 + 16u, 2u * so);
                } else {
                    load16_uncounted(r.lo[j], rr, live ? ovo[j] : kOutOfRange, so);
                }
            }
        }
    };
    // the residual registers of the tile about to be added: valid from here on
    auto res_wait = [&](Res& r, bool first_tile) {
        if constexpr (kAdd) {
            // younger than this tile's residual request: the stores of the tile before, the DMA groups of this tile's steps,
            // the next tile's residual request.  (First tile: its request sits behind the prologue's DMA groups and no
            // stores exist yet -- wait for everything but the next tile's request.)
            if (first_tile) wait_vmcnt_steps<R_OPS, 0, 0>(0);
            else wait_vmcnt_steps<ST_OPS + R_OPS, L, 8>(sp.nsteps < 8 ? sp.nsteps : 8);
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                asm volatile("" : "+v"(r.lo[a]));
                if constexpr (kRes16) asm volatile("" : "+v"(r.hi[a]));
            }
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the weight tile has landed ...
    __syncthreads();                                      // ... and everybody else's, and the bias: the only barrier
#pragma unroll
    for (int d = 0; d < D; ++d) issue();
    Res res_a, res_b;                                     // ping-pong: even tiles add res_a, odd tiles res_b
    load_res(res_a, 0);

    // vmcnt bookkeeping.  Operations complete in issue order; before step s reads its slot, everything up to DMA(s) must
    // be complete.  Younger than DMA(s): the DMAs of steps s+1 .. s+D-1 and, for every LAST step of a tile among the
    // previous D steps, that tile's residual prefetch and stores (issued after that step's own DMA).  `ends` remembers
    // which of the previous steps were last steps.  (The first residual request sits behind the first D DMA groups and is
    // not counted: fewer counted than in flight only waits longer.)
    unsigned ends = 0;
    int cslot = 0;
    v16i acc[MT];
    auto pre_step = [&](bool last) {
        wait_vmcnt_steps<(D - 1) * L, R_OPS + ST_OPS, D>(__builtin_popcount(ends & ((1u << D) - 1u)));
        ends = (ends << 1) | (last ? 1u : 0u);
        issue();                                          // refills the slot the previous step read
    };
    auto mma_step = [&](int t) {
        const int8_t* const bs = smem + b_off + (unsigned)(cslot * 32 * SB);
        const int8_t* const as = smem + (unsigned)(t * TK * SB);
        v4i fb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fb[ks] = *reinterpret_cast<const v4i*>(bs + frag_off[ks]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const v4i fa = *reinterpret_cast<const v4i*>(as + a * 32 * SB + frag_off[ks]);
                acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb[ks], acc[a], 0, 0, 0);
            }
        }
        cslot = cslot + 1 == S ? 0 : cslot + 1;
    };

    auto tile = [&](int ti, Res& res, Res& res_next) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0;
        for (int t = 0; t + 1 < sp.nsteps; ++t) {
            pre_step(false);
            mma_step(t);
        }
        // the last step of the tile: the residual of the NEXT tile is requested in front of this tile's arithmetic
        pre_step(true);
        load_res(res_next, ti + 1);
        mma_step(sp.nsteps - 1);
        res_wait(res, ti == 0);
        // ---- epilogue.  Register i of tile a is channel k0 + 32 a + 16 half + i of this lane's pixel: RightShift + BiasAdd +
        // Sp on it, 16 int8 per tile into this wave's own LDS rows, read back in the store layout.  LDS instructions of one
        // wave execute in order, and no other wave touches these rows: no barrier. ----
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            v4i b4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) b4[i] = *reinterpret_cast<const v4i*>(&sBias[32 * a + 16 * half + 4 * i]);
            int cv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) cv[i] = conv_tail_i(acc[a][i], b4[i >> 2][i & 3], p);
            v4u o;
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = pack4(cv[4 * d], cv[4 * d + 1], cv[4 * d + 2], cv[4 * d + 3]);
            *reinterpret_cast<v4u*>(my_out + prow * OP + 32 * a + 16 * half) = o;
        }
        const int px0 = (stream + ti * sp.ns) * (32 * NW);
        // The tile offset goes into the VECTOR offset of the stores, the scalar offset stays the immediate 0: hipcc (ROCm 7.2)
        // pads "16-byte store, then a write of its data registers" only for stores WITHOUT a scalar-offset register (the
        // rule of older parts), and on gfx950 the hazard exists with one too -- with the offset in an SGPR the next
        // element's v_ashrrev landed in the data of the store in front of it (its last lanes carried a sign mask).
        const unsigned so = (unsigned)px0 * (unsigned)p.Kpad;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool live = px0 + orow[j] < p.M;
            const unsigned vo = live && want_narrow && !FQ_SA(1) ? ovo[j] + so : kOutOfRange;
            const unsigned vo2 = live && want_wide && !FQ_SA(1) ? 2u * (ovo[j] + so) : kOutOfRange;
            Vec16<int8_t> cvv;
            cvv.a = *reinterpret_cast<const v4i_r*>(my_out + (lane / CPP + PPI * j) * OP + 16 * ochunk);
            if constexpr (kAdd) {
                Add16Out o;
                if constexpr (kRes16) {
                    Vec16<int16_t> rv;
                    rv.a = res.lo[j]; rv.b = res.hi[j];
                    o = add_resident_16_regs(cvv, rv, want_wide, want_narrow, p.ap);
                } else {
                    Vec16<int8_t> rv;
                    rv.a = res.lo[j];
                    o = add_resident_16_regs(cvv, rv, want_wide, want_narrow, p.ap);
                }
                __builtin_amdgcn_raw_buffer_store_b128((v4u)o.w0, wdr, (int)vo2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128((v4u)o.w1, wdr, (int)(vo2 + 16u), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128((v4u)o.n, nr, (int)vo, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128((v4u)cvv.a, nr, (int)vo, 0, 0);
            }
        }
    };
    for (int ti = 0; ti < ntiles; ti += 2) {
        tile(ti, res_a, res_b);
        if (ti + 1 < ntiles) tile(ti + 1, res_b, res_a);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // requests past the last tile (all out of range) are done before the LDS is released
}

// LDS a launch needs
template <int SB, int MT, int NW, int D>
constexpr size_t stream_lds_bytes(int nsteps) {
    return (size_t)nsteps * 32 * MT * SB + (size_t)NW * (D + 1) * 32 * SB + 32 * MT * sizeof(int) + (size_t)NW * 32 * (32 * MT + 16);
}

constexpr size_t kLdsPerCU = 160 * 1024;

template <int SB, int MT, int NW, int D, bool kAdd, bool kRes16, bool kGather>
bool launch_variant(hipStream_t st, const int8_t* x, const int8_t* w, const float* qbias, int8_t* q, const ConvParams& p,
                    StreamParams sp) {
    const size_t lds = stream_lds_bytes<SB, MT, NW, D>(sp.nsteps);
    if (lds > kLdsPerCU) return false;
    auto kern = conv1x1_i8_stream_kernel<SB, MT, NW, D, kAdd, kRes16, kGather>;
    static bool lds_ok[kMaxDevices] = {};                 // > 64 KB of dynamic LDS needs the attribute once per kernel and device
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)kLdsPerCU, lds_ok)) return false;
    // workgroups resident per CU: LDS, and 8 waves per CU (two per SIMD: the register budget of the add epilogue)
    int per_cu = (int)(kLdsPerCU / lds);
    const int by_waves = (MT == 2 ? 16 : 8) / NW;          // registers: the 128-channel add epilogue fits two waves per SIMD, the 64-channel one four
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    static const int wg_env = [] { const char* e = getenv("FQ_STREAM_WG_PER_CU"); return e ? atoi(e) : 0; }();
    if (wg_env > 0) per_cu = wg_env;
    sp.kt = p.K / (32 * MT);
    sp.npt = (p.M + 32 * NW - 1) / (32 * NW);
    const int unit = 8 * sp.kt;                           // workgroups of 8 streams (one per XCD) x all channel tiles
    int groups = (kCUs * per_cu) / unit;
    if (groups < 1) groups = 1;
    const int need = (sp.npt + 7) / 8;                    // more streams than pixel tiles would idle
    if (groups > need) groups = need;
    static const int groups_env = [] { const char* e = getenv("FQ_STREAM_GROUPS"); return e ? atoi(e) : 0; }();   // tests: few streams, many tiles each
    if (groups_env > 0 && groups > groups_env) groups = groups_env;
    sp.ns = groups * 8;
    const int grid = groups * unit;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NW), lds, st, x, w, qbias, q, p, sp);
    return true;
}

template <int SB, int MT, int NW, int D>
bool launch_outputs(hipStream_t st, const int8_t* x, const int8_t* w, const float* qbias, int8_t* q, const ConvParams& p,
                    StreamParams sp, bool gather) {
    if (p.res) {
        if (gather) return false;
        if (p.res_bytes == 2) return launch_variant<SB, MT, NW, D, true, true, false>(st, x, w, qbias, q, p, sp);
        return launch_variant<SB, MT, NW, D, true, false, false>(st, x, w, qbias, q, p, sp);
    }
    if (gather) {
        if constexpr (SB == 128) return launch_variant<SB, MT, NW, D, false, false, true>(st, x, w, qbias, q, p, sp);
        else return false;
    }
    return launch_variant<SB, MT, NW, D, false, false, false>(st, x, w, qbias, q, p, sp);
}

}  // namespace

bool launch_conv1x1_stream(hipStream_t st, const int8_t* x, const int8_t* w, const float* qbias, float* y, int8_t* q,
                           const ConvParams& p) {
    // Opt-in (FQ_CONV_STREAM=1): bit-exact, but on ResNet-50 at 256 images it is not faster than the general kernel inside
    // the network (round 3: 3.37 vs 3.12 ms per forward; DESIGN.md 5b "the streaming 1x1 kernel").
    static const bool on = [] { const char* e = getenv("FQ_CONV_STREAM"); return e && e[0] == '1'; }();
    if (!on || y || p.R != 1 || p.S != 1 || p.pad_h || p.pad_w || p.stride_h != p.stride_w || p.stride_h > 2) return false;
    if (p.rs == 0 || p.Kpad != p.K || (!q && !p.wide)) return false;           // integer tail, no channel padding
    if (!(p.C == 64 || p.C % 128 == 0) || p.K % 64) return false;
    const bool gather = p.stride_h != 1;
    if (gather && p.M >= (1 << 24)) return false;
    const unsigned long long out_bytes = (unsigned long long)p.M * p.Kpad;
    if (2ull * (out_bytes + 256ull * 8 * p.Kpad) >= 0x80000000ull) return false;   // 32-bit offsets, the int16 arrays included
    if (!gather && (unsigned long long)p.x_bytes + 2048ull * p.C >= 0x80000000ull) return false;
    StreamParams sp = {};
#ifdef FQ_STREAM_ABLATE
    { const char* e = getenv("FQ_STREAM_ABLATE"); sp.ablate = e ? atoi(e) : 0; }
#endif
    sp.out_bytes = (unsigned)out_bytes;
    sp.inv_pq = 1.0f / (float)(p.P * p.Q);
    sp.inv_q = 1.0f / (float)p.Q;
    static const int mt_env = [] { const char* e = getenv("FQ_STREAM_MT"); return e ? atoi(e) : 0; }();
    if (p.C == 64) {
        sp.nsteps = 1;
        if (p.K % 128 == 0 && mt_env != 2) return launch_outputs<64, 4, 4, 2>(st, x, w, qbias, q, p, sp, gather);
        return launch_outputs<64, 2, 4, 2>(st, x, w, qbias, q, p, sp, gather);
    }
    sp.nsteps = p.C / 128;
    // the widest channel tile whose weights, with the activation rings of 8 waves, fit in LDS
    if (p.K % 128 == 0 && mt_env != 2 && stream_lds_bytes<128, 4, 8, 2>(sp.nsteps) <= kLdsPerCU)
        return launch_outputs<128, 4, 8, 2>(st, x, w, qbias, q, p, sp, gather);
    if (stream_lds_bytes<128, 2, 8, 2>(sp.nsteps) <= kLdsPerCU)
        return launch_outputs<128, 2, 8, 2>(st, x, w, qbias, q, p, sp, gather);
    return false;
}

}  // namespace fq
