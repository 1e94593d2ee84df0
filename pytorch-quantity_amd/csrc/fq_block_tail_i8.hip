// fq_block_tail_i8.hip -- the tail of a bottleneck block of the integer-simulation model AND the head of the next one in ONE
// kernel:   conv3 (1x1 expand, C -> K3 = 4 C channels)  ->  NewAdd with the shortcut (+ the nn.ReLU behind it)  ->  the next
// block's conv1 (1x1 reduce, K3 -> C2 channels, + its nn.ReLU).
//
// Reference: new_quantity_op.py:124-133 (NewConv2d.forward, twice) and :166-174 (NewAdd.forward) run this chain as
// Quantity -> conv -> RightShift -> BiasAdd -> Sp -> DeQuantity, add, clamp, ReLU, Quantity -> conv -> ... on fp32 NCHW tensors.
// With resident integer activations (fq_resident.h) the chain was two launches: fq_conv2d_i8_add_resident (reads the int16
// shortcut, writes the exact int16 sum AND its int8 re-quantisation, 5.25 B per element) and fq_conv2d_i8_resident for the next
// conv1, which reads that int8 tensor straight back (1.25 B per element).  At 256 images the 56 x 56 and 28 x 28 stages are
// bound by exactly those bytes.  Here the re-quantised sum never leaves the CU: a workgroup owns 128 pixels and ALL K3 channels
// of them, walks over the channels in slices of 128, and after the add of a slice multiplies the slice's int8 values -- staged
// in LDS in MFMA operand layout -- into the next conv1's accumulators (32 pixels x C2 channels per wave, kept in registers
// across the slices).  Per element of the sum: 2 B read (shortcut) + 2 B written (sum) + C2 / K3 B (the next conv1's output)
// instead of 6.5, and one launch instead of two.
//
// Layout of a workgroup (256 threads = 4 waves, wave w owns pixels [32 w, 32 w + 32) of the tile and everything about them):
//   x fragments        lane = pixel, 16 bytes per 32-channel sub-step, straight from the int8 NHWC tensor into registers, once
//   conv3 weights      one 128-row slice at a time in LDS (XOR-swizzled rows), fetched global -> registers one slice ahead
//   conv3 epilogue     RightShift + bias + Sp in integer arithmetic, int8 through the wave's OWN 32 rows of an LDS tile
//                      (row = pixel, 128 bytes = the slice's channels), read back 16 channels per lane, NewAdd with the
//                      shortcut (fq_resident.h), 16-byte stores of the sum; the re-quantised int8 goes back to the SAME 16
//                      bytes of the tile, which is then the B operand of ...
//   the next conv1     C2 x 128-byte slice of its weights in LDS, 4 MFMA sub-steps per slice and accumulator tile
// LDS instructions of one wave execute in order and a wave only touches its own 32 rows of the tile, so the whole epilogue has
// NO workgroup barrier; the two barriers per slice order the weight slices' hand-over only.
//
// Same integers, bit for bit, as the two launches (tests/test_gpu_block_tail.py runs both against the CPU oracle).
#include <utility>

#include "fq_conv_i8_common.h"

namespace fq {
namespace {

struct TailParams {                      // what conv_tail_i (fq_int_tail.h) reads
    int rs, half_rs, ilo, ihi, slo, shi;
};

struct BtParams {
    int M;                               // pixels (N * H * W)
    int K3;                              // conv3 output channels = channel stride of the shortcut / sum / narrow tensors
    unsigned x_bytes;                    // M * C
    TailParams t3, t1;
    const void* res;                     // shortcut: int8 / int16 [M][K3]
    int res_bytes;
    int16_t* wide;                       // exact sum (may be null)
    int8_t* narrow;                      // its re-quantisation (may be null: nobody but the fused conv1 reads it)
    AddResParams ap;
    // the projection shortcut computed in the kernel (CP > 0): res = Sp(RightShift(conv1x1(xp, wp)) + bias), int8 on grid g_res
    const int8_t* xp;                    // [Mp][CP] int8 NHWC: the block's input
    const int8_t* wp;                    // [K3][CP]
    const float* qbiasp;                 // [K3], integer valued
    TailParams tp;
    unsigned xp_bytes;
    int sp, Ho, Wo, Hp, Wp;              // stride of the projection; output plane; the plane of xp
};

// LDS rows of RB bytes hold RB / 16 chunks of 16 bytes; position c of row r holds chunk c ^ swz_of(r), chosen so that the 16 lanes
// of every ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) reading chunk c of 16 different rows land in
// 16 different 16-byte slots of the 256-byte bank row
template <int RB> __device__ __forceinline__ int swz_of(int row) {
    return RB == 256 ? row & 15 : (RB == 128 ? (row >> 1) & 7 : (row >> 2) & 3);
}

// The four-instruction tail (fq_int_tail.h) of four consecutive channels: their constants are three 16-byte LDS reads (G = 4), or
// two rounds of three 8-byte reads (G = 2) where twelve constant registers at a time are more than the kernel has left
typedef int v2i_t __attribute__((ext_vector_type(2)));
template <int G, int N>
__device__ __forceinline__ void tail4(int (&v)[4], int a0, int a1, int a2, int a3, const int (&t)[3][N], int ch, int rs) {
    const int acc[4] = {a0, a1, a2, a3};
    if constexpr (G == 4) {
        const v4i cB = *reinterpret_cast<const v4i*>(&t[0][ch]), cL = *reinterpret_cast<const v4i*>(&t[1][ch]),
                  cH = *reinterpret_cast<const v4i*>(&t[2][ch]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = conv_tail_k(acc[e], cB[e], cL[e], cH[e], rs);
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const v2i_t cB = *reinterpret_cast<const v2i_t*>(&t[0][ch + 2 * h]), cL = *reinterpret_cast<const v2i_t*>(&t[1][ch + 2 * h]),
                        cH = *reinterpret_cast<const v2i_t*>(&t[2][ch + 2 * h]);
            v[2 * h] = conv_tail_k(acc[2 * h], cB[0], cL[0], cH[0], rs);
            v[2 * h + 1] = conv_tail_k(acc[2 * h + 1], cB[1], cL[1], cH[1], rs);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// C: conv3's input channels (64 / 128).  C2: the fused next conv1's output channels (64 / 128; 0: no next conv -- the kernel is
// then conv3 + NewAdd alone, in the barrier-free form).  kRes16: the shortcut is int16 (a previous sum) or int8 (a projection).
#ifndef FQ_BT_WAVES
#define FQ_BT_WAVES 2
#endif
// CP: input channels of a projection shortcut that is computed HERE instead of being read (0: the shortcut is a tensor).  The first
// block of a stage adds conv3's output to a 1x1 convolution of the block's input; as its own launch that convolution writes K3
// bytes per pixel which this kernel reads straight back (at 256 images and 56 x 56: 205 MB each way, 75 us for the launch).  Here the
// workgroup multiplies its 128 pixels of the block input with the slice's 128 rows of wp right before conv3's slice, runs the
// projection's own integer tail, and passes the bytes through the wave's rows of the LDS tile into the registers the shortcut would
// have been loaded into: same integers, CP instead of K3 bytes per pixel read, nothing written.
template <int C, int C2, bool kRes16, int CP = 0>
__global__ __launch_bounds__(kConvBlock) __attribute__((amdgpu_waves_per_eu(FQ_BT_WAVES))) void block_tail_i8_kernel(
    const int8_t* __restrict__ x, const int8_t* __restrict__ w3, const float* __restrict__ qbias3,
    const int8_t* __restrict__ w1, const float* __restrict__ qbias1, int8_t* __restrict__ q1, const BtParams p) {
    constexpr int KS3 = C / 32;                          // MFMA sub-steps of conv3 (its whole reduction)
    constexpr int W3_LOADS = (128 * (C / 16)) / kConvBlock;      // 16-byte chunks of a conv3 weight slice per thread
    constexpr int MT1 = C2 / 32;                         // accumulator tiles of the next conv1 per wave
    constexpr int W1_LOADS = C2 ? (C2 * 8) / kConvBlock : 1;
    constexpr bool kNext = C2 != 0;
    constexpr bool kProj = CP != 0;
    constexpr int TG = (C == 128 && C2 == 128 && kRes16) ? 2 : 4;      // (the one variant without twelve registers to spare)
    static_assert(!(kProj && kRes16), "a projection shortcut is int8");
    constexpr int KSP = kProj ? CP / 32 : 1;             // MFMA sub-steps of the projection
    constexpr int WP_LOADS = kProj ? (128 * (CP / 16)) / kConvBlock : 1;
    __shared__ __attribute__((aligned(16))) int8_t sWP[kProj ? 128 * CP : 16];
    // per-channel constants of the integer tails (fq_int_tail.h, tail_consts): [0] rounding constant with the bias in it, [1] / [2]
    // the merged clamp's bounds -- four consecutive channels of one kind are one 16-byte read
    __shared__ __attribute__((aligned(16))) int sTP[kProj ? 3 : 1][kProj ? 1024 : 4];
    __shared__ __attribute__((aligned(16))) int8_t sW3[128 * C];
    __shared__ __attribute__((aligned(16))) int8_t sW1[kNext ? C2 * 128 : 16];
    __shared__ __attribute__((aligned(16))) int8_t sN[kTP * 128];
    __shared__ __attribute__((aligned(16))) int sT3[3][1024];
    __shared__ __attribute__((aligned(16))) int sT1[3][kNext ? C2 : 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, prow = wave * 32 + (lane & 31);
    const int m0 = blockIdx.x * kTP;
    const int KT = p.K3 >> 7;

    for (int i = tid; i < p.K3; i += kConvBlock) {
        const TailK k = tail_consts((int)qbias3[i], p.t3);                           // (biases are integer valued by contract)
        sT3[0][i] = k.B; sT3[1][i] = k.lo; sT3[2][i] = k.hi;
    }
    if constexpr (kProj) {
        for (int i = tid; i < p.K3; i += kConvBlock) {
            const TailK k = tail_consts((int)p.qbiasp[i], p.tp);
            sTP[0][i] = k.B; sTP[1][i] = k.lo; sTP[2][i] = k.hi;
        }
    }
    if (kNext && tid < C2) {
        const TailK k = tail_consts((int)qbias1[tid], p.t1);
        sT1[0][tid] = k.B; sT1[1][tid] = k.lo; sT1[2][tid] = k.hi;
    }

    // ---- x: this lane's pixel, 16 bytes per sub-step, for the whole tile's life
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(x), 0, p.x_bytes, 0x00020000);
    v4i fb[KS3];
    {
        const int m = m0 + prow;
        const unsigned off = m < p.M ? (unsigned)m * (unsigned)C + (unsigned)(half * 16) : kOutOfRange;
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) fb[ks] = load_act(xr, off + (unsigned)(ks * 32));
    }

    // ---- the projection's operand: the same lane's pixel of the block input (at the projection's stride)
    v4i fp[KSP];
    if constexpr (kProj) {
        const __amdgpu_buffer_rsrc_t xpr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.xp), 0, p.xp_bytes, 0x00020000);
        const int m = m0 + prow;
        unsigned off = kOutOfRange;
        if (m < p.M) {
            unsigned mp = (unsigned)m;
            if (p.sp != 1 || p.Hp != p.Ho || p.Wp != p.Wo) {
                const unsigned hw = (unsigned)(p.Ho * p.Wo), n = (unsigned)m / hw, r = (unsigned)m - n * hw;
                const unsigned oh = r / (unsigned)p.Wo, ow = r - oh * (unsigned)p.Wo;
                mp = (n * (unsigned)p.Hp + oh * (unsigned)p.sp) * (unsigned)p.Wp + ow * (unsigned)p.sp;
            }
            off = mp * (unsigned)CP + (unsigned)(half * 16);
        }
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) fp[ks] = load_act(xpr, off + (unsigned)(ks * 32));
    }

    // ---- weight slices: global -> registers (one slice ahead) -> LDS
    // (everything below addresses memory through buffer descriptors: a 32-bit per-lane offset that is fixed for the tile's life plus
    //  a scalar slice offset -- 64-bit per-lane pointers for three output streams, the shortcut and two weight matrices were 40
    //  registers of loop invariants, and with the next conv1's 64 accumulators alive that meant spills)
    const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(w3), 0, (unsigned)(p.K3 * C), 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(w1), 0, (unsigned)(C2 * p.K3), 0x00020000);
    const __amdgpu_buffer_rsrc_t wpr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.wp), 0, kProj ? (unsigned)(p.K3 * CP) : 0u, 0x00020000);
    v4i r3[W3_LOADS], r1[W1_LOADS], rp[WP_LOADS];
    auto fetch = [&](int kt) {
        if constexpr (kProj) {
#pragma unroll
            for (int j = 0; j < WP_LOADS; ++j) {
                const int i = tid + kConvBlock * j, row = i / (CP / 16), c = i % (CP / 16);
                rp[j] = load_act(wpr, (unsigned)(row * CP + c * 16) + (unsigned)(kt * 128 * CP));
            }
        }
#pragma unroll
        for (int j = 0; j < W3_LOADS; ++j) {
            const int i = tid + kConvBlock * j, row = i / (C / 16), c = i % (C / 16);
            r3[j] = load_act(w3r, (unsigned)(row * C + c * 16) + (unsigned)(kt * 128 * C));
        }
        if constexpr (kNext) {
#pragma unroll
            for (int j = 0; j < W1_LOADS; ++j) {
                const int i = tid + kConvBlock * j, row = i >> 3, c = i & 7;
                r1[j] = load_act(w1r, (unsigned)(row * p.K3 + c * 16) + (unsigned)(kt * 128));
            }
        }
    };
    auto stage = [&]() {
        if constexpr (kProj) {
#pragma unroll
            for (int j = 0; j < WP_LOADS; ++j) {
                const int i = tid + kConvBlock * j, row = i / (CP / 16), c = i % (CP / 16);
                *reinterpret_cast<v4i*>(&sWP[row * CP + ((c ^ swz_of<CP ? CP : 64>(row)) * 16)]) = rp[j];
            }
        }
#pragma unroll
        for (int j = 0; j < W3_LOADS; ++j) {
            const int i = tid + kConvBlock * j, row = i / (C / 16), c = i % (C / 16);
            *reinterpret_cast<v4i*>(&sW3[row * C + ((c ^ swz_of<C>(row)) * 16)]) = r3[j];
        }
        if constexpr (kNext) {
#pragma unroll
            for (int j = 0; j < W1_LOADS; ++j) {
                const int i = tid + kConvBlock * j, row = i >> 3, c = i & 7;
                *reinterpret_cast<v4i*>(&sW1[row * 128 + ((c ^ swz_of<128>(row)) * 16)]) = r1[j];
            }
        }
    };
    fetch(0);
    stage();
    __syncthreads();

    // operand-fragment offsets: row (lane & 31) of a 32-row block, chunk 2 ks + half, swizzled by the row
    int a3_off[KS3], a1_off[4], ap_off[KSP];
#pragma unroll
    for (int ks = 0; ks < KS3; ++ks) a3_off[ks] = (lane & 31) * C + (((2 * ks + half) ^ swz_of<C>(lane & 31)) * 16);
#pragma unroll
    for (int ks = 0; ks < KSP; ++ks) ap_off[ks] = kProj ? (lane & 31) * CP + (((2 * ks + half) ^ swz_of<CP ? CP : 64>(lane & 31)) * 16) : 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a1_off[ks] = (lane & 31) * 128 + (((2 * ks + half) ^ swz_of<128>(lane & 31)) * 16);
    int8_t* const my_row = sN + prow * 128;
    const int my_swz = swz_of<128>(prow);

    // store layout of the epilogue: instruction j of a wave covers its pixels 8 j .. 8 j + 7, lane -> (pixel, 16-channel group)
    const int s_pix = wave * 32 + (lane >> 3), s_ch = lane & 7;            // (+ 8 j)

    v16i acc1[kNext ? MT1 : 1];
    if constexpr (kNext) {
#pragma unroll
        for (int a = 0; a < MT1; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[a][r] = 0;
    }

    // the shortcut of a slice: 16 channels x 4 pixels per lane, requested at the head of the slice -- in front of ~700 vector
    // instructions of matrix work and tail.  (Measured and rejected, round 4, in the network at 256 images: a second register set
    // holding the NEXT slice's shortcut one slice ahead -- 174 / 188 / 135 us became 177 / 196 / 151 on the three 56 x 56 tails; and
    // a persistent form with every weight stationary in LDS, no barrier at all and the next tile's operands requested a slice
    // ahead: 171 / 193 / 148.  These launches run within 20-35 % of what HBM delivers; what is left is not request latency.)
    struct Res { v4i_r lo[4], hi[kRes16 ? 4 : 1]; };
    Res res;
    // element offset of (pixel of store item j, first channel of this lane's 16-channel group) in an [M][K3] tensor; beyond the last
    // pixel: out of range for every descriptor below (loads give zeros, stores are dropped)
    const unsigned total = (unsigned)p.M * (unsigned)p.K3;
    unsigned o_el[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + s_pix + 8 * j;
        // (0x40000000: beyond M x K3 < 2^30 elements -- host check -- and still beyond the int16 tensors' byte count when doubled)
        o_el[j] = m < p.M ? (unsigned)m * (unsigned)p.K3 + (unsigned)(16 * s_ch) : 0x40000000u;
    }
    const __amdgpu_buffer_rsrc_t resr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, total * (kRes16 ? 2u : 1u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wider = __builtin_amdgcn_make_buffer_rsrc(p.wide, 0, p.wide ? 2u * total : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t narr = __builtin_amdgcn_make_buffer_rsrc(p.narrow, 0, p.narrow ? total : 0u, 0x00020000);
#pragma unroll 1
    for (int kt = 0; kt < KT; ++kt) {
        const int k0 = kt * 128;
        if constexpr (kProj) {
            // the shortcut of this slice = the projection of this wave's 32 pixels: 128 channels, 64 at a time, tail, int8 through
            // the wave's own rows of the tile and back as 16 channels x 4 pixels per lane (LDS instructions of one wave run in order)
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                v16i acc[2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][r] = 0;
#pragma unroll
                for (int ks = 0; ks < KSP; ++ks) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const v4i fa = *reinterpret_cast<const v4i*>(&sWP[(2 * hs + a) * 32 * CP + ap_off[ks]]);
                        acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fp[ks], acc[a], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        int v[4];
                        const int byte = (2 * hs + a) * 32 + 8 * g + 4 * half;          // first of this lane's four channels
                        tail4<TG>(v, acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3], sTP, k0 + byte, p.tp.rs);
                        *reinterpret_cast<unsigned*>(my_row + (((byte >> 4) ^ my_swz) * 16) + (byte & 15)) = pack4(v[0], v[1], v[2], v[3]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pix = s_pix + 8 * j;
                res.lo[j] = *reinterpret_cast<const v4i_r*>(sN + pix * 128 + ((s_ch ^ swz_of<128>(pix)) * 16));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (kRes16) {
                    res.lo[j] = (v4i_r)load_act(resr, 2u * o_el[j] + (unsigned)(2 * k0));
                    res.hi[j] = (v4i_r)load_act(resr, 2u * o_el[j] + (unsigned)(2 * k0 + 16));
                } else {
                    res.lo[j] = (v4i_r)load_act(resr, o_el[j] + (unsigned)k0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- conv3, this slice: 32 pixels x 128 channels per wave, 64 channels (two accumulator tiles) at a time -- RightShift +
        // BiasAdd + Sp of the first half runs while the second half's accumulators do not exist yet (32 registers fewer)
        constexpr int PT = 2;
#pragma unroll
        for (int hs = 0; hs < 4 / PT; ++hs) {
            v16i acc[PT];
#pragma unroll
            for (int a = 0; a < PT; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][r] = 0;
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) {
#pragma unroll
                for (int a = 0; a < PT; ++a) {
                    const v4i fa = *reinterpret_cast<const v4i*>(&sW3[(PT * hs + a) * 32 * C + a3_off[ks]]);
                    acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb[ks], acc[a], 0, 0, 0);
                }
            }
            // 4 channels = one dword into this lane's own row of the tile
#pragma unroll
            for (int a = 0; a < PT; ++a) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int v[4];
                    const int byte = (PT * hs + a) * 32 + 8 * g + 4 * half;          // channel of v[0] inside the slice
                    tail4<TG>(v, acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3], sT3, k0 + byte, p.t3.rs);
                    *reinterpret_cast<unsigned*>(my_row + (((byte >> 4) ^ my_swz) * 16) + (byte & 15)) = pack4(v[0], v[1], v[2], v[3]);
                    __builtin_amdgcn_sched_barrier(0);        // four values at a time: left alone the scheduler runs all 32 tails abreast
                }
            }
        }
        // ---- NewAdd (+ ReLU + the consumers' Quantity) on 16 channels of one pixel per lane; rows of this wave only
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = s_pix + 8 * j;
            int8_t* const cell = sN + pix * 128 + ((s_ch ^ swz_of<128>(pix)) * 16);
            Vec16<int8_t> cv;
            cv.a = *reinterpret_cast<const v4i_r*>(cell);
            Add16Out o;
            if constexpr (kRes16) {
                Vec16<int16_t> rv;
                rv.a = res.lo[j]; rv.b = res.hi[j];
                o = add_resident_16_regs(cv, rv, p.wide != nullptr, true, p.ap);
            } else {
                Vec16<int8_t> rv;
                rv.a = res.lo[j];
                o = add_resident_16_regs(cv, rv, p.wide != nullptr, true, p.ap);
            }
            // (an output nobody wants has a descriptor of zero records: its stores are dropped in the address unit.  The slice offset
            //  rides in the VECTOR offset: a 16-byte buffer store with a scalar-offset register is the store hipcc does not pad against
            //  a write of its data registers on gfx950)
            const unsigned ov = o_el[j] + (unsigned)k0;
            __builtin_amdgcn_raw_buffer_store_b128((v4u)o.w0, wider, (int)(2u * ov), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128((v4u)o.w1, wider, (int)(2u * ov + 16u), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128((v4u)o.n, narr, (int)ov, 0, 0);
            if constexpr (kNext) *reinterpret_cast<v4i_r*>(cell) = o.n;
            __builtin_amdgcn_sched_barrier(0);                // one pixel group at a time: interleaving the four inflates the live set
        }
        // the next slice's weights are requested only now -- the shortcut registers are free again -- and land under the matrix
        // work below; they come from L2 (a slice is 8-32 KB that every workgroup reads)
        if (kt + 1 < KT) fetch(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- the next conv1, this slice of its reduction: B = the re-quantised sums of this wave's pixels
        if constexpr (kNext) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const v4i fn = *reinterpret_cast<const v4i*>(sN + wave * 32 * 128 + a1_off[ks]);
#pragma unroll
                for (int a = 0; a < MT1; ++a) {
                    const v4i fa = *reinterpret_cast<const v4i*>(&sW1[a * 32 * 128 + a1_off[ks]]);
                    acc1[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fn, acc1[a], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < KT) {
            __syncthreads();                                  // everybody is done reading this slice's weights
            stage();
            __syncthreads();
        }
    }

    // ---- the next conv1's tail (+ its ReLU): int8 through the wave's own rows, 16-byte stores of C2 contiguous bytes per pixel
    if constexpr (kNext) {
#pragma unroll
        for (int a = 0; a < MT1; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int v[4];
                const int byte = a * 32 + 8 * g + 4 * half;
                tail4<4>(v, acc1[a][4 * g], acc1[a][4 * g + 1], acc1[a][4 * g + 2], acc1[a][4 * g + 3], sT1, byte, p.t1.rs);
                *reinterpret_cast<unsigned*>(my_row + (((byte >> 4) ^ my_swz) * 16) + (byte & 15)) = pack4(v[0], v[1], v[2], v[3]);
            }
        }
        constexpr int CPP = C2 / 16, PPI = 64 / CPP;          // 16-byte groups per pixel, pixels per store instruction
        const __amdgpu_buffer_rsrc_t q1r = __builtin_amdgcn_make_buffer_rsrc(q1, 0, (unsigned)p.M * (unsigned)C2, 0x00020000);
#pragma unroll
        for (int j = 0; j < 32 / PPI; ++j) {
            const int pix = wave * 32 + PPI * j + lane / CPP, ch = lane % CPP;
            const v4i_r o = *reinterpret_cast<const v4i_r*>(sN + pix * 128 + ((ch ^ swz_of<128>(pix)) * 16));
            const int m = m0 + pix;
            __builtin_amdgcn_raw_buffer_store_b128((v4u)o, q1r, m < p.M ? (int)((unsigned)m * (unsigned)C2 + (unsigned)(16 * ch)) : (int)kOutOfRange, 0, 0);
        }
    }
}

TailParams tail_params(int rs, int relu) {
    TailParams t;
    t.rs = rs; t.half_rs = 1 << (rs - 1);
    t.ilo = -128; t.ihi = 127;
    t.slo = relu ? 0 : -128; t.shi = 127;
    return t;
}

template <int C, int C2>
void launch_bt(dim3 grid, hipStream_t st, const int8_t* x, const int8_t* w3, const float* qb3, const int8_t* w1, const float* qb1,
               int8_t* q1, const BtParams& p) {
    if (p.res_bytes == 2)
        hipLaunchKernelGGL((block_tail_i8_kernel<C, C2, true>), grid, dim3(kConvBlock), 0, st, x, w3, qb3, w1, qb1, q1, p);
    else
        hipLaunchKernelGGL((block_tail_i8_kernel<C, C2, false>), grid, dim3(kConvBlock), 0, st, x, w3, qb3, w1, qb1, q1, p);
}

template <int C, int C2, int CP>
void launch_bt_proj(dim3 grid, hipStream_t st, const int8_t* x, const int8_t* w3, const float* qb3, const int8_t* w1, const float* qb1,
                    int8_t* q1, const BtParams& p) {
    hipLaunchKernelGGL((block_tail_i8_kernel<C, C2, false, CP>), grid, dim3(kConvBlock), 0, st, x, w3, qb3, w1, qb1, q1, p);
}

}  // namespace
}  // namespace fq

using namespace fq;

extern "C" int fq_block_tail_i8_supported(int C, int K3, int C2, int rs3, int rs1, int ob3, int g_res, int res_bytes, int ib) {
    if (!(C == 64 || C == 128 || C == 256) || K3 < 128 || K3 > 1024 || (K3 & 127)) return 0;
    if (!(C2 == 0 || C2 == 64 || C2 == 128) || (C == 256 && C2 != 0)) return 0;
    if (rs3 < 1 || rs3 > 16 || (C2 && (rs1 < 1 || rs1 > 16))) return 0;
    (void)ob3; (void)g_res; (void)ib;                     // every grid fq_conv2d_i8_add_resident takes (packed, 32-bit and fp32 forms of NewAdd)
    return (res_bytes == 1 || res_bytes == 2) ? 1 : 0;
}

extern "C" int fq_block_tail_i8(const int8_t* x_nhwc, const int8_t* w3_krsc, const float* qbias3, int rs3, int ob3, const void* res,
                                int res_bytes, int g_res, int16_t* wide, int g_wide, int8_t* narrow, int ib, int relu,
                                const int8_t* w1_krsc, const float* qbias1, int rs1, int relu1, int8_t* q1_nhwc, long M, int C,
                                int K3, int C2, fq_stream_t stream) {
    if (!res || (res_bytes != 1 && res_bytes != 2)) return FQ_ERR_INVALID_ARG;
    if (M < 0 || !fq_block_tail_i8_supported(C, K3, C2, rs3, rs1, ob3, g_res, res_bytes, ib)) return FQ_ERR_UNSUPPORTED;
    if (M == 0) return FQ_OK;
    if (!x_nhwc || !w3_krsc || !qbias3 || (C2 && (!w1_krsc || !qbias1 || !q1_nhwc)) || (!C2 && !wide && !narrow)) return FQ_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x_nhwc) | reinterpret_cast<uintptr_t>(w3_krsc) | reinterpret_cast<uintptr_t>(res) |
         reinterpret_cast<uintptr_t>(wide) | reinterpret_cast<uintptr_t>(narrow) | reinterpret_cast<uintptr_t>(w1_krsc) |
         reinterpret_cast<uintptr_t>(q1_nhwc)) & 15u)
        return FQ_ERR_INVALID_ARG;
    if (M * (long)K3 >= 0x3fffffffL || M * (long)C >= 0x7fffffffL) return FQ_ERR_UNSUPPORTED;
    BtParams p;
    p.M = (int)M; p.K3 = K3; p.x_bytes = (unsigned)(M * C);
    p.t3 = tail_params(rs3, 0);
    p.t1 = tail_params(C2 ? rs1 : 1, relu1);
    p.res = res; p.res_bytes = res_bytes; p.wide = wide; p.narrow = narrow;
    p.xp = nullptr; p.wp = nullptr; p.qbiasp = nullptr; p.tp = tail_params(1, 0); p.xp_bytes = 0; p.sp = 1; p.Ho = p.Wo = p.Hp = p.Wp = 1;
    const int rc = make_add_params(ob3, g_res, g_wide, wide != nullptr, ib, relu, &p.ap);
    if (rc != FQ_OK) return rc;
    // the integer tail of conv3 is only the reference's fp32 chain while |acc| + 2^15 < 2^31: C * 127 * 128 is far below
    const dim3 grid((unsigned)((M + kTP - 1) / kTP));
    hipStream_t st = as_stream(stream);
    if (C == 64) {
        if (C2 == 64) launch_bt<64, 64>(grid, st, x_nhwc, w3_krsc, qbias3, w1_krsc, qbias1, q1_nhwc, p);
        else if (C2 == 128) launch_bt<64, 128>(grid, st, x_nhwc, w3_krsc, qbias3, w1_krsc, qbias1, q1_nhwc, p);
        else launch_bt<64, 0>(grid, st, x_nhwc, w3_krsc, qbias3, w1_krsc, qbias1, q1_nhwc, p);
    } else if (C == 128) {
        if (C2 == 64) launch_bt<128, 64>(grid, st, x_nhwc, w3_krsc, qbias3, w1_krsc, qbias1, q1_nhwc, p);
        else if (C2 == 128) launch_bt<128, 128>(grid, st, x_nhwc, w3_krsc, qbias3, w1_krsc, qbias1, q1_nhwc, p);
        else launch_bt<128, 0>(grid, st, x_nhwc, w3_krsc, qbias3, w1_krsc, qbias1, q1_nhwc, p);
    } else {
        launch_bt<256, 0>(grid, st, x_nhwc, w3_krsc, qbias3, w1_krsc, qbias1, q1_nhwc, p);
    }
    note_conv_variant(11, 128);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

// The first block of a stage: conv3 + NewAdd + (the next block's conv1) with the PROJECTION shortcut -- a 1x1 convolution of the
// block's input, stride 1 or 2, its own RightShift + BiasAdd + Sp, no ReLU -- computed in the same kernel instead of being read.
extern "C" int fq_block_tail_proj_i8_supported(int C, int K3, int C2, int CP, int rs3, int rs1, int rsp, int stride_p) {
    if (!(C == 64 && CP == 64 && (C2 == 64 || C2 == 0))) return 0;
    if (K3 < 128 || K3 > 1024 || (K3 & 127)) return 0;
    if (rs3 < 1 || rs3 > 16 || rsp < 1 || rsp > 16 || (C2 && (rs1 < 1 || rs1 > 16))) return 0;
    return (stride_p == 1 || stride_p == 2) ? 1 : 0;
}

extern "C" int fq_block_tail_proj_i8(const int8_t* x_nhwc, const int8_t* w3_krsc, const float* qbias3, int rs3, int ob3,
                                     const int8_t* xp_nhwc, const int8_t* wp_krsc, const float* qbiasp, int rsp, int obp, int stride_p,
                                     int Hp, int Wp, int16_t* wide, int g_wide, int8_t* narrow, int ib, int relu,
                                     const int8_t* w1_krsc, const float* qbias1, int rs1, int relu1, int8_t* q1_nhwc, int N, int H,
                                     int W, int C, int K3, int C2, int CP, fq_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || Hp <= 0 || Wp <= 0) return FQ_ERR_INVALID_ARG;
    if (!fq_block_tail_proj_i8_supported(C, K3, C2, CP, rs3, rs1, rsp, stride_p)) return FQ_ERR_UNSUPPORTED;
    if ((Hp - 1) / stride_p + 1 != H || (Wp - 1) / stride_p + 1 != W) return FQ_ERR_INVALID_ARG;   // a 1x1 convolution without padding
    const long M = (long)N * H * W, Mp = (long)N * Hp * Wp;
    if (M == 0) return FQ_OK;
    if (!x_nhwc || !w3_krsc || !qbias3 || !xp_nhwc || !wp_krsc || !qbiasp || (C2 && (!w1_krsc || !qbias1 || !q1_nhwc)) ||
        (!C2 && !wide && !narrow))
        return FQ_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x_nhwc) | reinterpret_cast<uintptr_t>(w3_krsc) | reinterpret_cast<uintptr_t>(xp_nhwc) |
         reinterpret_cast<uintptr_t>(wp_krsc) | reinterpret_cast<uintptr_t>(wide) | reinterpret_cast<uintptr_t>(narrow) |
         reinterpret_cast<uintptr_t>(w1_krsc) | reinterpret_cast<uintptr_t>(q1_nhwc)) & 15u)
        return FQ_ERR_INVALID_ARG;
    if (M * (long)K3 >= 0x3fffffffL || M * (long)C >= 0x7fffffffL || Mp * (long)CP >= 0x7fffffffL) return FQ_ERR_UNSUPPORTED;
    BtParams p;
    p.M = (int)M; p.K3 = K3; p.x_bytes = (unsigned)(M * C);
    p.t3 = tail_params(rs3, 0);
    p.t1 = tail_params(C2 ? rs1 : 1, relu1);
    p.res = nullptr; p.res_bytes = 1; p.wide = wide; p.narrow = narrow;
    p.xp = xp_nhwc; p.wp = wp_krsc; p.qbiasp = qbiasp; p.tp = tail_params(rsp, 0); p.xp_bytes = (unsigned)(Mp * CP);
    p.sp = stride_p; p.Ho = H; p.Wo = W; p.Hp = Hp; p.Wp = Wp;
    const int rc = make_add_params(ob3, obp, g_wide, wide != nullptr, ib, relu, &p.ap);
    if (rc != FQ_OK) return rc;
    const dim3 grid((unsigned)((M + kTP - 1) / kTP));
    hipStream_t st = as_stream(stream);
    if (C2 == 64) launch_bt_proj<64, 64, 64>(grid, st, x_nhwc, w3_krsc, qbias3, w1_krsc, qbias1, q1_nhwc, p);
    else launch_bt_proj<64, 0, 64>(grid, st, x_nhwc, w3_krsc, qbias3, w1_krsc, qbias1, q1_nhwc, p);
    note_conv_variant(12, 128);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}
