// fq_conv_wino_f32.hip -- the stride-1 3x3 convolutions of the float calibration forward as Winograd F(2x2, 3x3) on the
// fp32 matrix cores, with the calibration's statistic taken in the epilogue (the contract of fq_conv_kxk_f32).
//
// Why.  The 3x3 layers are 7.9 of the 18.8 ms of a pass-1 forward of ResNet-50 at 256 images, and the direct kernel
// (fq_conv1x1_f32.hip, R x S form) already runs them at 117-127 of the 157 TFLOP/s the fp32 MFMA has: what is left is
// the number of multiplications.  F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input tile with 16 multiplications
// per (input channel, output channel) instead of 36:
//     Y = At [ sum_c (G g_c Gt) .* (Bt d_c B) ] A
// -- 16 independent GEMMs  M_e[k][t] = sum_c U_e[c][k] V_e[c][t]  (e = the 16 positions of the transformed tile, k = output
// channel, t = tile), between an input transform (32 additions per tile and channel) and an output transform (24 per
// tile and output channel).
//
// Mapping.  A work item = 64 output channels x 64 (or 32) tiles (tiles are numbered through the whole batch:
// t = (n * TH + ty) * TW + tx, so a 7x7 plane costs its 4x4 tiles and nothing else).  A (32 channels x 32 tiles) block of
// it with all 16 positions would be 16 accumulator tiles of v_mfma_f32_32x32x2_f32 = 256 registers, i.e. one wave per SIMD
// with nobody to cover its stalls (built first: 84 % of the matrix cycles at best, scripts/mfma_f32_probe.hip).  So TWO
// waves share a block, eight positions (two rows of the transformed 4x4 tile) each: 128 accumulators + 105 other registers
// -> two waves per SIMD; the two meet once per work item, in the output transform (wino_epilogue).  Two shapes (Geo):
// 8 waves on 64 x 64, one workgroup per CU, for layers with few K steps; 4 waves on 64 x 32, two workgroups per CU,
// for the rest (half-length items: a cheaper last round over the CUs, and two workgroups that cover each other's ends).
// A persistent grid filling the CUs once, work items k-block major: the workgroups an XCD runs at one time share one
// 64-channel slice of U (<= 2 MB: its L2).  The K loop runs over input channels in steps of 8:
//   * U (the transformed weights) never touches LDS: the host packs it (fq_conv3x3_wino_f32_pack) as
//     [c / 8][e][c % 2][k][c % 8 / 2], so that the A operands of the four MFMAs of (step, e) are ONE 16-byte load per lane
//     straight from L2 into the registers the MFMAs read; the load of step s + 1 is issued as soon as (step s, e) has
//     issued its MFMAs -- a whole step of latency cover, no double buffer.
//   * V is made in the kernel: thread = (tile, one input channel of the step); four 16-byte loads of the rows of the
//     4x4 input tile through a buffer descriptor (per-thread byte offsets computed once per work item; "row outside the
//     image" = an out-of-range offset that the address unit answers with zero; the two edge columns are dropped by select;
//     the channel is the scalar offset), 32 additions, 16 LDS writes into the stage of step s + 1; the loads of step s + 2
//     follow at once.  Three 32 (16) KB stages and one barrier per step, in the middle of it (the scheme of conv1x1_tiles):
//     nothing is waited for at a step boundary.  B operands: one ds_read_b128 per (step, e).
//   * Epilogue: each wave applies the column stage of the output transform to its two rows, the two waves of a block
//     exchange one row of the result through LDS and each makes one row of every 2x2 output tile: + bias, statistic, ReLU copy, one 8-byte store per tile row (4-byte stores with per-pixel "exists" for odd H or
//     W: 7x7).
// What was measured on the way (profiles/r04_conv_wino_*): every vector instruction between two MFMAs costs the matrix
// pipe ~7 cycles whatever it does, so the input rows come as whole 16-byte loads (16 dword loads per tile and channel: 23 %
// slower), the B operands as one read per (step, e), and the accumulators are copied out of the AGPRs where they are used.
//
// Numerics.  fp32 throughout; U is computed from the weights in fp64 and rounded once.  Per output: an fmaf chain over
// c = 0 .. Cin-1 per position, then the fixed additions of the two transforms (output: along a row first, then down),
// then the bias.  Deterministic and
// batch-size independent (no K split, no workspace: an image computes the same bits wherever it sits in whatever batch).
// It is NOT the direct sum: measured |error| <= 1.8e-7 of sum |w||x| on Gaussian data against 3.5e-7 for the direct kernel
// (fewer terms per chain) -- the reference's own GPU forward goes through the convolution library's Winograd kernels for
// these layers, so no table ever depended on the direct order.
#include "fq_common.h"
#include "fq_producer_stat.h"

#ifndef FQ_WINO_ABLATE
#define FQ_WINO_ABLATE 0      // debug builds only (make wino_ablate): 1 no x loads in the loop, 2 no U loads, 4 no transform, 8 no epilogue, 16 no MFMAs, 32 no barrier in the loop, 64 no B operand reads in the loop -- wrong results, timing only
#endif
#define FQ_WINO_OFF(bit) ((FQ_WINO_ABLATE) & (bit))

namespace fq {
namespace {

constexpr unsigned kBK = 64;                                  // output channels per workgroup
constexpr unsigned kCS = 8;                                   // input channels per step
// Two shapes of a work item: WT = 2: 64 channels x 64 tiles, 8 waves, one workgroup per CU; WT = 1: 64 channels x 32 tiles, 4 waves,
// two workgroups per CU.  Either way two waves per SIMD.  The small one halves the length of a work item: the last, partly filled
// round over the CUs costs half as much (784 equal items on 256 CUs are 4 rounds for 3.06 rounds of work; 1 568 half items on 512
// slots are 3.06 + the tail at double speed) and two independent workgroups overlap each other's prologue and epilogue -- 15 %
// on the 14x14 layers; a layer with few K steps (Cin = 64: 8 steps per item) pays more per item than that returns (6 % slower).
template <int WT>
struct Geo {
    static constexpr int kT = 256 * WT;
    static constexpr unsigned kWaves = 4u * WT;
    static constexpr unsigned kBT = 32u * WT;                 // tiles per workgroup
    // One stage of V: 16 positions of [c % 2][tile][c % 8 / 2] floats: the B operands of the four MFMAs of a (step, position) are
    // ONE 16-byte LDS read per lane (lanes 0..31: c % 2 = 0, consecutive tiles; lanes 32..63: c % 2 = 1).  The transform's 4-byte
    // writes go 16 bytes apart across the lanes of a wave (four lanes per bank): the LDS pipe has the time, the instruction stream
    // does not -- every vector instruction, whatever it does, takes ~7 cycles from the matrix pipe.
    static constexpr unsigned kKkBytes = kBT * 16u;
    static constexpr unsigned kPlane = 2u * kKkBytes;
    static constexpr unsigned kStageBytes = 16u * kPlane;     // 32 KB / 16 KB
    static constexpr unsigned kLdsBytes = 3u * kStageBytes;
};
constexpr unsigned kOob = 0x80000000u;                        // a byte offset no tensor of <= 2^31 bytes contains
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2v __attribute__((__vector_size__(2 * sizeof(unsigned))));

struct WArgs {
    const float* x;
    const float* u;           // packed by wino_pack_kernel
    const float* bias;        // [Cout] or null
    float* y;                 // or null (only the ReLU copy is wanted)
    float* relu;              // or null
    unsigned Cin, Cout, H, W, HW, TH, TW, tiles_img, tiles, tiles_t, work;
    unsigned x_bytes, u_bytes, y_bytes;
    int stream_stores;
};

struct NoStat {
    __device__ __forceinline__ void add(float) {}
};

__device__ __forceinline__ void stat_add_if(NoStat&, bool, float) {}
__device__ __forceinline__ void stat_add_if(QdStat&, bool, float) {}
__device__ __forceinline__ void stat_add_if(MaxStat& s, bool ok, float v) { s.add(ok ? v : 0.0f); }
template <bool kFast>
__device__ __forceinline__ void stat_add_if(HistStat<kFast>& s, bool ok, float v) { if (ok) s.add(v); }

// The output transform  Y = At M A,  At = [1 1 1 0; 0 1 -1 -1], split between the two waves that hold one (32 channels x 32
// tiles) block.  Each wave has two whole ROWS of M (positions 8 ph .. 8 ph + 7), so the column stage  N = M A  -- (m0 + m1) + m2
// and (m1 - m2) - m3 along a row -- is its own business: two values per row.  The row stage  Y = At N  needs all four rows: the
// wave with rows 0 and 1 makes output row 0 of every tile, (n0 + n1) + n2, and needs n2 for it; the wave with rows 2 and 3 makes
// output row 1, (n1 - n2) - n3, and needs n1.  So each hands ONE row of N (2 values per (channel, tile) pair, 32 registers per
// lane, 8 bytes at a time) to the other through LDS -- 64 (32) KB, once, in the stages the K loop has finished with -- and each
// does half of the transform, the bias, the statistic and the stores.
// kEven: H and W even -- both pixels of a tile row exist together and the pair is 8-byte aligned.
template <int WT, bool kEven, bool kRelu, bool kStream, typename Stat>
__device__ __forceinline__ void wino_epilogue(const f16v (&acc)[8], unsigned ph, unsigned wave, unsigned lane, char* smem, const WArgs& a,
                                              Stat& stat, unsigned kbase, unsigned tbase, unsigned r, unsigned h) {
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y ? a.y : a.relu, 0, a.y ? a.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(kRelu ? a.relu : a.y, 0, a.y_bytes, 0x00020000);
    constexpr int aux = kStream ? 2 : 0;                      // nt
    const unsigned t = tbase + r;
    const bool tile_ok = t < a.tiles;
    const unsigned tc = tile_ok ? t : 0u;
    const unsigned n = tc / a.tiles_img, rem = tc - n * a.tiles_img, ty = rem / a.TW, tx = rem - ty * a.TW;
    const unsigned oy = 2u * ty + ph, ox = 2u * tx;           // this wave's output row of the tile
    const bool row_ok = tile_ok && (kEven || oy < a.H), px1_ok = row_ok && (kEven || ox + 1u < a.W);
    const unsigned off0 = (((n * a.Cout + kbase + 4u * h) * a.H + oy) * a.W + ox) * 4u;     // < 2^31 (host check)
    const unsigned v0 = row_ok ? off0 : kOob, v1 = px1_ok ? off0 + 4u : kOob;
    f4v b4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        b4[q] = a.bias ? *reinterpret_cast<const f4v*>(a.bias + kbase + 8u * q + 4u * h) : f4v{0.f, 0.f, 0.f, 0.f};
    constexpr unsigned kSlot = Geo<WT>::kWaves * 64u;
    f2v* const mine = reinterpret_cast<f2v*>(smem) + wave * 64u + lane;                 // + e * kSlot: [e][wave][lane], 8 bytes each
    const f2v* const theirs = reinterpret_cast<const f2v*>(smem) + (wave ^ (2u * WT)) * 64u + lane;
    // (the accumulators live in AGPRs and are copied out where they are used -- left to itself the compiler copies all of
    //  them at the end of the K loop, in front of the branch that picks the epilogue, and spills to make room)
    float na[16][2], nb[16][2];                               // this wave's two rows of N: [e][output column]
    __syncthreads();                                          // the K loop is done with this memory
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float lo[4], hi[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo[c]) : "a"(acc[c][e]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hi[c]) : "a"(acc[4 + c][e]));
        }
        na[e][0] = (lo[0] + lo[1]) + lo[2]; na[e][1] = (lo[1] - lo[2]) - lo[3];         // N of this wave's first row (0 or 2)
        nb[e][0] = (hi[0] + hi[1]) + hi[2]; nb[e][1] = (hi[1] - hi[2]) - hi[3];         // ... and of its second (1 or 3)
        mine[e * kSlot] = ph == 0 ? f2v{nb[e][0], nb[e][1]} : f2v{na[e][0], na[e][1]};  // rows 0, 1 here: hand over row 1; rows 2, 3: row 2
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const unsigned dm = (e & 3) + 8u * (e >> 2);      // row of this register within the wave's 32 (+ 4 h: in off0)
        const int row4 = (int)(dm * a.HW * 4u);           // uniform
        const f2v got = theirs[e * kSlot];
        float tc4[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
            tc4[c] = ph == 0 ? (na[e][c] + nb[e][c]) + got[c] : (got[c] - na[e][c]) - nb[e][c];
        const float bias = b4[e >> 2][e & 3];
        float o0 = tc4[0] + bias, o1 = tc4[1] + bias;
        o0 = stat_map(stat, o0); o1 = stat_map(stat, o1);
        if (kEven) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, f2v{o0, o1}), yrs, (int)v0, row4, aux);
            if (kRelu)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, f2v{relu_like_torch(o0), relu_like_torch(o1)}), rrs, (int)v0, row4, aux);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o0), yrs, (int)v0, row4, aux);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o1), yrs, (int)v1, row4, aux);
            if (kRelu) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, relu_like_torch(o0)), rrs, (int)v0, row4, aux);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, relu_like_torch(o1)), rrs, (int)v1, row4, aux);
            }
        }
        stat_add_if(stat, row_ok, o0);
        stat_add_if(stat, px1_ok, o1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// What the transform role needs to know about its tile of one work item: the byte offset of the first pixel of each of its
// four input rows in channel 0 of its image (a whole row is one 16-byte load; "row outside the image" = an out-of-range
// offset, answered with zeros) and which columns exist.  Column -1 of a tile at the left edge is the pixel in front of the
// row: it is loaded and dropped -- except in front of the tensor's very first pixel, where that row is loaded from column 0
// and shifted.  The other end needs the same care: the 16 bytes of the last tile of a row end one pixel (even W) or two (odd W)
// behind the row, and behind the LAST row of the LAST image that is the pixel behind the plane -- inside the tensor for every
// channel but the last, whose load would read 4 or 8 bytes past x (the channel rides in the scalar offset, which the address
// unit's range check does not see, so it would really be issued).  Those rows (`endrow`: at most one per tile) are loaded from
// the plane's last four pixels instead and shifted the other way; nothing is ever read outside [x, x + x_bytes).
struct TileIn {
    unsigned xo[4];
    bool c0ok, c2ok, c3ok, shift1;
    int endrow;               // the row of this tile that is (last image, row H - 1, last tile of the row), or -1
};

template <int WT>
__device__ __forceinline__ TileIn tile_in(const WArgs& a, unsigned tb, unsigned lane) {
    TileIn t;
    const unsigned ti = tb * Geo<WT>::kBT + (WT == 1 ? lane & 31u : lane);   // (WT = 1: lanes 32..63 are the same tiles, the wave's second channel)
    const bool tile_ok = ti < a.tiles;
    const unsigned tc = tile_ok ? ti : 0u;
    const unsigned n = tc / a.tiles_img, rem = tc - n * a.tiles_img, ty = rem / a.TW, tx = rem - ty * a.TW;
    const int iy0 = (int)(2u * ty) - 1;
    const unsigned nb = n * a.Cin * a.HW;
    t.c0ok = tx != 0u; t.c2ok = 2u * tx + 1u < a.W; t.c3ok = 2u * tx + 2u < a.W;
    t.shift1 = tile_ok && tc == 0u;
    const bool last_col = tile_ok && tc + a.tiles_img >= a.tiles && tx + 1u == a.TW;        // last image, last tile of a row
    t.endrow = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int iy = iy0 + i;
        const bool ok = tile_ok && (unsigned)iy < a.H;
        const bool end = last_col && (unsigned)iy + 1u == a.H;
        t.xo[i] = ok ? (end ? nb + a.HW - 4u : nb + (unsigned)iy * a.W + 2u * tx - 1u) * 4u : kOob;      // (HW >= 4: host check)
        t.endrow = end ? i : t.endrow;
    }
    if (t.shift1) t.xo[1] = 0u;                               // (H = W = 2: tile 0 is also the last; its rows 1 and 2 are the two cases)
    const unsigned second = WT == 1 ? (lane >> 5) * a.HW * 4u : 0u;   // (an out-of-range offset stays out of range: x is below 2^31 bytes)
#pragma unroll
    for (int i = 0; i < 4; ++i) t.xo[i] += second;
    return t;
}

template <int WT, bool kOddW, typename Stat>
__device__ __forceinline__ void wino_tiles(const WArgs& a, Stat& stat, char* smem) {
    typedef Geo<WT> G_;
    constexpr unsigned kBT = G_::kBT, kKkBytes = G_::kKkBytes, kPlane = G_::kPlane, kStageBytes = G_::kStageBytes;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned r = lane & 31u, h = lane >> 5;
    // the MFMA role: (32 channels wk, 32 tiles wt) of the workgroup's 64 x 32 WT, positions 8 ph .. 8 ph + 7
    const unsigned wk = WT == 2 ? (wave & 3u) >> 1 : wave & 1u, wt = WT == 2 ? wave & 1u : 0u, ph = wave / (2u * WT);
    // the transform role: WT = 2: tile `lane`, input channel `wave` of the step; WT = 1: tile lane % 32, channel 2 wave + lane / 32
    const unsigned tkk = WT == 2 ? wave & 1u : h, tks = WT == 2 ? wave >> 1 : wave, ttile = WT == 2 ? lane : r;
    const unsigned tch = WT == 2 ? wave : 2u * wave;          // (WT = 1: the odd channel of the pair is in the lanes' offsets)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);
    const unsigned vrd = h * kKkBytes + (wt * 32u + r) * 16u + ph * 8u * kPlane;       // B operand reads:  + stage + e * kPlane
    const unsigned vwr = tkk * kKkBytes + ttile * 16u + tks * 4u;                      // transform writes: + stage + e * kPlane
    const unsigned nsteps = a.Cin / kCS;
    const unsigned upos = 2u * a.Cout * 16u;                  // bytes of one position's slice of a step of U
    const unsigned uvo = (h * a.Cout + wk * 32u + r) * 16u;   // + (step, position, k-block): the scalar offset
    // Workgroup g runs on XCD g % 8: give an XCD a contiguous run of work items (k-block major: one slice of U per XCD at a time)
    const unsigned G = gridDim.x, G8 = G & ~7u, g = blockIdx.x;
    const unsigned v0 = g < G8 ? (g & 7u) * (G8 >> 3) + (g >> 3) : g;

    for (unsigned wi = v0; wi < a.work; wi += G) {
        const unsigned kb = wi / a.tiles_t, tb = wi - kb * a.tiles_t;
        const TileIn ti = tile_in<WT>(a, tb, lane);
        // (odd H: row H - 1 is in the last tile row and in the one above it, TW tiles earlier)
        const bool ends = (tb + 1u) * kBT + a.TW >= a.tiles;
        f4v d[4];                                             // this thread's input tile of the step being loaded, row by row
        auto xload = [&](unsigned s) {
            const int so = (int)((s * kCS + tch) * a.HW * 4u);
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)ti.xo[i], so, 0));
        };
        // Bt d B,  Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: rows first, then columns
        auto transform_store = [&](unsigned stage_off) {
            if (tb == 0u) d[1] = ti.shift1 ? f4v{0.0f, d[1][0], d[1][1], d[1][2]} : d[1];     // (uniform: the block that holds tile 0)
            if (ends) {                                       // (uniform: the blocks that hold a tile with an `endrow`)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f4v sh = kOddW ? f4v{d[i][2], d[i][3], 0.0f, 0.0f} : f4v{d[i][1], d[i][2], d[i][3], 0.0f};
                    d[i] = ti.endrow == i ? sh : d[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i][0] = ti.c0ok ? d[i][0] : 0.0f;
                d[i][3] = ti.c3ok ? d[i][3] : 0.0f;
            }
            if (kOddW) {
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i][2] = ti.c2ok ? d[i][2] : 0.0f;
            }
            const f4v t0 = d[0] - d[2], t1 = d[1] + d[2], t2 = d[2] - d[1], t3 = d[1] - d[3];
            const f4v t[4] = {t0, t1, t2, t3};
            char* const w = smem + stage_off + vwr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float*>(w + (unsigned)(4 * i) * kPlane) = t[i][0] - t[i][2];
                *reinterpret_cast<float*>(w + (unsigned)(4 * i + 1) * kPlane) = t[i][1] + t[i][2];
                *reinterpret_cast<float*>(w + (unsigned)(4 * i + 2) * kPlane) = t[i][2] - t[i][1];
                *reinterpret_cast<float*>(w + (unsigned)(4 * i + 3) * kPlane) = t[i][1] - t[i][3];
            }
        };
        f4v ua[8];                                            // A operands of the current step: [position][k pair of the step]
        auto uload = [&](int e, unsigned s) {
            ua[e] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(
                urs, (int)uvo, (int)((s * 16u + 8u * ph + (unsigned)e) * upos + kb * (kBK * 16u)), 0));
        };
        f4v bq[2][4];                                         // B operands: [half of the step][position % 4][k pair]
        auto bload = [&](int half, unsigned stage_off, int e0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bq[half][e] = *reinterpret_cast<const f4v*>(smem + stage_off + vrd + (unsigned)(e0 + e) * kPlane);
        };
        f16v acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[e][i] = 0.0f;
        auto mfma_half = [&](int half, int e0, unsigned sn) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!FQ_WINO_OFF(16)) acc[e0 + e] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[e0 + e][k], bq[half][e][k], acc[e0 + e], 0, 0, 0);
                    else acc[e0 + e][k] += ua[e0 + e][k] * bq[half][e][k];
                }
            if (!FQ_WINO_OFF(2)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) uload(e0 + e, sn);
            }
        };

        // (the loads are issued in the order a step of the loop leaves them in -- x, then U: the wait counts the compiler puts
        //  into the loop are the minimum over both ways into it, and a prologue that issued x last made every step wait for all)
        xload(0);
        transform_store(0u);
        xload(nsteps > 1 ? 1u : 0u);
#pragma unroll
        for (int e = 0; e < 8; ++e) uload(e, 0);
        __syncthreads();
        bload(0, 0u, 0);
        if (FQ_WINO_OFF(64)) bload(1, 0u, 4);
        unsigned cur = 0;                                     // byte offset of the stage step s multiplies out of
        for (unsigned s = 0; s < nsteps; ++s) {
            const unsigned nxt = cur == 2u * kStageBytes ? 0u : cur + kStageBytes;
            const unsigned sn = s + 1 < nsteps ? s + 1 : s;   // (the last step re-reads its own slice of U and re-transforms a step nobody
                                                              //  reads: no branch in the loop -- a branch makes every s_waitcnt behind it
                                                              //  assume the shorter path and wait for loads that were only just issued)
            if (!FQ_WINO_OFF(64)) bload(1, cur, 4);           // the second half's B operands, under the first half's MFMAs
            mfma_half(0, 0, sn);
            if (!FQ_WINO_OFF(4)) transform_store(nxt);
            if (!FQ_WINO_OFF(1)) xload(s + 2 < nsteps ? s + 2 : s);
            if (!FQ_WINO_OFF(32)) __syncthreads();
            if (!FQ_WINO_OFF(64)) bload(0, nxt, 0);           // the next step's first half, under this step's second
            mfma_half(1, 4, sn);
            cur = nxt;
        }
        const unsigned kbase = kb * kBK + wk * 32u, tbase = tb * kBT + wt * 32u;
        // (the epilogue copies the accumulators out with v_accvgpr_read in inline assembly, which the compiler's hazard
        //  recogniser does not see into: the wait states between the last MFMA and the first copy are put here by hand)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        const bool even = ((a.H | a.W) & 1u) == 0u;           // uniform
#define FQ_WINO_EPI(E, R, S) wino_epilogue<WT, E, R, S>(acc, ph, wave, lane, smem, a, stat, kbase, tbase, r, h)
        if (FQ_WINO_OFF(8)) {
            float sum = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int i = 0; i < 16; ++i) sum += acc[e][i];
            if (sum == 12345.678f) a.y[0] = sum;
        } else if (even) {
            if (a.relu) { if (a.stream_stores) FQ_WINO_EPI(true, true, true); else FQ_WINO_EPI(true, true, false); }
            else { if (a.stream_stores) FQ_WINO_EPI(true, false, true); else FQ_WINO_EPI(true, false, false); }
        } else {
            if (a.relu) FQ_WINO_EPI(false, true, false); else FQ_WINO_EPI(false, false, false);
        }
#undef FQ_WINO_EPI
        __syncthreads();                                      // the next work item overwrites stage 0 (the exchange area)
    }
}

#define FQ_WINO_TILES(ST) do { if (a.W & 1u) wino_tiles<WT, true>(a, ST, wino_smem); else wino_tiles<WT, false>(a, ST, wino_smem); } while (0)

template <int WT>
__global__ __launch_bounds__(Geo<WT>::kT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_f32_kernel(const WArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wino_smem[];
    NoStat st;
    FQ_WINO_TILES(st);
}

template <int WT>
__global__ __launch_bounds__(Geo<WT>::kT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_f32_absmax_kernel(const WArgs a, unsigned int* __restrict__ max_bits) {
    extern __shared__ __attribute__((aligned(16))) char wino_smem[];
    MaxStat st;
    FQ_WINO_TILES(st);
    publish_max<Geo<WT>::kT>(st.m, max_bits);
}

// TestConv's forward in one kernel (new_quantity_op.py:283-292): QuanDequan where the value leaves the output transform
template <int WT>
__global__ __launch_bounds__(Geo<WT>::kT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_f32_qd_kernel(const WArgs a, const QdStat qd) {
    extern __shared__ __attribute__((aligned(16))) char wino_smem[];
    QdStat st = qd;
    FQ_WINO_TILES(st);
}

// (every workgroup flushes its 2048 LDS bins with 64-bit atomics at its end)
template <int WT>
__global__ __launch_bounds__(Geo<WT>::kT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_f32_hist_kernel(
    const WArgs a, const float* __restrict__ interval, unsigned long long* __restrict__ hist_row, const int allow_fast) {
    extern __shared__ __attribute__((aligned(16))) char wino_smem[];
    __shared__ unsigned int s_bins[FQ_BINS + kWave];
    for (int b = threadIdx.x; b < FQ_BINS + kWave; b += Geo<WT>::kT) s_bins[b] = 0u;
    __syncthreads();
    const float iv = *interval;
    unsigned int* park = s_bins + FQ_BINS + (threadIdx.x & (kWave - 1));
    if (allow_fast && fast_quotient_ok(iv)) {
        HistStat<true> st{s_bins, park, iv, 1.0f / iv};
        FQ_WINO_TILES(st);
    } else {
        HistStat<false> st{s_bins, park, iv, 1.0f / iv};
        FQ_WINO_TILES(st);
    }
    hist_flush<Geo<WT>::kT>(s_bins, hist_row);
}
#undef FQ_WINO_TILES

// U = G g Gt,  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], in fp64, rounded once; written where the kernel's A operand loads
// find it: [c / 8][e][c % 2][k][c % 8 / 2].  One thread per (k, c).
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, unsigned Cin, unsigned Cout) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= Cin * Cout) return;
    const unsigned k = i / Cin, c = i - k * Cin;
    const float* g = w + (size_t)i * 9u;
    double gg[3][3], t[4][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) gg[a][b] = (double)g[3 * a + b];
    for (int b = 0; b < 3; ++b) {
        t[0][b] = gg[0][b];
        t[1][b] = 0.5 * (gg[0][b] + gg[1][b] + gg[2][b]);
        t[2][b] = 0.5 * (gg[0][b] - gg[1][b] + gg[2][b]);
        t[3][b] = gg[2][b];
    }
    const unsigned cb = c / kCS, kk = c & 1u, ks = (c % kCS) >> 1;
    for (int a = 0; a < 4; ++a) {
        const double o[4] = {t[a][0], 0.5 * (t[a][0] + t[a][1] + t[a][2]), 0.5 * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
        for (int b = 0; b < 4; ++b) {
            const unsigned e = 4u * (unsigned)a + (unsigned)b;
            u[((((size_t)cb * 16u + e) * 2u + kk) * Cout + k) * 4u + ks] = (float)o[b];
        }
    }
}

bool wino_shape_ok(int N, int Cin, int Hin, int Win, int Cout) {
    if (N < 0 || Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0) return false;
    if ((Cin % (int)kCS) != 0 || (Cout % (int)kBK) != 0) return false;
    if ((size_t)Hin * Win < 4) return false;                  // (the end-of-tensor row is loaded from the plane's last four pixels)
    const size_t in_bytes = (size_t)N * Cin * Hin * Win * 4, out_bytes = (size_t)N * Cout * Hin * Win * 4;
    const size_t tiles = (size_t)N * ((Hin + 1) / 2) * ((Win + 1) / 2);
    const size_t work = ((tiles + 31) / 32) * (size_t)(Cout / (int)kBK);                        // (32-bit work item numbers)
    return in_bytes < (1ULL << 31) && out_bytes < (1ULL << 31) && (size_t)Cin * Cout * 64 < (1ULL << 31) && tiles < (1ULL << 30) &&
           work < (1ULL << 31);
}

}  // namespace
}  // namespace fq

using namespace fq;

extern "C" int fq_conv3x3_wino_f32_supported(int N, int Cin, int Hin, int Win, int Cout) { return wino_shape_ok(N, Cin, Hin, Win, Cout) ? 1 : 0; }

extern "C" size_t fq_conv3x3_wino_f32_packed_floats(int Cin, int Cout) {
    return (Cin > 0 && Cout > 0) ? (size_t)16 * (size_t)Cin * (size_t)Cout : 0;
}

extern "C" int fq_conv3x3_wino_f32_pack(const float* w_kcrs, float* u, int Cin, int Cout, fq_stream_t stream) {
    if (!w_kcrs || !u || Cin <= 0 || Cout <= 0) return FQ_ERR_INVALID_ARG;
    if ((Cin % (int)kCS) != 0 || (size_t)Cin * Cout * 64 >= (1ULL << 31)) return FQ_ERR_UNSUPPORTED;
    const unsigned n = (unsigned)Cin * (unsigned)Cout;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((n + 255u) / 256u), dim3(256), 0, as_stream(stream), w_kcrs, u, (unsigned)Cin, (unsigned)Cout);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

// A persistent grid -- as many workgroups as the CUs hold (one of 8 waves or two of 4), each taking every grid-th work item:
// measured 5 % faster than one workgroup per item at 784 items, and the histogram form needs it anyway (one flush per workgroup).
template <int WT>
static int wino_launch_geo(WArgs a, float* max_inout, const float* interval, int64_t* hist_row, const QdStat* qd, fq_stream_t stream) {
    typedef Geo<WT> G;
    a.tiles_t = (a.tiles + G::kBT - 1u) / G::kBT;
    a.work = a.tiles_t * (a.Cout / kBK);
    const unsigned slots = (unsigned)kCUs * (WT == 1 ? 2u : 1u);
    const unsigned grid = a.work < slots ? a.work : slots;
    hipStream_t st = as_stream(stream);
    static bool done_plain[kMaxDevices], done_max[kMaxDevices], done_hist[kMaxDevices], done_qd[kMaxDevices];
    if (qd) {
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(wino_f32_qd_kernel<WT>), (int)G::kLdsBytes, done_qd)) return FQ_ERR_HIP;
        hipLaunchKernelGGL(wino_f32_qd_kernel<WT>, dim3(grid), dim3(G::kT), G::kLdsBytes, st, a, *qd);
    } else if (hist_row) {
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(wino_f32_hist_kernel<WT>), (int)G::kLdsBytes, done_hist)) return FQ_ERR_HIP;
        static const int fast = [] { const char* e = getenv("FQ_HIST_IEEE_DIV"); return (e && e[0] && e[0] != '0') ? 0 : 1; }();
        hipLaunchKernelGGL(wino_f32_hist_kernel<WT>, dim3(grid), dim3(G::kT), G::kLdsBytes, st, a, interval,
                           reinterpret_cast<unsigned long long*>(hist_row), fast);
    } else if (max_inout) {
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(wino_f32_absmax_kernel<WT>), (int)G::kLdsBytes, done_max)) return FQ_ERR_HIP;
        hipLaunchKernelGGL(wino_f32_absmax_kernel<WT>, dim3(grid), dim3(G::kT), G::kLdsBytes, st, a, reinterpret_cast<unsigned int*>(max_inout));
    } else {
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(wino_f32_kernel<WT>), (int)G::kLdsBytes, done_plain)) return FQ_ERR_HIP;
        hipLaunchKernelGGL(wino_f32_kernel<WT>, dim3(grid), dim3(G::kT), G::kLdsBytes, st, a);
    }
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

static int wino_launch(const float* x, const float* u, const float* bias, float* y, float* relu_out, int N, int Cin, int Hin, int Win,
                       int Cout, float* max_inout, const float* interval, int64_t* hist_row, const QdStat* qd, fq_stream_t stream) {
    if (N < 0 || Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0) return FQ_ERR_INVALID_ARG;
    if (max_inout && hist_row) return FQ_ERR_INVALID_ARG;
    if (hist_row && !interval) return FQ_ERR_INVALID_ARG;
    if (N == 0) return FQ_OK;
    if (!x || !u || (!y && !relu_out)) return FQ_ERR_INVALID_ARG;
    if (!wino_shape_ok(N, Cin, Hin, Win, Cout) || (reinterpret_cast<uintptr_t>(u) & 15u) ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 15u)))
        return FQ_ERR_UNSUPPORTED;
    WArgs a;
    a.x = x; a.u = u; a.bias = bias; a.y = y; a.relu = relu_out;
    a.Cin = (unsigned)Cin; a.Cout = (unsigned)Cout; a.H = (unsigned)Hin; a.W = (unsigned)Win; a.HW = a.H * a.W;
    a.TH = (a.H + 1u) / 2u; a.TW = (a.W + 1u) / 2u; a.tiles_img = a.TH * a.TW; a.tiles = (unsigned)N * a.tiles_img;
    a.x_bytes = (unsigned)((size_t)N * Cin * a.HW * 4); a.y_bytes = (unsigned)((size_t)N * Cout * a.HW * 4);
    a.u_bytes = (unsigned)((size_t)16 * Cin * Cout * 4);
    // non-temporal stores beyond the Infinity Cache, where a plane is a whole number of 64-byte blocks (fq_conv1x1_f32.hip)
    a.stream_stores = (size_t)a.y_bytes * (relu_out && y ? 2 : 1) > ((size_t)256 << 20) && (a.HW % 16u) == 0;
    // the shape of a work item (Geo): half items unless the layer has few K steps; FQ_WINO_WT = 1 | 2 forces one (probing)
    static const int forced = [] { const char* e = getenv("FQ_WINO_WT"); return (e && e[0]) ? atoi(e) : 0; }();
    const int wt = (forced == 1 || forced == 2) ? forced : (Cin < 128 ? 2 : 1);
    return wt == 2 ? wino_launch_geo<2>(a, max_inout, interval, hist_row, qd, stream) : wino_launch_geo<1>(a, max_inout, interval, hist_row, qd, stream);
}

extern "C" int fq_conv3x3_wino_f32(const float* x, const float* u, const float* bias, float* y, float* relu_out, int N, int Cin,
                                   int Hin, int Win, int Cout, float* max_inout, const float* interval, int64_t* hist_row,
                                   fq_stream_t stream) {
    return wino_launch(x, u, bias, y, relu_out, N, Cin, Hin, Win, Cout, max_inout, interval, hist_row, nullptr, stream);
}

// TestConv.forward (new_quantity_op.py:283-292) on a stride-1 3x3 layer in one kernel: y = QuanDequan(conv(x) + bias, bit).  The value
// QuanDequan sees is the kernel's own sum -- the one fq_conv3x3_wino_f32 would have stored -- so the result equals
// fq_quandequan_f32 of its output bit for bit.
extern "C" int fq_conv3x3_wino_qd_f32(const float* x, const float* u, const float* bias, float* y, int N, int Cin, int Hin, int Win,
                                      int Cout, int bit, int bitwidth, fq_stream_t stream) {
    if ((bitwidth != 8 && bitwidth != 16) || bit < -120 || bit > 120 || !y) return FQ_ERR_INVALID_ARG;
    QdStat qd;
    qd.scale = ldexpf(1.0f, bit); qd.inv = ldexpf(1.0f, -bit);
    qd.lo = bitwidth == 8 ? -128.0f : -32768.0f; qd.hi = bitwidth == 8 ? 127.0f : 32767.0f;
    return wino_launch(x, u, bias, y, nullptr, N, Cin, Hin, Win, Cout, nullptr, nullptr, nullptr, &qd, stream);
}
