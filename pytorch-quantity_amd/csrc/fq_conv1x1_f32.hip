// fq_conv1x1_f32.hip -- the float 1x1 convolutions of the calibration forward on the fp32 matrix cores, with the
// calibration's statistic taken in the epilogue.
//
// The calibration's wall time is the model's float forward: 36 of ResNet-50's 53 convolutions are
// 1x1, the library runs them at 43-96 TFLOP/s (157 peak), and every hooked convolution output was then read and written
// once more by the bias-add producer (fq_bias_add_absmax_f32 / fq_bias_add_hist_f32).  Here the convolution itself is
// the producer:  y[n][co][p] = sum_ci W[co][ci] * x[n][ci][p*] + bias[co]  with the tensor's abs-max (pass 1,
// distribution_collector.py:70-78) or 2048-bin histogram (pass 2, distribution_collector.py:127-135) taken while the
// value is still in the accumulator registers, and the following nn.ReLU's output written by the same epilogue.
//
// Mapping.  One GEMM over the whole batch: M = Cout, K = Cin, and the N dimension is the flat list of output positions
// (image n, pixel p) -- j = n * HWout + p -- so a 7x7 plane costs no padding (12 544 columns for 256 images, not 256 tiles
// of 49).  NCHW keeps p contiguous for x and y, which is exactly what the f32 MFMA wants: v_mfma_f32_32x32x2_f32 takes
// B[k][j] with j across lanes 0..31 (a 128-byte run of one input plane) and leaves D[i][j] with j across lanes (a
// 128-byte run of one output plane).  The weights are passed transposed, Wt[Cin][Cout], so the A operand's tile is
// k-major with Cout contiguous as well.  A stride-2 convolution (the downsample branches) only changes a column's input
// offset.
//
// Workgroup: 256 threads = 4 waves, tile 128 (Cout) x 128 (columns), K in steps of 8 through three LDS stages (see
// conv1x1_tiles: one barrier per step, in the middle of it).  Each wave owns 64 x 64 = 2 x 2 MFMA tiles: per two k it
// reads 4 dwords from LDS (conflict-free: lanes 0..31 read consecutive words) and issues 4 MFMAs of 64 cycles -- the f32
// matrix rate equals the vector rate, so feeding it is cheap; the short K step keeps the staging registers at 8 and the
// stages at 24 KB, so FOUR workgroups per CU are resident (64 accumulators + <= 64 other registers per lane) and cover
// each other's load latency, barriers and epilogues.
//
// Numerics: the MFMA is bit for bit an fmaf chain over k = 0 .. Cin-1 from 0 (cdna_hip_programming.md "FP32-input
// MFMA"), then one rounding for the bias -- "convolution without bias, then the bias add", as torch does it.  It is
// deterministic (the library's Winograd kernels are not), but it is NOT bit-identical to the library's convolution:
// the float forward of the drop-in never was bit-identical to the reference's CPU forward either; tables are.
#include "fq_common.h"
#include "fq_producer_stat.h"

#include <utility>

namespace fq {
namespace {

constexpr int kT = 256;
constexpr int kBK = 16;                                       // the unit of the K-tail test; K step of the 64-row tiles
constexpr int kStepWide = 8;                                  // K step of the 128 x 128 tiles: 4 workgroups per CU (16: 3; measured +2 %)
template <int WM> constexpr int step_of() { return WM == 2 ? kStepWide : kBK; }
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct C1Args {
    const float* x;
    const float* wt;          // [Cin][Cout]
    const unsigned short* wsb; // the split-bf16 form (conv1x1_tiles_sb): three bf16 planes [3][Cout][Cin], fq_conv1x1_sb_pack
    const float* bias;        // [Cout] or null
    float* y;
    float* relu;              // or null
    const float* res;         // the add form (fq_conv1x1_add_f32): the other operand of the Eltwise that consumes y, y's shape
    float* sum;               //   where y + res goes, or null (nobody reads it: only its ReLU is handed on)
    int store_y;              //   0: y itself is not written either (only its statistic is wanted)
    unsigned Cin, Cout, HWin, HWout, Win, Wout, stride;
    int Hin, R, S, pad;       // (R x S taps, zero padding: the K x K form; 1, 1, 0 for the 1x1 form)
    unsigned x_bytes, w_bytes, y_bytes;
    unsigned cols;            // N * HWout
    unsigned tiles_m, tiles;
    int stream_stores;
    // The tail split (plan_split): tiles >= split_first are computed by split_s workgroups each, one K slice per workgroup;
    // the partial accumulators meet in `ws` and the workgroup that arrives last (ws_count) sums them in slice order and
    // runs the epilogue.  work = split_first + (tiles - split_first) * split_s items; no split: split_first = tiles.
    unsigned split_first, split_s, work;
    float* ws;
    unsigned* ws_count;
#ifdef FQ_C1_ABLATE
    int ablate;               // debug build only (scripts/conv1x1_ablate.py): 1 no stores, 2 global loads of the first K step only, 4 no barriers (wrong results)
#endif
};

#ifdef FQ_C1_ABLATE
#define FQ_C1_OFF(bit) (a.ablate & (bit))
#else
#define FQ_C1_OFF(bit) false
#endif

struct NoStat {
    __device__ __forceinline__ void add(float) {}
};

// the add form's pair of statistics: the convolution output's and the sum's running abs-max
// (kMayStore = false: the form that never writes the two tensors themselves -- pass 2, where nothing is kept)
template <typename SC, typename SS, bool kStores>
struct AddStat {
    static constexpr bool kMayStore = kStores;
    SC c;
    SS s;
};
template <typename T> struct is_add_stat { static constexpr bool value = false; };
template <typename SC, typename SS, bool K> struct is_add_stat<AddStat<SC, SS, K>> { static constexpr bool value = true; };

template <int I>
struct Stage {
    static constexpr int value = I;
};

// Tile shapes: the 4 waves sit 2 x 2, each owns WM x WN MFMA tiles of 32 x 32 -> the workgroup tile is (64 WM) x (64 WN).
// <2,2> = 128 x 128 for Cout >= 128; <1,2> = 64 x 128 for the 64-channel layers (no empty half tile).
template <int WM, int WN, int BK = kBK>
struct Shape {
    static constexpr int BM = 64 * WM, BN = 64 * WN;
    static constexpr int kXRows = BK * BN / kT;              // x-tile dwords per thread and K step (column fixed per thread)
    static constexpr int kXStep = kT / BN > 0 ? kT / BN : 1;  // rows between them (BN = 128: 2; BN = 256: 1)
    static constexpr int kWVecs = BK * BM / 4 / kT;          // W-tile float4 per thread and K step
    static constexpr int kWRowStep = kT / (BM / 4);           // rows between them
    static constexpr int kFloats = 3 * BK * (BM + BN) + BM;
    static_assert(kXRows >= 1 && kWVecs >= 1 && BK % 4 == 0, "tile too small for 256 loading threads");  // three stages of both tiles + the bias slice
};

// Epilogue of one workgroup tile.  D[i][j] has j = lane & 31 and i = (e & 3) + 8 (e >> 2) + 4 (lane >> 5): the four rows
// of a register quad are consecutive, so their biases come as one 16-byte LDS read; the ReLU copy, the store flavour and
// "every row of this tile exists" are compile-time here (uniform per launch / per tile), so a value costs its add, its
// store(s) and the statistic -- no branch, no LDS wait per value.
template <int WM, int WN, bool kRelu, bool kStream, bool kFullM, typename Stat>
__device__ __forceinline__ void c1_epilogue(const f16v (&acc)[WM][WN], const C1Args& a, Stat& stat, const float* s_bias,
                                            unsigned jbase, unsigned mbase, unsigned m0, unsigned n0, unsigned r, unsigned h) {
    // stores through buffer descriptors too: per-lane byte offset of (column, first row of the lane) once per column
    // block, the row advance as the scalar offset -- a value costs its bias add, the ReLU select and the statistic
    // (y == null with a ReLU copy: only the ReLU's output is wanted -- a descriptor of zero records drops every store of y in
    //  the address unit, the loop below stays as it is)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y ? a.y : a.relu, 0, a.y ? a.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(kRelu ? a.relu : a.y, 0, a.y_bytes, 0x00020000);
    constexpr int aux = kStream ? 2 : 0;                      // nt
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const unsigned jn = jbase + n0 + 32u * ni + r;
        if (jn < a.cols) {
            const unsigned n = jn / a.HWout, p = jn - n * a.HWout;
            const unsigned col4 = (n * a.Cout * a.HWout + p + (mbase + m0 + 4u * h) * a.HWout) * 4u;   // < 2^32 (host check)
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) {
                f4v b4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) b4[q] = *reinterpret_cast<const f4v*>(s_bias + m0 + 32u * mi + 8u * q + 4u * h);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned dm = 32u * mi + (e & 3) + 8u * (e >> 2);           // compile-time row within the wave tile
                    if (kFullM || mbase + m0 + 4u * h + dm < a.Cout) {
                        const float val = stat_map(stat, acc[mi][ni][e] + b4[e >> 2][e & 3]);
                        const int row4 = (int)(dm * a.HWout * 4u);                     // uniform
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yrs, (int)col4, row4, aux);
                        if (kRelu)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, relu_like_torch(val)), rrs, (int)col4,
                                                                  row4, aux);
                        stat.add(val);
                    }
                }
            }
        }
    }
}

// Epilogue of the add form (conv3 + Eltwise + ReLU of a bottleneck in one kernel, fabu_layer.py:5-11 behind nn.Conv2d): per value
// v = acc + bias (rounded: "the convolution's output", whose abs-max goes to the first statistic and which is stored only when
// somebody keeps it), s = v + res (the Eltwise's output: second statistic, stored only when kept), max(s, 0) to a.relu.
// Separately the two kernels move 20 bytes per element (4 written, 8 read, 8 written); this moves 8 when nothing is kept.
template <int WM, int WN, bool kStoreY, bool kStoreSum, bool kStream, typename Stat>
__device__ __forceinline__ void c1_epilogue_add(const f16v (&acc)[WM][WN], const C1Args& a, Stat& stat, const float* s_bias,
                                                unsigned jbase, unsigned mbase, unsigned m0, unsigned n0, unsigned r, unsigned h) {
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(kStoreY ? a.y : a.relu, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(kStoreSum ? a.sum : a.relu, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(a.relu, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.y_bytes, 0x00020000);
    constexpr int aux = kStream ? 2 : 0;                      // nt
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const unsigned jn = jbase + n0 + 32u * ni + r;
        if (jn < a.cols) {
            const unsigned n = jn / a.HWout, p = jn - n * a.HWout;
            const unsigned col4 = (n * a.Cout * a.HWout + p + (mbase + m0 + 4u * h) * a.HWout) * 4u;   // < 2^32 (host check)
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) {
                float rv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned dm = 32u * mi + (e & 3) + 8u * (e >> 2);
                    rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ers, (int)col4, (int)(dm * a.HWout * 4u), aux));
                }
                f4v b4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) b4[q] = *reinterpret_cast<const f4v*>(s_bias + m0 + 32u * mi + 8u * q + 4u * h);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned dm = 32u * mi + (e & 3) + 8u * (e >> 2);
                    const int row4 = (int)(dm * a.HWout * 4u);                         // uniform
                    const float val = acc[mi][ni][e] + b4[e >> 2][e & 3];
                    stat.c.add(val);
                    if (kStoreY) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yrs, (int)col4, row4, aux);
                    const float sm = val + rv[e];
                    stat.s.add(sm);
                    if (kStoreSum) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sm), srs, (int)col4, row4, aux);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, relu_like_torch(sm)), rrs, (int)col4, row4, aux);
                }
            }
        }
    }
}

// What happens to a finished accumulator tile -- shared by the fp32 and the split-bf16 K loops.  A K slice of a split tile goes to the
// workspace and the last slice to arrive sums them in slice order; then the epilogue the launch asked for.
template <int WM, int WN, typename Stat>
__device__ __forceinline__ void finish_tile(f16v (&acc)[WM][WN], const C1Args& a, Stat& stat, const float* s_bias, bool split, unsigned t,
                                            unsigned slice, unsigned jbase, unsigned mbase, unsigned m0, unsigned n0, unsigned r, unsigned h,
                                            unsigned tid, unsigned wave) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    bool finish = true;
    if (split) {
        // This workgroup holds one K slice of the tile.  Its accumulators go to the workspace; the workgroup of the tile
        // that gets there last adds all slices up IN SLICE ORDER (so the result does not depend on who was last) and
        // carries on with the epilogue.  Release / acquire at agent scope as MI355X_MICROARCH.md prescribes for a
        // counter hand-off: every storing wave waits for its stores, a barrier, one lane releases and adds; the lane
        // whose add completes the count acquires, a barrier, then plain loads.
        const unsigned rr = t - a.split_first;
        constexpr unsigned kTileFloats = (unsigned)(BM * BN);
        float* const mine = a.ws + ((size_t)rr * a.split_s + slice) * kTileFloats;
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4v v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                    *reinterpret_cast<f4v*>(mine + ((((mi * WN + ni) * 4 + q) * kT) + tid) * 4u) = v;
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ unsigned s_last;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned prev = __hip_atomic_fetch_add(a.ws_count + rr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = prev + 1u == a.split_s;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.ws_count + rr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            s_last = last;
        }
        __syncthreads();
        finish = s_last != 0u;                            // uniform
        if (finish) {
            if (wave != 0) {                              // (wave 0's lane 0 made the acquire: its L1 is this CU's L1 -- one
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   //  invalidate per CU would do, one per wave is cheap here)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
            for (unsigned sl = 0; sl < a.split_s; ++sl) {
                const float* part = a.ws + ((size_t)rr * a.split_s + sl) * kTileFloats;
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f4v v = *reinterpret_cast<const f4v*>(part + ((((mi * WN + ni) * 4 + q) * kT) + tid) * 4u);
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[mi][ni][4 * q + c] += v[c];
                        }
            }
        }
    }
    if (finish) {
        const bool full_m = mbase + BM <= a.Cout;
#ifdef FQ_C1_ABLATE
        if (FQ_C1_OFF(1)) {
        } else
#endif
#define FQ_C1_EPI(R, S, F) c1_epilogue<WM, WN, R, S, F>(acc, a, stat, s_bias, jbase, mbase, m0, n0, r, h)
#define FQ_C1_ADD(Y, S, N) c1_epilogue_add<WM, WN, Y, S, N>(acc, a, stat, s_bias, jbase, mbase, m0, n0, r, h)
        if constexpr (is_add_stat<Stat>::value) {         // (whole row tiles only: host check)
            if constexpr (!Stat::kMayStore) {
                if (a.stream_stores) FQ_C1_ADD(false, false, true); else FQ_C1_ADD(false, false, false);
            } else
            if (a.stream_stores) {
                if (a.store_y) { if (a.sum) FQ_C1_ADD(true, true, true); else FQ_C1_ADD(true, false, true); }
                else { if (a.sum) FQ_C1_ADD(false, true, true); else FQ_C1_ADD(false, false, true); }
            } else {
                if (a.store_y) { if (a.sum) FQ_C1_ADD(true, true, false); else FQ_C1_ADD(true, false, false); }
                else { if (a.sum) FQ_C1_ADD(false, true, false); else FQ_C1_ADD(false, false, false); }
            }
        } else
        if (a.relu) {
            if (a.stream_stores) { if (full_m) FQ_C1_EPI(true, true, true); else FQ_C1_EPI(true, true, false); }
            else { if (full_m) FQ_C1_EPI(true, false, true); else FQ_C1_EPI(true, false, false); }
        } else {
            if (a.stream_stores) { if (full_m) FQ_C1_EPI(false, true, true); else FQ_C1_EPI(false, true, false); }
            else { if (full_m) FQ_C1_EPI(false, false, true); else FQ_C1_EPI(false, false, false); }
        }
#undef FQ_C1_EPI
#undef FQ_C1_ADD
    }
}

// kMode: 0 = 1x1, Cin a multiple of the K step; 1 = 1x1 with a K tail; 2 = R x S taps with zero padding (Cin a multiple of 16)
template <int WM, int WN, int BK, int kMode, typename Stat>
__device__ __forceinline__ void conv1x1_tiles(const C1Args& a, Stat& stat, float* smem) {
    constexpr bool kTailK = kMode == 1, kTaps = kMode == 2;
    typedef Shape<WM, WN, BK> S;
    constexpr int BM = S::BM, BN = S::BN;
    float* Ws = smem;                                         // [3][BK][BM]
    float* Xs = smem + 3 * BK * BM;                          // [3][BK][BN]
    float* s_bias = smem + 3 * BK * (BM + BN);
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned r = lane & 31u, h = lane >> 5;
    const unsigned m0 = (wave >> 1) * (32u * WM), n0 = (wave & 1u) * (32u * WN);
    // Workgroup g runs on XCD g % 8, each with its own L2: give an XCD a contiguous run of tiles, so the m-tiles that
    // share one x tile (and the column tiles that share one W tile) meet in the same L2.
    // (the K slices of the tail split keep their launch order: they are the last workgroups to start, spread over all XCDs)
    const unsigned G = gridDim.x, G8 = (G < a.split_first ? G : a.split_first) & ~7u, g = blockIdx.x;
    const unsigned v0 = g < G8 ? (g & 7u) * (G8 >> 3) + (g >> 3) : g;
    const unsigned xc = tid % BN, xk = tid / BN;              // x tile: this thread's column, rows xk + kXStep * i
    const unsigned wr = tid / (BM / 4), wc = (tid % (BM / 4)) * 4u;   // W tile: rows wr + kWRowStep * i, columns wc .. wc + 3
    // K x K: the reduction runs tap by tap over the same x rows, shifted -- a 1x1 convolution per tap whose per-thread
    // pixel offset (or "outside the image: zero") is worked out once per tap
    const unsigned nk = kTaps ? (unsigned)(a.R * a.S) * (a.Cin / BK) : (a.Cin + BK - 1) / BK;
    // x and Wt through buffer descriptors: address = descriptor base + scalar offset (the K row, advanced with scalar adds)
    // + 32-bit per-thread byte offset (computed once per tile) -- no vector instruction per load
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wt), 0, a.w_bytes, 0x00020000);
    // Every LDS address of the K loop is one of these per-thread bases plus a compile-time offset (the stage index is a
    // template constant: the loop is unrolled over the three stages), and every global address is a uniform base (scalar
    // registers, advanced by scalar adds) plus a 32-bit per-thread byte offset computed once per tile.  A vector
    // instruction of ANY wave of a SIMD takes issue cycles from its matrix pipe (scripts/mfma_f32_probe.hip: 99 % of the
    // matrix cycles with register operands, 93 % with this loop's LDS reads and barriers): the first form of this loop
    // spent 28 vector instructions per 16 MFMAs on addresses and ran at 82 %.
    const float* wrd[WM];                                     // A operand reads: wrd[mi] + (stage * BK + kk) * BM
    const float* xrd[WN];                                     // B operand reads: xrd[ni] + (stage * BK + kk) * BN
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
        unsigned i0 = h * BM + m0 + r + 32 * mi;
        asm volatile("" : "+v"(i0));                          // (unrelated bases: two ds_read_b32 with 16-bit immediates each,
        wrd[mi] = Ws + i0;                                    //  not one ds_read2_b32 whose 8-bit offsets need a vector add)
    }
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        unsigned i0 = h * BN + n0 + r + 32 * ni;
        asm volatile("" : "+v"(i0));
        xrd[ni] = Xs + i0;
    }
    float* const ww0 = Ws + wr * BM + wc;                     // W tile stores:   + (stage * BK + kWRowStep * i) * BM
    float* const xw0 = Xs + xk * BN + xc;                     // x tile stores:   + (stage * BK + kXStep * i) * BN

    for (unsigned wi = v0; wi < a.work; wi += G) {
        unsigned t = wi, ks_begin = 0, ks_end = nk, slice = 0;
        const bool split = wi >= a.split_first;                // uniform
        if (split) {
            const unsigned rr = (wi - a.split_first) / a.split_s;
            slice = (wi - a.split_first) - rr * a.split_s;
            t = a.split_first + rr;
            ks_begin = slice * nk / a.split_s;
            ks_end = (slice + 1u) * nk / a.split_s;
        }
        const unsigned ct = t / a.tiles_m, mt = t - ct * a.tiles_m;
        const unsigned mbase = mt * BM, jbase = ct * BN;
        // Loads never leave the tensors: a column past the end re-reads the last one, a W column past Cout the last four
        // (those accumulators are never stored), so no load is predicated; only a K tail needs zeros.
        unsigned xo[S::kXRows], wo[S::kWVecs];                // byte offsets of this thread's loads in K step 0 (< 2^32: host check)
        int iy0 = 0, ix0 = 0;                                 // K x K: top-left input pixel of this thread's column
        unsigned nbase = 0;
        bool tap_ok = true;
        int ld_r = 0, ld_s = 0;                               // K x K: the tap the NEXT gload reads (uniform), its channel row
        unsigned ld_kb = 0;
        auto tap_offsets = [&]() {                            // per-thread pixel of tap (ld_r, ld_s), or "zero"
            const int iy = iy0 + ld_r, ix = ix0 + ld_s;
            tap_ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const unsigned pix = tap_ok ? nbase + (unsigned)iy * a.Win + (unsigned)ix : 0u;
#pragma unroll
            for (int i = 0; i < S::kXRows; ++i) xo[i] = (pix + (xk + S::kXStep * i) * a.HWin) * 4u;
        };
        {
            const unsigned j = min(jbase + xc, a.cols - 1u);
            const unsigned n = j / a.HWout, p = j - n * a.HWout;
            if (kTaps) {
                const unsigned oh = p / a.Wout, ow = p - oh * a.Wout;
                iy0 = (int)(oh * a.stride) - a.pad;
                ix0 = (int)(ow * a.stride) - a.pad;
                nbase = n * a.Cin * a.HWin;
                if (ks_begin) {                               // a K slice starts inside the tap sequence
                    const unsigned per_tap = a.Cin / BK, tap = ks_begin / per_tap;
                    ld_r = (int)(tap / (unsigned)a.S);
                    ld_s = (int)(tap - (unsigned)ld_r * (unsigned)a.S);
                    ld_kb = (ks_begin - tap * per_tap) * BK;
                }
                tap_offsets();
            } else {
                unsigned pin = p;
                if (a.stride != 1) {
                    const unsigned oh = p / a.Wout, ow = p - oh * a.Wout;
                    pin = oh * a.stride * a.Win + ow * a.stride;
                }
                const unsigned xoff = n * a.Cin * a.HWin + pin;
#pragma unroll
                for (int i = 0; i < S::kXRows; ++i) xo[i] = (xoff + (xk + S::kXStep * i) * a.HWin) * 4u;
            }
            const unsigned woff = min(mbase + wc, a.Cout - 4u);
#pragma unroll
            for (int i = 0; i < S::kWVecs; ++i) wo[i] = ((wr + S::kWRowStep * i) * a.Cout + woff) * 4u;
        }
        float xr[S::kXRows];
        f4v wreg[S::kWVecs];
        auto gload = [&](unsigned kb) {                       // kb is uniform: the row advance is scalar arithmetic
            const int xs = (int)((kTaps ? ld_kb : kb) * a.HWin * 4u), ws = (int)(kb * a.Cout * 4u);
#pragma unroll
            for (int i = 0; i < S::kXRows; ++i) {
                if (kTailK) {                                 // rows past Cin: re-read row Cin - 1, use zero
                    const unsigned k = kb + xk + S::kXStep * i, back = k < a.Cin ? 0u : (k - (a.Cin - 1u)) * a.HWin * 4u;
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(xo[i] - back), xs, 0));
                    xr[i] = k < a.Cin ? v : 0.0f;
                } else if (kTaps) {
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)xo[i], xs, 0));
                    xr[i] = tap_ok ? v : 0.0f;
                } else {
                    xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)xo[i], xs, 0));
                }
            }
#pragma unroll
            for (int i = 0; i < S::kWVecs; ++i) {
                if (kTailK) {
                    const unsigned k = kb + wr + S::kWRowStep * i, back = k < a.Cin ? 0u : (k - (a.Cin - 1u)) * a.Cout * 4u;
                    const f4v v = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(wo[i] - back), ws, 0));
                    wreg[i] = k < a.Cin ? v : f4v{0.f, 0.f, 0.f, 0.f};
                } else {
                    wreg[i] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)wo[i], ws, 0));
                }
            }
            if (kTaps) {                                      // gload is called once per K step, in order: advance to the next rows / tap
                ld_kb += BK;
                if (ld_kb == a.Cin) {
                    ld_kb = 0;
                    if (++ld_s == a.S) { ld_s = 0; ++ld_r; }
                    tap_offsets();
                }
            }
        };
        auto lstore = [&](auto stage) {
            constexpr int buf = decltype(stage)::value;
#pragma unroll
            for (int i = 0; i < S::kXRows; ++i) xw0[(buf * BK + S::kXStep * i) * BN] = xr[i];
#pragma unroll
            for (int i = 0; i < S::kWVecs; ++i) *reinterpret_cast<f4v*>(ww0 + (buf * BK + S::kWRowStep * i) * BM) = wreg[i];
        };
        f16v acc[WM][WN];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

        // K pipeline over THREE LDS stages.  Step s multiplies out of stage s % 3; in the middle of it every wave stores
        // the operands of step s + 1 (loaded one step earlier) into stage (s + 1) % 3, issues the global loads of step
        // s + 2 and meets the others at the step's only barrier.  That stage was last read in step s - 2, which every wave
        // had finished before it could pass the barrier of step s - 1; and what is read at the start of step s + 1 was
        // complete half a step earlier.  So no wave waits for data at a step boundary: the first operands of the next
        // step are fetched under the last MFMAs of this one, and the MFMA stream of a wave does not stop between steps.
        gload(ks_begin * BK);
        if (tid < (unsigned)BM) s_bias[tid] = (a.bias && mbase + tid < a.Cout) ? a.bias[mbase + tid] : 0.0f;
        lstore(Stage<0>{});
        if (ks_begin + 1 < ks_end) gload((ks_begin + 1) * BK);
        __syncthreads();
        float fa[2][WM], fb[2][WN];                           // operands of this and of the next k pair
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[0][mi] = wrd[mi][0];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) fb[0][ni] = xrd[ni][0];
        auto kstep = [&](auto stage, unsigned ks) {
            constexpr int cur = decltype(stage)::value, nxt = (cur + 1) % 3;
            const bool more = ks + 1 < ks_end;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const int c = (kk >> 1) & 1, nx = c ^ 1;
                if (kk == BK / 2 && more) {
                    lstore(Stage<nxt>{});
                    if (ks + 2 < ks_end && !FQ_C1_OFF(2)) gload((ks + 2) * BK);
                    if (!FQ_C1_OFF(4)) __syncthreads();
                }
                if (kk + 2 < BK) {
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) fa[nx][mi] = wrd[mi][(cur * BK + kk + 2) * BM];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) fb[nx][ni] = xrd[ni][(cur * BK + kk + 2) * BN];
                } else if (more) {
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) fa[nx][mi] = wrd[mi][nxt * BK * BM];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) fb[nx][ni] = xrd[ni][nxt * BK * BN];
                }
                __builtin_amdgcn_sched_barrier(0);            // keep the next pair's LDS reads ahead of these MFMAs
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][mi], fb[c][ni], acc[mi][ni], 0, 0, 0);
            }
        };
        static_assert((BK / 2) % 2 == 0, "a K step must hold an even number of k pairs: the operand double buffer starts each step at 0");
        for (unsigned ks = ks_begin;;) {
            kstep(Stage<0>{}, ks);
            if (++ks >= ks_end) break;
            kstep(Stage<1>{}, ks);
            if (++ks >= ks_end) break;
            kstep(Stage<2>{}, ks);
            if (++ks >= ks_end) break;
        }
        finish_tile<WM, WN>(acc, a, stat, s_bias, split, t, slice, jbase, mbase, m0, n0, r, h, tid, wave);
        __syncthreads();                                      // the next tile overwrites s_bias and stage 0
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same GEMM with every fp32 operand as THREE bf16 values (conv1x1_tiles_sb, fq_conv1x1_sb_f32 and friends).
//
// The fp32 MFMA runs at the vector rate -- 1/16 of the bf16 rate on this chip (scripts/bf16x3_probe.hip: 145-154 TFLOP/s
// against 2.1-2.2 PFLOP/s).  An fp32 value IS the sum of three bf16 values (hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi -
// mid): 8 + 8 + 8 significant bits, both residuals exact), so a product is nine bf16 products; the three smallest (mid lo, lo mid,
// lo lo: below 2^-23 of the product) are left out and the other six accumulate in fp32 inside v_mfma_f32_32x32x16_bf16.
// Measured against fp64 (the probe; tests/test_gpu_float_forward_kernels.py): 1.0-1.3e-7 of sum |w||x| -- the fp32 fma chain
// of conv1x1_tiles: 1.4-2.1e-7 -- at 6 MFMAs of 32 cycles per 16 of K instead of 8 of 64.  Integer-valued operands below 2^8
// have no mid and lo part: such data is exact, as before.  (A non-finite input gives NaN: inf - inf in its residual.)
//
// The weights are split once (fq_conv1x1_sb_pack: three planes [Cout][Cin], k contiguous -- the A operand of the MFMA wants a
// row's 8 consecutive k in one lane); the activations are split by the threads that stage the x tile: thread = (column, half of
// the K step) loads its 8 k, splits them (~5 vector instructions per value, once per 128 output channels) and writes three
// 16-byte pieces.  LDS per stage: [plane 3][k / 8: 2][row or column][8 bf16] for both operands -- every fragment read is one
// conflict-free ds_read_b128.  Same three-stage pipeline with one barrier in the middle of a step, same accumulator layout,
// hence the same epilogues, statistics and tail split as the fp32 kernel.
#ifndef FQ_SB_ABLATE
#define FQ_SB_ABLATE 0      // debug builds only (scripts/_dbg/build_sb_variants.sh): 1 no x loads after the first two K steps, 2 no W loads, 4 no split, 8 no LDS stores -- wrong results, timing only
#endif
#define FQ_SB_OFF(bit) ((FQ_SB_ABLATE) & (bit))
constexpr int kSbBK = 16;
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int WM, int WN>
struct ShapeSb {
    static constexpr int BM = 64 * WM, BN = 64 * WN;
    static constexpr int kStage = 3 * 2 * BN * 16;                     // the x tile of one K step: three planes, 12 KB
    static constexpr int kBytes = 3 * kStage + BM * 4;                 // three stages + the bias slice
    static_assert(BN == 128, "the x tile is staged by 128 columns x 2 halves of the K step = 256 threads");
};

// v -> (hi, mid, lo) as bf16 bit patterns, round to nearest even (the residuals are exact in fp32)
__device__ __forceinline__ unsigned bf16_bits(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void split3(float v, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = bf16_bits(v);
    const float r1 = v - __builtin_bit_cast(float, hi << 16);
    mid = bf16_bits(r1);
    const float r2 = r1 - __builtin_bit_cast(float, mid << 16);
    lo = bf16_bits(r2);
}

// Where the time of the first form went (profiles/r04_conv1x1_split_bf16_ablation.txt): not into the split (compiled out: -2 %) but
// into LDS -- a bf16 MFMA eats its operands 12 x faster per matrix cycle than the fp32 one, and with both operands staged the LDS ran at
// ~ 90 of its 128 bytes per cycle and CU.  So the WEIGHTS do not go through LDS at all: the pack is laid out [k / 16][plane][k % 16 / 8]
// [Cout][8] and a lane's fragment is ONE 16-byte global load (coalesced: consecutive rows are consecutive pieces; L2-resident, the
// second wave with the same rows hits L1), issued a whole K step ahead into the second of two fragment sets.  Only the x tile is staged.
template <int WM, int WN, typename Stat>
__device__ __forceinline__ void conv1x1_tiles_sb(const C1Args& a, Stat& stat, char* smem) {
    typedef ShapeSb<WM, WN> S;
    constexpr int BM = S::BM, BN = S::BN;
    float* const s_bias = reinterpret_cast<float*>(smem + 3 * S::kStage);
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned r = lane & 31u, h = lane >> 5;
    const unsigned m0 = (wave >> 1) * (32u * WM), n0 = (wave & 1u) * (32u * WN);
    const unsigned G = gridDim.x, G8 = (G < a.split_first ? G : a.split_first) & ~7u, g = blockIdx.x;
    const unsigned v0 = g < G8 ? (g & 7u) * (G8 >> 3) + (g >> 3) : g;
    const unsigned xc = tid % BN, xh = tid / BN;                  // x tile: this thread's column, its half of the K step
    const unsigned nk = a.Cin / kSbBK;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wsb), 0, a.w_bytes, 0x00020000);
    const unsigned brd = (h * BN + n0 + r) * 16u;                 // B fragment reads: + stage + (plane * 2 * BN + 32 ni) * 16
    const unsigned bwr = (xh * BN + xc) * 16u;                    // x tile stores:    + stage + plane * 2 * BN * 16
    const unsigned wplane = 2u * a.Cout * 16u, wstep = 3u * wplane;   // bytes of one plane / of one K step of the pack

    for (unsigned wi = v0; wi < a.work; wi += G) {
        unsigned t = wi, ks_begin = 0, ks_end = nk, slice = 0;
        const bool split = wi >= a.split_first;                // uniform
        if (split) {
            const unsigned rr = (wi - a.split_first) / a.split_s;
            slice = (wi - a.split_first) - rr * a.split_s;
            t = a.split_first + rr;
            ks_begin = slice * nk / a.split_s;
            ks_end = (slice + 1u) * nk / a.split_s;
        }
        const unsigned ct = t / a.tiles_m, mt = t - ct * a.tiles_m;
        const unsigned mbase = mt * BM, jbase = ct * BN;
        // x: byte offset of (this column, k = 8 xh) in K step 0; the 8 rows follow at HWin * 4 (scalar), the K step is scalar too
        unsigned xo;
        {
            const unsigned j = min(jbase + xc, a.cols - 1u);
            const unsigned n = j / a.HWout, p = j - n * a.HWout;
            unsigned pin = p;
            if (a.stride != 1) {
                const unsigned oh = p / a.Wout, ow = p - oh * a.Wout;
                pin = oh * a.stride * a.Win + ow * a.stride;
            }
            xo = (n * a.Cin * a.HWin + pin + 8u * xh * a.HWin) * 4u;
        }
        // W fragments: this lane's row of each 32-row block (rows past Cout re-read the last row: those accumulators are never stored)
        unsigned wfo[WM];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) wfo[mi] = (h * a.Cout + min(mbase + m0 + 32u * mi + r, a.Cout - 1u)) * 16u;
        // the x values of step k wait in register set k % 2 (relative to ks_begin): they are loaded TWO steps before they are split
        // and stored (one step was not enough latency cover: the staging waited for them)
        float xr[2][8];
        auto gload = [&](int set, unsigned ks) {               // ks is uniform: the K advance is scalar arithmetic
            const unsigned xs = ks * kSbBK * a.HWin * 4u;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xr[set][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)xo, (int)(xs + (unsigned)i * a.HWin * 4u), 0));
        };
        auto lstore = [&](int set, unsigned stage_off) {
            // two values at a time: v_cvt_pk_bf16_f32 (round to nearest even), the two halves widened again by a shift and a mask,
            // one packed subtraction for the residuals -- 4.5 vector instructions per value
            u4v ph, pm, pl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2v v = {xr[set][2 * i], xr[set][2 * i + 1]};
                const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2v));
                const f2v r1 = v - f2v{__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
                const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf2v));
                const f2v r2 = r1 - f2v{__builtin_bit_cast(float, mb << 16), __builtin_bit_cast(float, mb & 0xffff0000u)};
                ph[i] = hb; pm[i] = mb; pl[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf2v));
            }
            *reinterpret_cast<u4v*>(smem + stage_off + bwr) = ph;
            *reinterpret_cast<u4v*>(smem + stage_off + bwr + 2u * BN * 16u) = pm;
            *reinterpret_cast<u4v*>(smem + stage_off + bwr + 4u * BN * 16u) = pl;
        };
        f16v acc[WM][WN];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

        // Two sets of operand fragments: a step multiplies out of one while the other is filled with the fragments of the following
        // step -- the weights' from global memory at the start of the step, the activations' from LDS behind the step's barrier.
        bf8v fa[2][WM][3], fb[2][WN][3];
        auto wfrags = [&](int set, unsigned ks) {
            const unsigned ws = ks * wstep;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
                    fa[set][mi][pl] = __builtin_bit_cast(bf8v, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)wfo[mi], (int)(ws + (unsigned)pl * wplane), 0));
        };
        auto xfrags = [&](int set, unsigned stage_off) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    fb[set][ni][pl] = *reinterpret_cast<const bf8v*>(smem + stage_off + brd + (unsigned)(pl * 2 * BN + 32 * ni) * 16u);
        };
        // (x first, then W: the order the loop leaves the loads in -- the wait counts in the loop are the minimum over both ways into it)
        gload(0, ks_begin);
        wfrags(0, ks_begin);
        if (tid < (unsigned)BM) s_bias[tid] = (a.bias && mbase + tid < a.Cout) ? a.bias[mbase + tid] : 0.0f;
        if (ks_begin + 1 < ks_end) gload(1, ks_begin + 1);
        lstore(0, 0u);
        if (ks_begin + 2 < ks_end) gload(0, ks_begin + 2);
        __syncthreads();
        xfrags(0, 0u);
        unsigned cur = 0;
        auto kstep = [&](auto set_c, unsigned ks) {
            constexpr int set = decltype(set_c)::value;
            const unsigned nxt = cur == 2u * S::kStage ? 0u : cur + (unsigned)S::kStage;
            if (ks + 1 < ks_end) wfrags(set ^ 1, ks + 1);
            // smallest products first: (hi, lo), (lo, hi), (mid, mid), (hi, mid), (mid, hi), (hi, hi)
            constexpr int kA[6] = {0, 2, 1, 0, 1, 0}, kB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                if (q == 3) {                                   // the middle of the step: the x tile of step ks + 1 goes to its stage,
                    if (ks + 1 < ks_end) {                      // the loads of step ks + 3 are issued, one barrier, then its fragments
                        lstore(set ^ 1, nxt);
                        if (ks + 3 < ks_end) gload(set ^ 1, ks + 3);
                    }
                    __syncthreads();
                    if (ks + 1 < ks_end) xfrags(set ^ 1, nxt);
                }
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mi][kA[q]], fb[set][ni][kB[q]], acc[mi][ni], 0, 0, 0);
            }
            cur = nxt;
        };
        for (unsigned ks = ks_begin;;) {
            kstep(Stage<0>{}, ks);
            if (++ks >= ks_end) break;
            kstep(Stage<1>{}, ks);
            if (++ks >= ks_end) break;
        }
        finish_tile<WM, WN>(acc, a, stat, s_bias, split, t, slice, jbase, mbase, m0, n0, r, h, tid, wave);
        __syncthreads();                                      // the next tile overwrites s_bias and stage 0
    }
}

template <int WM, int WN>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(2))) void conv1x1_sb_kernel(const C1Args a) {
    extern __shared__ __attribute__((aligned(16))) char sb_smem[];
    NoStat st;
    conv1x1_tiles_sb<WM, WN>(a, st, sb_smem);
}

template <int WM, int WN>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(2))) void conv1x1_sb_absmax_kernel(const C1Args a, unsigned int* __restrict__ max_bits) {
    extern __shared__ __attribute__((aligned(16))) char sb_smem[];
    MaxStat st;
    conv1x1_tiles_sb<WM, WN>(a, st, sb_smem);
    publish_max<kT>(st.m, max_bits);
}

template <int WM, int WN>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(2))) void conv1x1_sb_qd_kernel(const C1Args a, const QdStat qd) {
    extern __shared__ __attribute__((aligned(16))) char sb_smem[];
    QdStat st = qd;
    conv1x1_tiles_sb<WM, WN>(a, st, sb_smem);
}

template <int WM, int WN>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(2))) void conv1x1_sb_hist_kernel(
    const C1Args a, const float* __restrict__ interval, unsigned long long* __restrict__ hist_row, const int allow_fast) {
    extern __shared__ __attribute__((aligned(16))) char sb_smem[];
    __shared__ unsigned int s_bins[FQ_BINS + kWave];
    for (int b = threadIdx.x; b < FQ_BINS + kWave; b += kT) s_bins[b] = 0u;
    __syncthreads();
    const float iv = *interval;
    unsigned int* park = s_bins + FQ_BINS + (threadIdx.x & (kWave - 1));
    if (allow_fast && fast_quotient_ok(iv)) {
        HistStat<true> st{s_bins, park, iv, 1.0f / iv};
        conv1x1_tiles_sb<WM, WN>(a, st, sb_smem);
    } else {
        HistStat<false> st{s_bins, park, iv, 1.0f / iv};
        conv1x1_tiles_sb<WM, WN>(a, st, sb_smem);
    }
    hist_flush<kT>(s_bins, hist_row);
}

template <int WM, int WN>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(2))) void conv1x1_sb_add_absmax_kernel(
    const C1Args a, unsigned int* __restrict__ max_y_bits, unsigned int* __restrict__ max_sum_bits) {
    extern __shared__ __attribute__((aligned(16))) char sb_smem[];
    AddStat<MaxStat, MaxStat, true> st;
    conv1x1_tiles_sb<WM, WN>(a, st, sb_smem);
    publish_max<kT>(st.c.m, max_y_bits);
    __syncthreads();
    publish_max<kT>(st.s.m, max_sum_bits);
}

template <int WM, int WN>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(2))) void conv1x1_sb_add_hist_kernel(
    const C1Args a, const float* __restrict__ interval_y, unsigned long long* __restrict__ hist_y,
    const float* __restrict__ interval_sum, unsigned long long* __restrict__ hist_sum, const int allow_fast) {
    extern __shared__ __attribute__((aligned(16))) char sb_smem[];
    __shared__ unsigned int s_bins[2][FQ_BINS + kWave];
    for (int b = threadIdx.x; b < 2 * (FQ_BINS + kWave); b += kT) (&s_bins[0][0])[b] = 0u;
    __syncthreads();
    const float ivy = *interval_y, ivs = *interval_sum;
    unsigned int* park0 = s_bins[0] + FQ_BINS + (threadIdx.x & (kWave - 1));
    unsigned int* park1 = s_bins[1] + FQ_BINS + (threadIdx.x & (kWave - 1));
    if (allow_fast && fast_quotient_ok(ivy) && fast_quotient_ok(ivs)) {
        AddStat<HistStat<true>, HistStat<true>, false> st{{s_bins[0], park0, ivy, 1.0f / ivy}, {s_bins[1], park1, ivs, 1.0f / ivs}};
        conv1x1_tiles_sb<WM, WN>(a, st, sb_smem);
    } else {
        AddStat<HistStat<false>, HistStat<false>, false> st{{s_bins[0], park0, ivy, 1.0f / ivy}, {s_bins[1], park1, ivs, 1.0f / ivs}};
        conv1x1_tiles_sb<WM, WN>(a, st, sb_smem);
    }
    hist_flush<kT>(s_bins[0], hist_y);
    hist_flush<kT>(s_bins[1], hist_sum);
}

// fp32 [Cout][Cin] -> the pack the kernel's A fragments are loaded from: [Cin / 16][plane: hi, mid, lo][k % 16 / 8][Cout][8] bf16
__global__ __launch_bounds__(256) void conv1x1_sb_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, unsigned Cin,
                                                              unsigned Cout) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= Cin * Cout) return;
    const unsigned co = i / Cin, k = i - co * Cin;
    unsigned p[3];
    split3(w[i], p[0], p[1], p[2]);
    for (unsigned pl = 0; pl < 3; ++pl)
        out[((((size_t)(k >> 4) * 3u + pl) * 2u + ((k >> 3) & 1u)) * Cout + co) * 8u + (k & 7u)] = (unsigned short)p[pl];
}

// (4 waves per SIMD = four workgroups per CU: <= 128 registers with the 64 accumulators; the K-tail form -- Cin not a multiple
//  of 16, no ResNet layer -- needs 130 and takes 3 waves instead of spilling)
template <int WM, int WN, int kTailK>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(kTailK == 1 ? 3 : 4))) void conv1x1_f32_kernel(const C1Args a) {
    __shared__ __attribute__((aligned(16))) float smem[Shape<WM, WN, step_of<WM>()>::kFloats];
    NoStat st;
    conv1x1_tiles<WM, WN, step_of<WM>(), kTailK>(a, st, smem);
}

template <int WM, int WN, int kTailK>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(kTailK == 1 ? 3 : 4))) void conv1x1_f32_absmax_kernel(const C1Args a, unsigned int* __restrict__ max_bits) {
    __shared__ __attribute__((aligned(16))) float smem[Shape<WM, WN, step_of<WM>()>::kFloats];
    MaxStat st;
    conv1x1_tiles<WM, WN, step_of<WM>(), kTailK>(a, st, smem);
    publish_max<kT>(st.m, max_bits);
}

// conv3 + Eltwise + ReLU of a bottleneck in one kernel (pass 1: both tensors' abs-max)
template <int WM, int WN>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(4))) void conv1x1_f32_add_absmax_kernel(
    const C1Args a, unsigned int* __restrict__ max_y_bits, unsigned int* __restrict__ max_sum_bits) {
    __shared__ __attribute__((aligned(16))) float smem[Shape<WM, WN, step_of<WM>()>::kFloats];
    AddStat<MaxStat, MaxStat, true> st;
    conv1x1_tiles<WM, WN, step_of<WM>(), 0>(a, st, smem);
    publish_max<kT>(st.c.m, max_y_bits);
    __syncthreads();                                          // (publish_max's LDS slots are about to be reused)
    publish_max<kT>(st.s.m, max_sum_bits);
}

// ... and in pass 2: both tensors histogrammed on their way through the registers, neither written (nothing is kept in pass 2);
// two 8 KB sets of LDS bins on top of the stages, a persistent grid for the flushes as in conv1x1_f32_hist_kernel
template <int WM, int WN>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(3))) void conv1x1_f32_add_hist_kernel(
    const C1Args a, const float* __restrict__ interval_y, unsigned long long* __restrict__ hist_y,
    const float* __restrict__ interval_sum, unsigned long long* __restrict__ hist_sum, const int allow_fast) {
    __shared__ __attribute__((aligned(16))) float smem[Shape<WM, WN, step_of<WM>()>::kFloats];
    __shared__ unsigned int s_bins[2][FQ_BINS + kWave];
    for (int b = threadIdx.x; b < 2 * (FQ_BINS + kWave); b += kT) (&s_bins[0][0])[b] = 0u;
    __syncthreads();
    const float ivy = *interval_y, ivs = *interval_sum;
    unsigned int* park0 = s_bins[0] + FQ_BINS + (threadIdx.x & (kWave - 1));
    unsigned int* park1 = s_bins[1] + FQ_BINS + (threadIdx.x & (kWave - 1));
    if (allow_fast && fast_quotient_ok(ivy) && fast_quotient_ok(ivs)) {
        AddStat<HistStat<true>, HistStat<true>, false> st{{s_bins[0], park0, ivy, 1.0f / ivy}, {s_bins[1], park1, ivs, 1.0f / ivs}};
        conv1x1_tiles<WM, WN, step_of<WM>(), 0>(a, st, smem);
    } else {
        AddStat<HistStat<false>, HistStat<false>, false> st{{s_bins[0], park0, ivy, 1.0f / ivy}, {s_bins[1], park1, ivs, 1.0f / ivs}};
        conv1x1_tiles<WM, WN, step_of<WM>(), 0>(a, st, smem);
    }
    hist_flush<kT>(s_bins[0], hist_y);
    hist_flush<kT>(s_bins[1], hist_sum);
}

// TestConv / TestLinear's forward in one kernel: convolution + bias, then QuanDequan on the accumulator's way out
template <int WM, int WN, int kTailK>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(kTailK == 1 ? 3 : 4))) void conv1x1_f32_qd_kernel(const C1Args a, const QdStat qd) {
    __shared__ __attribute__((aligned(16))) float smem[Shape<WM, WN, step_of<WM>()>::kFloats];
    QdStat st = qd;
    conv1x1_tiles<WM, WN, step_of<WM>(), kTailK>(a, st, smem);
}

// (the histogram form carries 8 KB of LDS bins on top of the stages; the persistent grid is what the occupancy query says)
template <int WM, int WN, int BK, int kTailK>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(3))) void conv1x1_f32_hist_kernel(
    const C1Args a, const float* __restrict__ interval, unsigned long long* __restrict__ hist_row, const int allow_fast) {
    __shared__ __attribute__((aligned(16))) float smem[Shape<WM, WN, BK>::kFloats];
    __shared__ unsigned int s_bins[FQ_BINS + kWave];
    for (int b = threadIdx.x; b < FQ_BINS + kWave; b += kT) s_bins[b] = 0u;
    __syncthreads();
    const float iv = *interval;
    unsigned int* park = s_bins + FQ_BINS + (threadIdx.x & (kWave - 1));
    if (allow_fast && fast_quotient_ok(iv)) {
        HistStat<true> st{s_bins, park, iv, 1.0f / iv};
        conv1x1_tiles<WM, WN, BK, kTailK>(a, st, smem);
    } else {
        HistStat<false> st{s_bins, park, iv, 1.0f / iv};
        conv1x1_tiles<WM, WN, BK, kTailK>(a, st, smem);
    }
    hist_flush<kT>(s_bins, hist_row);
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}

// The tail split.  A launch of T tiles on 256 CUs takes ceil(T / 256) tile times on the busiest CU: 784 tiles (3.06 per CU) cost
// what 1 024 would (scripts/_dbg/conv_balance_probe.py: 512 -> 512 3x3 @7x7 takes 449 us at 250 images = 3.00 tiles per CU and
// 568 us at 252 = 3.03).  So the T mod 256 tiles of the last, partly filled round are cut along K into 256 / (T mod 256)
// slices each: the tail becomes one more FULL round of short workgroups, 3.06 tile times instead of 4.  The slices meet in
// a workspace (conv1x1_tiles); the summation order is fixed, and which tiles are split is a function of the layer's shape
// alone, the same for every form of the kernel -- so a value does not depend on the statistic that rides on it.
constexpr unsigned kSplitMaxItems = 256, kSplitMaxSlices = 16;
// The workspace belongs to the CALLER (include/fq.h: fq_conv_f32_workspace_bytes): [kSplitMaxItems partial tiles of 128 x 128
// floats -- (split tiles) x (slices per tile) never exceeds 256][kSplitMaxItems arrival counters, one per split tile].  The
// counters must be zero when a launch starts; the workgroup that arrives last at a tile puts its counter back to zero, so a
// workspace zero-filled once stays usable launch after launch on one stream.  NULL (or too small): the launch runs unsplit.
constexpr size_t kSplitWsFloatBytes = (size_t)kSplitMaxItems * 128 * 128 * sizeof(float);
constexpr size_t kSplitWsBytes = kSplitWsFloatBytes + kSplitMaxItems * sizeof(unsigned);

void plan_split(C1Args& a, unsigned nk, void* workspace, size_t workspace_bytes) {
    a.split_first = a.tiles; a.split_s = 1; a.work = a.tiles; a.ws = nullptr; a.ws_count = nullptr;
    static const int on = env_int("FQ_CONV_TAIL_SPLIT", 1);
    if (!workspace || workspace_bytes < kSplitWsBytes || (reinterpret_cast<uintptr_t>(workspace) & 15u)) return;
    const unsigned rounds = a.tiles / (unsigned)kCUs, rem = a.tiles % (unsigned)kCUs;
    if (!on || rounds == 0 || rounds >= 16 || rem == 0 || rem > (unsigned)kCUs / 2) return;
    unsigned s = (unsigned)kCUs / rem;
    if (s > kSplitMaxSlices) s = kSplitMaxSlices;
    if (s > nk / 8) s = nk / 8;                               // at least 8 K steps per slice
    if (s < 2) return;
    a.split_first = a.tiles - rem; a.split_s = s; a.work = a.split_first + rem * s;
    a.ws = static_cast<float*>(workspace);
    a.ws_count = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + kSplitWsFloatBytes);
}

template <int WM, int WN, int kTailK>
void launch(C1Args a, unsigned cols, float* max_inout, const float* interval, int64_t* hist_row, int hist_per_cu, int fast,
            const QdStat* qd, void* workspace, size_t workspace_bytes, hipStream_t st) {
    typedef Shape<WM, WN> S;
    a.tiles_m = (a.Cout + S::BM - 1) / S::BM;
    a.tiles = ((cols + S::BN - 1) / S::BN) * a.tiles_m;
    constexpr unsigned BK = (unsigned)step_of<WM>();
    plan_split(a, kTailK == 2 ? (unsigned)(a.R * a.S) * (a.Cin / BK) : (a.Cin + BK - 1) / BK, workspace, workspace_bytes);
    if (qd) {
        hipLaunchKernelGGL((conv1x1_f32_qd_kernel<WM, WN, kTailK>), dim3(a.work), dim3(kT), 0, st, a, *qd);
    } else if (hist_row) {
        // every workgroup flushes up to 2048 bins with 64-bit atomics at its end: a persistent grid of exactly the
        // workgroups the chip holds at once (LDS: three stages + 8 KB of bins), each taking every grid-th tile
        static const int resident = [] {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv1x1_f32_hist_kernel<WM, WN, step_of<WM>(), kTailK>, kT, 0) != hipSuccess || n < 1) n = 1;
            return n;
        }();
        unsigned grid = (unsigned)kCUs * (unsigned)(hist_per_cu > 0 ? hist_per_cu : resident);
        if (grid > a.work) grid = a.work;
        hipLaunchKernelGGL((conv1x1_f32_hist_kernel<WM, WN, step_of<WM>(), kTailK>), dim3(grid), dim3(kT), 0, st, a, interval,
                           reinterpret_cast<unsigned long long*>(hist_row), fast);
    } else if (max_inout) {
        hipLaunchKernelGGL((conv1x1_f32_absmax_kernel<WM, WN, kTailK>), dim3(a.work), dim3(kT), 0, st, a,
                           reinterpret_cast<unsigned int*>(max_inout));
    } else {
        hipLaunchKernelGGL((conv1x1_f32_kernel<WM, WN, kTailK>), dim3(a.work), dim3(kT), 0, st, a);
    }
}

// the split-bf16 form: same tiling, tail split and statistics; LDS is dynamic (74 KB for the 128 x 128 tile: two workgroups per CU)
template <int WM, int WN>
int launch_sb(C1Args a, float* max_inout, const float* interval, int64_t* hist_row, int fast, const QdStat* qd, const float* res_interval_y,
              int64_t* hist_y, const float* interval_sum, int64_t* hist_sum, float* max_y, float* max_sum, bool add, void* workspace,
              size_t workspace_bytes, hipStream_t st) {
    typedef ShapeSb<WM, WN> S;
    a.tiles_m = (a.Cout + S::BM - 1) / S::BM;
    a.tiles = ((a.cols + S::BN - 1) / S::BN) * a.tiles_m;
    plan_split(a, a.Cin / (unsigned)kSbBK, workspace, workspace_bytes);
    constexpr int lds = S::kBytes;
    static bool d_plain[kMaxDevices], d_max[kMaxDevices], d_hist[kMaxDevices], d_qd[kMaxDevices], d_am[kMaxDevices], d_ah[kMaxDevices];
    auto resident = [&](const void* k, int static_lds) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, kT, (size_t)lds) != hipSuccess || n < 1) n = 1;
        (void)static_lds;
        return (unsigned)n;
    };
    if (add) {
        if (hist_y) {
            const void* k = reinterpret_cast<const void*>(conv1x1_sb_add_hist_kernel<WM, WN>);
            if (!ensure_dynamic_lds(k, lds, d_ah)) return FQ_ERR_HIP;
            unsigned grid = (unsigned)kCUs * resident(k, 0);
            if (grid > a.work) grid = a.work;
            hipLaunchKernelGGL((conv1x1_sb_add_hist_kernel<WM, WN>), dim3(grid), dim3(kT), lds, st, a, res_interval_y,
                               reinterpret_cast<unsigned long long*>(hist_y), interval_sum, reinterpret_cast<unsigned long long*>(hist_sum), fast);
        } else {
            if (!ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_sb_add_absmax_kernel<WM, WN>), lds, d_am)) return FQ_ERR_HIP;
            hipLaunchKernelGGL((conv1x1_sb_add_absmax_kernel<WM, WN>), dim3(a.work), dim3(kT), lds, st, a,
                               reinterpret_cast<unsigned int*>(max_y), reinterpret_cast<unsigned int*>(max_sum));
        }
    } else if (qd) {
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_sb_qd_kernel<WM, WN>), lds, d_qd)) return FQ_ERR_HIP;
        hipLaunchKernelGGL((conv1x1_sb_qd_kernel<WM, WN>), dim3(a.work), dim3(kT), lds, st, a, *qd);
    } else if (hist_row) {
        const void* k = reinterpret_cast<const void*>(conv1x1_sb_hist_kernel<WM, WN>);
        if (!ensure_dynamic_lds(k, lds, d_hist)) return FQ_ERR_HIP;
        unsigned grid = (unsigned)kCUs * resident(k, 0);
        if (grid > a.work) grid = a.work;
        hipLaunchKernelGGL((conv1x1_sb_hist_kernel<WM, WN>), dim3(grid), dim3(kT), lds, st, a, interval,
                           reinterpret_cast<unsigned long long*>(hist_row), fast);
    } else if (max_inout) {
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_sb_absmax_kernel<WM, WN>), lds, d_max)) return FQ_ERR_HIP;
        hipLaunchKernelGGL((conv1x1_sb_absmax_kernel<WM, WN>), dim3(a.work), dim3(kT), lds, st, a, reinterpret_cast<unsigned int*>(max_inout));
    } else {
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_sb_kernel<WM, WN>), lds, d_plain)) return FQ_ERR_HIP;
        hipLaunchKernelGGL((conv1x1_sb_kernel<WM, WN>), dim3(a.work), dim3(kT), lds, st, a);
    }
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

}  // namespace
}  // namespace fq

using namespace fq;

namespace {

// the common host side of fq_conv1x1_f32 (R = S = 1, pad = 0) and fq_conv_kxk_f32
int conv_f32_launch(const float* x, const float* wt, const float* bias, float* y, float* relu_out, int N, int Cin, int Hin, int Win,
                    int Cout, int R, int S, int stride, int pad, float* max_inout, const float* interval, int64_t* hist_row,
                    void* workspace, size_t workspace_bytes, fq_stream_t stream, const QdStat* qd = nullptr,
                    const unsigned short* wsb = nullptr) {
    if (N < 0 || Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0 || stride < 1 || R < 1 || S < 1 || pad < 0) return FQ_ERR_INVALID_ARG;
    if (wsb) wt = reinterpret_cast<const float*>(wsb);                                        // (the checks below: non-null, 16-byte aligned)
    if (Hin + 2 * pad < R || Win + 2 * pad < S) return FQ_ERR_INVALID_ARG;
    if (max_inout && hist_row) return FQ_ERR_INVALID_ARG;
    if (hist_row && !interval) return FQ_ERR_INVALID_ARG;
    if (N == 0) return FQ_OK;
    if (!x || !wt || (!y && (!relu_out || qd))) return FQ_ERR_INVALID_ARG;                   // (y may be null when only its ReLU is wanted)
    if ((Cout & 3) || (reinterpret_cast<uintptr_t>(wt) & 15u)) return FQ_ERR_UNSUPPORTED;       // float4 loads of Wt rows
    const bool taps = R * S > 1 || pad > 0;
    if (taps && (Cin % 16) != 0) return FQ_ERR_UNSUPPORTED;                                   // whole K steps per tap
    if (wsb && (taps || (Cin % kSbBK) != 0)) return FQ_ERR_UNSUPPORTED;                        // the split-bf16 form: 1x1, whole K steps of 16
    const int Hout = (Hin + 2 * pad - R) / stride + 1, Wout = (Win + 2 * pad - S) / stride + 1;
    const size_t cols = (size_t)N * Hout * Wout;
    const size_t in_elems = (size_t)N * Cin * Hin * Win, out_elems = cols * Cout, w_elems = (size_t)R * S * Cin * Cout;
    // 32-bit BYTE offsets into x, Wt and y
    if (cols >= 0xffffff00ULL || in_elems >= (1ULL << 30) || w_elems >= (1ULL << 30) || out_elems >= (1ULL << 30))
        return FQ_ERR_UNSUPPORTED;
    C1Args a;
    a.x = x; a.wt = wt; a.wsb = wsb; a.bias = bias; a.y = y; a.relu = relu_out; a.res = nullptr; a.sum = nullptr; a.store_y = 1;
    a.Cin = (unsigned)Cin; a.Cout = (unsigned)Cout; a.HWin = (unsigned)(Hin * Win); a.HWout = (unsigned)(Hout * Wout);
    a.Win = (unsigned)Win; a.Wout = (unsigned)Wout; a.stride = (unsigned)stride;
    a.Hin = Hin; a.R = R; a.S = S; a.pad = pad;
    a.cols = (unsigned)cols;
    a.x_bytes = (unsigned)(in_elems * 4);
    a.w_bytes = wsb ? (unsigned)(w_elems * 6) : (unsigned)(w_elems * 4);
    a.y_bytes = (unsigned)(out_elems * 4);
    a.tiles_m = a.tiles = 0;
    a.split_first = a.split_s = a.work = 0; a.ws = nullptr; a.ws_count = nullptr;
    a.stream_stores = out_elems * (relu_out && y ? 8 : 4) > ((size_t)256 << 20);  // beyond the Infinity Cache
#ifdef FQ_C1_ABLATE
    a.ablate = env_int("FQ_C1_ABLATE", 0);
#endif
    hipStream_t st = as_stream(stream);
    static const int hist_per_cu = env_int("FQ_CONV1X1_HIST_WG_PER_CU", 0);   // 0: what the occupancy query says
    static const int fast = [] { const char* e = getenv("FQ_HIST_IEEE_DIV"); return (e && e[0] && e[0] != '0') ? 0 : 1; }();
    const int mode = taps ? 2 : ((Cin % kBK) != 0 ? 1 : 0);
    // tile shape: 128 x 128; 64 x 128 for the 64-channel layers (a 128-row tile would be half empty) and for launches whose
    // 128 x 128 tiles would not even fill the 1 024 resident slots once (1024 -> 256 @14x14 at 256 images: 784 tiles leave
    // a quarter of the CUs with 4 tiles and the rest with 3; 1 568 half-size tiles balance better: 0.257 -> 0.230 ms).
    // FQ_CONV1X1_SHAPE = 22 | 12 forces one (probing)
    static const int forced = env_int("FQ_CONV1X1_SHAPE", 0);
    const size_t tiles22 = ((cols + 127) / 128) * (size_t)((Cout + 127) / 128);
    const int shape = forced ? forced : ((Cout <= 64 || tiles22 <= (size_t)kCUs * 4) ? 12 : 22);
    if (wsb) {
        if (shape == 12) return launch_sb<1, 2>(a, max_inout, interval, hist_row, fast, qd, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, workspace, workspace_bytes, st);
        return launch_sb<2, 2>(a, max_inout, interval, hist_row, fast, qd, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, workspace, workspace_bytes, st);
    }
#define FQ_C1_LAUNCH(WM, WN)                                                                                       \
    do {                                                                                                           \
        if (mode == 2) launch<WM, WN, 2>(a, a.cols, max_inout, interval, hist_row, hist_per_cu, fast, qd, workspace, workspace_bytes, st);     \
        else if (mode == 1) launch<WM, WN, 1>(a, a.cols, max_inout, interval, hist_row, hist_per_cu, fast, qd, workspace, workspace_bytes, st); \
        else launch<WM, WN, 0>(a, a.cols, max_inout, interval, hist_row, hist_per_cu, fast, qd, workspace, workspace_bytes, st);               \
    } while (0)
    if (shape == 12) FQ_C1_LAUNCH(1, 2);
    else FQ_C1_LAUNCH(2, 2);
#undef FQ_C1_LAUNCH
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

}  // namespace

extern "C" size_t fq_conv_f32_workspace_bytes(void) { return kSplitWsBytes; }

extern "C" int fq_conv1x1_f32(const float* x, const float* wt, const float* bias, float* y, float* relu_out, int N, int Cin,
                              int Hin, int Win, int Cout, int stride, float* max_inout, const float* interval,
                              int64_t* hist_row, void* workspace, size_t workspace_bytes, fq_stream_t stream) {
    return conv_f32_launch(x, wt, bias, y, relu_out, N, Cin, Hin, Win, Cout, 1, 1, stride, 0, max_inout, interval, hist_row,
                           workspace, workspace_bytes, stream);
}

extern "C" int fq_conv_kxk_f32(const float* x, const float* wt, const float* bias, float* y, float* relu_out, int N, int Cin,
                               int Hin, int Win, int Cout, int R, int S, int stride, int pad, float* max_inout,
                               const float* interval, int64_t* hist_row, void* workspace, size_t workspace_bytes,
                               fq_stream_t stream) {
    return conv_f32_launch(x, wt, bias, y, relu_out, N, Cin, Hin, Win, Cout, R, S, stride, pad, max_inout, interval, hist_row,
                           workspace, workspace_bytes, stream);
}

// The last 1x1 convolution of a residual block together with the Eltwise (fabu_layer.py:5-11) and the ReLU behind it:
// y = conv(x) + bias (abs-max -> *max_y; written to y unless y is null), sum = y + res (abs-max -> *max_sum; written unless
// sum is null), relu_out = max(sum, 0).  Bit for bit what fq_conv1x1_f32 followed by fq_add_absmax_f32 leave.
static int conv_add_launch(const float* x, const float* wt_in, const unsigned short* wsb, const float* bias, const float* res, float* y, float* sum,
                           float* relu_out, int N, int Cin, int Hin, int Win, int Cout, int stride, float* max_y, float* max_sum,
                           const float* interval_y, int64_t* hist_y, const float* interval_sum, int64_t* hist_sum,
                           void* workspace, size_t workspace_bytes, fq_stream_t stream) {
    const float* wt = wsb ? reinterpret_cast<const float*>(wsb) : wt_in;
    if (N < 0 || Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0 || stride < 1) return FQ_ERR_INVALID_ARG;
    if (N == 0) return FQ_OK;
    const bool hist = hist_y != nullptr;
    if (!x || !wt || !res || !relu_out) return FQ_ERR_INVALID_ARG;
    if (hist ? (!hist_sum || !interval_y || !interval_sum || hist_y == hist_sum) : (!max_y || !max_sum)) return FQ_ERR_INVALID_ARG;
    if ((Cin % kBK) != 0 || (Cout % 128) != 0 || (reinterpret_cast<uintptr_t>(wt) & 15u)) return FQ_ERR_UNSUPPORTED;
    const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;
    const size_t cols = (size_t)N * Hout * Wout;
    const size_t in_elems = (size_t)N * Cin * Hin * Win, out_elems = cols * Cout, w_elems = (size_t)Cin * Cout;
    if (cols >= 0xffffff00ULL || in_elems >= (1ULL << 30) || w_elems >= (1ULL << 30) || out_elems >= (1ULL << 30))
        return FQ_ERR_UNSUPPORTED;
    C1Args a;
    a.x = x; a.wt = wt; a.wsb = wsb; a.bias = bias; a.y = y; a.relu = relu_out; a.res = res; a.sum = sum; a.store_y = y != nullptr;
    a.Cin = (unsigned)Cin; a.Cout = (unsigned)Cout; a.HWin = (unsigned)(Hin * Win); a.HWout = (unsigned)(Hout * Wout);
    a.Win = (unsigned)Win; a.Wout = (unsigned)Wout; a.stride = (unsigned)stride;
    a.Hin = Hin; a.R = 1; a.S = 1; a.pad = 0;
    a.cols = (unsigned)cols;
    a.x_bytes = (unsigned)(in_elems * 4); a.w_bytes = wsb ? (unsigned)(w_elems * 6) : (unsigned)(w_elems * 4); a.y_bytes = (unsigned)(out_elems * 4);
    // non-temporal accesses beyond the Infinity Cache -- but only where a plane is a whole number of 64-byte blocks: the runs of
    // 128 bytes a wave stores are then whole blocks, and anything else streamed past the L2 becomes partial writes
    // (scripts/conv_add_bench.py: 256 -> 1024 @14x14, both tensors kept, 356 us with default stores, 486 with nt)
    a.stream_stores = out_elems * (size_t)(8 + (y ? 4 : 0) + (sum ? 4 : 0)) > ((size_t)256 << 20) && (a.HWout % 16u) == 0;
    static const int stream_env = env_int("FQ_CONV_ADD_STREAM", -1);      // 0 / 1: force default / non-temporal accesses (A/B)
    if (stream_env >= 0) a.stream_stores = stream_env;
#ifdef FQ_C1_ABLATE
    a.ablate = 0;
#endif
    static const int forced = env_int("FQ_CONV1X1_SHAPE", 0);
    const size_t tiles22 = ((cols + 127) / 128) * (size_t)(Cout / 128);
    const bool narrow = forced ? forced == 12 : tiles22 <= (size_t)kCUs * 4;
    hipStream_t st = as_stream(stream);
    a.tiles_m = (unsigned)Cout / (narrow ? 64u : 128u);
    a.tiles = (unsigned)((cols + 127) / 128) * a.tiles_m;
    if (wsb) {
        static const int fast_sb = [] { const char* e = getenv("FQ_HIST_IEEE_DIV"); return (e && e[0] && e[0] != '0') ? 0 : 1; }();
        if (narrow) return launch_sb<1, 2>(a, nullptr, nullptr, nullptr, fast_sb, nullptr, interval_y, hist_y, interval_sum, hist_sum, max_y, max_sum, true, workspace, workspace_bytes, st);
        return launch_sb<2, 2>(a, nullptr, nullptr, nullptr, fast_sb, nullptr, interval_y, hist_y, interval_sum, hist_sum, max_y, max_sum, true, workspace, workspace_bytes, st);
    }
    plan_split(a, (unsigned)Cin / (unsigned)(narrow ? step_of<1>() : step_of<2>()), workspace, workspace_bytes);
    if (hist) {
        static const int fast = [] { const char* e = getenv("FQ_HIST_IEEE_DIV"); return (e && e[0] && e[0] != '0') ? 0 : 1; }();
        static const int res12 = [] {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv1x1_f32_add_hist_kernel<1, 2>, kT, 0) != hipSuccess || n < 1) n = 1;
            return n;
        }();
        static const int res22 = [] {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv1x1_f32_add_hist_kernel<2, 2>, kT, 0) != hipSuccess || n < 1) n = 1;
            return n;
        }();
        unsigned grid = (unsigned)kCUs * (unsigned)(narrow ? res12 : res22);
        if (grid > a.work) grid = a.work;
        unsigned long long* hy = reinterpret_cast<unsigned long long*>(hist_y);
        unsigned long long* hs = reinterpret_cast<unsigned long long*>(hist_sum);
        if (narrow)
            hipLaunchKernelGGL((conv1x1_f32_add_hist_kernel<1, 2>), dim3(grid), dim3(kT), 0, st, a, interval_y, hy, interval_sum, hs, fast);
        else
            hipLaunchKernelGGL((conv1x1_f32_add_hist_kernel<2, 2>), dim3(grid), dim3(kT), 0, st, a, interval_y, hy, interval_sum, hs, fast);
    } else if (narrow) {
        hipLaunchKernelGGL((conv1x1_f32_add_absmax_kernel<1, 2>), dim3(a.work), dim3(kT), 0, st, a,
                           reinterpret_cast<unsigned int*>(max_y), reinterpret_cast<unsigned int*>(max_sum));
    } else {
        hipLaunchKernelGGL((conv1x1_f32_add_absmax_kernel<2, 2>), dim3(a.work), dim3(kT), 0, st, a,
                           reinterpret_cast<unsigned int*>(max_y), reinterpret_cast<unsigned int*>(max_sum));
    }
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_conv1x1_add_f32(const float* x, const float* wt, const float* bias, const float* res, float* y, float* sum,
                                  float* relu_out, int N, int Cin, int Hin, int Win, int Cout, int stride, float* max_y,
                                  float* max_sum, void* workspace, size_t workspace_bytes, fq_stream_t stream) {
    return conv_add_launch(x, wt, nullptr, bias, res, y, sum, relu_out, N, Cin, Hin, Win, Cout, stride, max_y, max_sum, nullptr, nullptr,
                           nullptr, nullptr, workspace, workspace_bytes, stream);
}

// The same chain in calibration pass 2: the convolution's output and the sum are histogrammed (2048 bins each, rows hist_y and
// hist_sum with their interval widths) while they pass through the registers and are not written at all; relu_out = max(sum, 0).
extern "C" int fq_conv1x1_add_hist_f32(const float* x, const float* wt, const float* bias, const float* res, float* relu_out,
                                       int N, int Cin, int Hin, int Win, int Cout, int stride, const float* interval_y,
                                       int64_t* hist_y, const float* interval_sum, int64_t* hist_sum, void* workspace,
                                       size_t workspace_bytes, fq_stream_t stream) {
    if (!hist_y) return FQ_ERR_INVALID_ARG;
    return conv_add_launch(x, wt, nullptr, bias, res, nullptr, nullptr, relu_out, N, Cin, Hin, Win, Cout, stride, nullptr, nullptr, interval_y,
                           hist_y, interval_sum, hist_sum, workspace, workspace_bytes, stream);
}

// TestConv.forward (new_quantity_op.py:283-292) / TestLinear.forward (:248-256 on the classifier seen as a 1x1 layer) in one
// kernel: y = QuanDequan(conv(x) + bias, bit).  The value QuanDequan sees is the kernel's own fp32 sum -- the same one
// fq_conv1x1_f32 / fq_conv_kxk_f32 would have stored -- so the result equals fq_quandequan_f32 of their output bit for bit.
static bool qd_params(int bit, int bitwidth, QdStat* qd) {
    if ((bitwidth != 8 && bitwidth != 16) || bit < -120 || bit > 120) return false;
    qd->scale = ldexpf(1.0f, bit); qd->inv = ldexpf(1.0f, -bit);
    qd->lo = bitwidth == 8 ? -128.0f : -32768.0f; qd->hi = bitwidth == 8 ? 127.0f : 32767.0f;
    return true;
}

extern "C" int fq_conv1x1_qd_f32(const float* x, const float* wt, const float* bias, float* y, int N, int Cin, int Hin, int Win,
                                 int Cout, int stride, int bit, int bitwidth, void* workspace, size_t workspace_bytes,
                                 fq_stream_t stream) {
    QdStat qd;
    if (!qd_params(bit, bitwidth, &qd)) return FQ_ERR_INVALID_ARG;
    return conv_f32_launch(x, wt, bias, y, nullptr, N, Cin, Hin, Win, Cout, 1, 1, stride, 0, nullptr, nullptr, nullptr, workspace,
                           workspace_bytes, stream, &qd);
}

extern "C" int fq_conv_kxk_qd_f32(const float* x, const float* wt, const float* bias, float* y, int N, int Cin, int Hin, int Win,
                                  int Cout, int R, int S, int stride, int pad, int bit, int bitwidth, void* workspace,
                                  size_t workspace_bytes, fq_stream_t stream) {
    QdStat qd;
    if (!qd_params(bit, bitwidth, &qd)) return FQ_ERR_INVALID_ARG;
    return conv_f32_launch(x, wt, bias, y, nullptr, N, Cin, Hin, Win, Cout, R, S, stride, pad, nullptr, nullptr, nullptr, workspace,
                           workspace_bytes, stream, &qd);
}

// ---- the split-bf16 form of the 1x1 convolutions (conv1x1_tiles_sb): same contracts, weights packed by fq_conv1x1_sb_pack
extern "C" size_t fq_conv1x1_sb_packed_bytes(int Cin, int Cout) { return (Cin > 0 && Cout > 0) ? (size_t)6 * (size_t)Cin * (size_t)Cout : 0; }

extern "C" int fq_conv1x1_sb_supported(int Cin, int Cout) { return Cin > 0 && Cout > 0 && (Cin % kSbBK) == 0 && (Cout & 3) == 0; }

extern "C" int fq_conv1x1_sb_pack(const float* w_kc, void* wsb, int Cin, int Cout, fq_stream_t stream) {
    if (!w_kc || !wsb || Cin <= 0 || Cout <= 0) return FQ_ERR_INVALID_ARG;
    if ((size_t)Cin * Cout >= (1ULL << 30)) return FQ_ERR_UNSUPPORTED;
    const unsigned n = (unsigned)Cin * (unsigned)Cout;
    hipLaunchKernelGGL(conv1x1_sb_pack_kernel, dim3((n + 255u) / 256u), dim3(256), 0, as_stream(stream), w_kc, static_cast<unsigned short*>(wsb),
                       (unsigned)Cin, (unsigned)Cout);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}

extern "C" int fq_conv1x1_sb_f32(const float* x, const void* wsb, const float* bias, float* y, float* relu_out, int N, int Cin,
                                 int Hin, int Win, int Cout, int stride, float* max_inout, const float* interval,
                                 int64_t* hist_row, void* workspace, size_t workspace_bytes, fq_stream_t stream) {
    if (!wsb) return FQ_ERR_INVALID_ARG;
    return conv_f32_launch(x, nullptr, bias, y, relu_out, N, Cin, Hin, Win, Cout, 1, 1, stride, 0, max_inout, interval, hist_row,
                           workspace, workspace_bytes, stream, nullptr, static_cast<const unsigned short*>(wsb));
}

extern "C" int fq_conv1x1_sb_qd_f32(const float* x, const void* wsb, const float* bias, float* y, int N, int Cin, int Hin, int Win,
                                    int Cout, int stride, int bit, int bitwidth, void* workspace, size_t workspace_bytes,
                                    fq_stream_t stream) {
    QdStat qd;
    if (!wsb || !qd_params(bit, bitwidth, &qd)) return FQ_ERR_INVALID_ARG;
    return conv_f32_launch(x, nullptr, bias, y, nullptr, N, Cin, Hin, Win, Cout, 1, 1, stride, 0, nullptr, nullptr, nullptr, workspace,
                           workspace_bytes, stream, &qd, static_cast<const unsigned short*>(wsb));
}

extern "C" int fq_conv1x1_sb_add_f32(const float* x, const void* wsb, const float* bias, const float* res, float* y, float* sum,
                                     float* relu_out, int N, int Cin, int Hin, int Win, int Cout, int stride, float* max_y,
                                     float* max_sum, void* workspace, size_t workspace_bytes, fq_stream_t stream) {
    if (!wsb) return FQ_ERR_INVALID_ARG;
    return conv_add_launch(x, nullptr, static_cast<const unsigned short*>(wsb), bias, res, y, sum, relu_out, N, Cin, Hin, Win, Cout, stride,
                           max_y, max_sum, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int fq_conv1x1_sb_add_hist_f32(const float* x, const void* wsb, const float* bias, const float* res, float* relu_out, int N,
                                          int Cin, int Hin, int Win, int Cout, int stride, const float* interval_y, int64_t* hist_y,
                                          const float* interval_sum, int64_t* hist_sum, void* workspace, size_t workspace_bytes,
                                          fq_stream_t stream) {
    if (!wsb || !hist_y) return FQ_ERR_INVALID_ARG;
    return conv_add_launch(x, nullptr, static_cast<const unsigned short*>(wsb), bias, res, nullptr, nullptr, relu_out, N, Cin, Hin, Win, Cout,
                           stride, nullptr, nullptr, interval_y, hist_y, interval_sum, hist_sum, workspace, workspace_bytes, stream);
}
