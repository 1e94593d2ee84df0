/* fq.h -- C ABI of libfq_hip.so: the MI355X (gfx950) implementation of the pytorch-quantity hot path.
 *
 * The reference (lswzjuer/pytorch-quantity) is pure Python and has no FFI; this header is the
 * boundary a maintainer would bind from the reference's Python modules with ctypes (the binding is
 * shown in INTEGRATION.md and is what pytorch-quantity_amd/quantity/common/quantity/_native.py does).
 * Every entry point names the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - Every function returns 0 (FQ_OK) or a negative fq_status; nothing throws.
 *   - Pointers are DEVICE pointers (hipMalloc'ed / torch.cuda tensors' data_ptr()) unless the
 *     parameter is documented "host".  The caller owns every buffer.
 *   - All device work is enqueued on the passed hipStream_t (as void*; NULL = the null stream) and is
 *     asynchronous; there is no hidden global state, so calls are thread-safe per stream and
 *     capturable into a hipGraph.  No function allocates device memory: scratch a kernel needs is a `workspace`
 *     argument sized by a fq_*_workspace_bytes() query (fq_kl_threshold, the float convolutions), owned by the caller.
 *   - "row" = one histogram row.  Parity mode uses one row per hooked tensor (the reference's
 *     calibrator is per tensor: distribution_collector.py:40-42); row = tensor x channel is the
 *     same kernels with more rows.
 */
#ifndef FQ_H
#define FQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQ_VERSION 103            /* 0.1.3: fq_block_tail_proj_i8; owner flush of per-channel histogram rows */
#define FQ_BINS 2048              /* INTERVAL_NUM, tools/configs.yml:24 */
#define FQ_KL_TARGET_BINS 128     /* quantizer.py:98 target_bin */
#define FQ_KL_CANDIDATES 1920     /* thresholds 128..2047, quantizer.py:103 */
#define FQ_MAX_SEGS 1024          /* per call */

typedef void* fq_stream_t;        /* hipStream_t */

typedef enum fq_status {
    FQ_OK = 0,
    FQ_ERR_INVALID_ARG = -1,
    FQ_ERR_HIP = -2,              /* a HIP runtime call failed; fq_last_hip_error() has the code */
    FQ_ERR_WORKSPACE = -3,        /* workspace too small */
    FQ_ERR_UNSUPPORTED = -4
} fq_status;

/* One contiguous run of fp32 values that is accumulated into histogram row `row`.
 * A tensor is one segment; several segments may share a row (multi-batch, merged groups). */
typedef struct fq_seg {
    const float* ptr;             /* device, 4-byte aligned (16-byte alignment is faster, not required) */
    uint64_t n;                   /* elements; 0 allowed */
    int32_t row;                  /* >= 0 */
    int32_t reserved;             /* must be 0 */
} fq_seg;

int fq_version(void);
const char* fq_status_string(int status);
int fq_last_hip_error(void);      /* thread-local hipError_t of the last FQ_ERR_HIP */
/* Diagnostic, thread-local like the error code: which kernel the last fq_conv2d_i8 / _resident / _add_resident / _stem call of
 * this thread launched -- low byte: 1 conv3x3_i8_c64 (stationary 64-channel 3x3), 2 conv1x1_i8_stream, 3 conv3x3_i8_halo8
 * (256-pixel tiles, eight waves), 4 conv3x3_i8_halo, 5 / 6 conv2d_i8_dma with a ring of 2 / 3, 7 / 8 / 9 conv2d_i8_kernel
 * (C % 128 / C = 64 / general path), 10 stem_conv_i8, 11 block_tail_i8 (fq_block_tail_i8), 12 the same with the projection
 * shortcut computed in the kernel (fq_block_tail_proj_i8), 13 linear_i8_wave (a linear layer, one wave per 32 x 32 tile), 0 nothing
 * launched; bits 8-15: output-channel
 * tile (64 / 128).  Lets a test assert that the dispatch it checked against a golden is the dispatch a benchmark timed. */
int fq_conv2d_i8_last_variant(void);

/* ---- calibration: abs-max, 2048-bin histogram -------------------------------------------- */

/* distribution_collector.py:70-78 (refresh_max_val): for every segment,
 *   max_inout[row] = max(max_inout[row], max |x|).
 * max_inout is fp32[rows] on the device, zero-initialised by the caller before the first batch
 * (the reference's running max starts at 0).  NaNs are ignored.  `segs` is a HOST array. */
int fq_absmax_seg(const fq_seg* segs, int nseg, float* max_inout, fq_stream_t stream);

/* distribution_collector.py:127-135 (_add_to_distribution) + :115-118 (accumulate):
 *   for x != 0:  hist[row][ min( (int32) fl32(|x| / interval[row]), 2047 ) ] += 1
 * with a correctly rounded fp32 divide.  interval is fp32[rows] on the device (computed by the
 * caller exactly as distribution_collector.py:61, merged as pytorch_quantizer.py:396-411).
 * hist is int64[rows][2048] on the device, accumulated into (caller zeroes it once).
 * Deliberate deviation: the reference accumulates in int32 and wraps past 2^31-1 per bin; int64
 * never wraps.  Quotients >= 2048 (pass 2 seeing larger values than pass 1), +inf and NaN land in
 * bin 2047 (the reference raises IndexError for the last two). */
int fq_hist2048_seg(const fq_seg* segs, int nseg, const float* interval, int64_t* hist,
                    fq_stream_t stream);

/* ---- calibration: KL threshold sweep ------------------------------------------------------ */

size_t fq_kl_workspace_bytes(int rows);

/* quantizer.py:95-96 (normalize_distribution) + :98-167 (threshold_distribution) + :169-174
 * (compute_kl_divergence) for `rows` histograms at once:
 *   thr_out[r] = the threshold bin t* in [128, 2047] minimising KL(P[:t] || expand(quantize(P,t)))
 * in float64 with NumPy's pairwise summation order; first strict minimum wins.
 * hist: int64[rows][2048] device (merged groups are summed by the caller beforehand; the
 * reference's float64 merged form holds the same integers).  thr_out: int32[rows] device.
 * kl_curve_out: NULL or float64[rows][1920] device, receives KL(t) for t = 128..2047.
 * workspace: device scratch of fq_kl_workspace_bytes(rows). */
int fq_kl_threshold(const int64_t* hist, int rows, int32_t* thr_out, double* kl_curve_out,
                    void* workspace, size_t workspace_bytes, fq_stream_t stream);

/* The same search with its evidence and a choice of evaluation strategy.
 *   best_kl_out      NULL or float64[rows]: KL(t*) (+inf when no candidate is below the reference's 66666 start value)
 *   runner_up_kl_out NULL or float64[rows]: the smallest KL of any OTHER candidate.  runner_up - best is the margin of
 *                    the argmin: a margin of a few ulps is a near-tie that a differently rounded logarithm could decide
 *                    the other way (the reference's np.log is faithful, not correctly rounded).
 *   mode  FQ_KL_EXHAUSTIVE  all 1920 candidates of every row in the reference's float64 operation order (quantizer.py
 *                           :98-174 line by line);
 *         FQ_KL_SCREENED    every candidate first through a closed form of the same sum (one logarithm per quantised
 *                           bin instead of one per histogram bin; |S(t) - KL(t)| < FQ_KL_SCREEN_BOUND), then the exhaustive
 *                           evaluation of the candidates within FQ_KL_SCREEN_MARGIN of the smallest S(t) only.  The minimum is among
 *                           them, so thr_out / best_kl_out are identical to FQ_KL_EXHAUSTIVE; runner_up_kl_out and
 *                           kl_curve_out hold the closed-form value where the exact one was not needed;
 *         FQ_KL_AUTO        screened from 256 rows up when no curve is asked for (per-channel calibration: 42 667
 *                           rows), exhaustive below; the environment variable FQ_KL_EXHAUSTIVE=1 forces exhaustive. */
#define FQ_KL_AUTO 0
#define FQ_KL_EXHAUSTIVE 1
#define FQ_KL_SCREENED 2
/* The two numbers the screened search rests on (one definition; csrc/fq_kl.hip asserts 2 * bound < margin):
 *   FQ_KL_SCREEN_BOUND   |S(t) - KL(t)| of the closed form, proven in csrc/fq_kl.hip (measured: < 1e-13 on every golden
 *                        and fuzz histogram, tests/test_kl_screen_cpu.py);
 *   FQ_KL_SCREEN_MARGIN  candidates with S(t) <= min S + margin (+ 1e-12 |min S|) get the exact evaluation: 500 x bound. */
#define FQ_KL_SCREEN_BOUND 2e-13
#define FQ_KL_SCREEN_MARGIN 1e-10
int fq_kl_threshold_ex(const int64_t* hist, int rows, int32_t* thr_out, double* best_kl_out,
                       double* runner_up_kl_out, double* kl_curve_out, int mode,
                       void* workspace, size_t workspace_bytes, fq_stream_t stream);

/* A whole stage of residual blocks in one pass.  The blocks of a stage chain on their shortcut: S_1 = y_1 + head and, for k > 1,
 * S_k = y_k + max(S_(k-1), 0) -- an identity block's shortcut IS the previous block's nn.ReLU output.  For every chain, y_k[i] is
 * counted into row_y[k] (skipped when -1) and S_k[i] into row_sum[k], k = 0 .. len - 1, each exactly as fq_hist2048_seg would count
 * the stored tensor (every addition one rounded fp32 add as fabu_layer.py:5-11 performs it, every ReLU as torch computes it); no
 * sum and no intermediate shortcut is read or written: (len + 1) x 4 bytes per position.  All pointers 16-byte aligned
 * (FQ_ERR_UNSUPPORTED otherwise), n floats each.  One launch per chain. */
#define FQ_CHAIN_MAX 6
typedef struct fq_chain_seg {
    const float* head;                 /* the first block's shortcut (a projection's output, or a kept ReLU output) */
    const float* y[FQ_CHAIN_MAX];      /* conv3 outputs of the chain's blocks, in order */
    size_t n;
    int32_t len;                       /* 1 .. FQ_CHAIN_MAX */
    int32_t row_y[FQ_CHAIN_MAX];
    int32_t row_sum[FQ_CHAIN_MAX];
} fq_chain_seg;
int fq_hist2048_chain_seg(const fq_chain_seg* segs, int nseg, const float* interval, int64_t* hist,
                          fq_stream_t stream);

/* INTERVAL_NUM other than 2048 (tools/configs.yml:24; distribution_collector.py:9-14 takes it as interval_num and
 * quantizer.py:98-167 sweeps t = 128 .. distribution.size - 1, whatever the size): the segmented histogram and the exhaustive
 * KL sweep for bins in {512, 1024, 2048, 4096} (FQ_ERR_UNSUPPORTED otherwise; 2048 is the entry points above).
 *   fq_hist_seg_n         hist is int64[rows][bins]; bin = min((int)(|x| / interval[row]), bins - 1) for x != 0
 *   fq_kl_threshold_n     hist int64[rows][bins] -> thr_out in [128, bins - 1]; kl_curve_out NULL or float64[rows][bins - 128]
 * The producer-fused statistics, the per-channel rows and the screened search exist for 2048 only. */
int fq_hist_seg_n(const fq_seg* segs, int nseg, const float* interval, int64_t* hist, int bins, fq_stream_t stream);
size_t fq_kl_workspace_bytes_n(int rows, int bins);
int fq_kl_threshold_n(const int64_t* hist, int rows, int bins, int32_t* thr_out, double* kl_curve_out,
                      void* workspace, size_t workspace_bytes, fq_stream_t stream);

/* HOST helper (no device work): quantizer.py:86-90
 *   thr_val = fl32((t + 0.5) * interval);  bits = 7 - ceil(log(thr_val) / log(2))
 * evaluated with the host libm in float64 exactly as CPython's math.log(x, 2) does.
 * All pointers are host pointers. */
int fq_bits_from_threshold(const int32_t* thr, const float* interval, int rows,
                           int32_t* bits_out, float* thr_val_out);

/* HOST helper: pytorch_quantizer.py:651-653  bits = 7 - ceil(log(absmax)/log(2)). absmax > 0. */
int fq_bits_from_absmax(const float* absmax, int n, int32_t* bits_out);

/* A tensor and a sum in one pass (pass 2 of a calibration whose pass 1 did not write a residual block's Eltwise output because
 * both of its operands are kept anyway): for every pair, a[i] is counted into row_a (skipped when row_a = -1) and
 * fl32(a[i] + b[i]) -- the fp32 addition fabu_layer.py:5-11 performs -- into row_sum, exactly as fq_hist2048_seg would count
 * the stored tensors (distribution_collector.py:127-135).  a and b are device pointers to n floats each, 16-byte aligned
 * (FQ_ERR_UNSUPPORTED otherwise: the caller then materialises the sum); row_a != row_sum. */
typedef struct fq_pair_seg {
    const float* a;
    const float* b;
    float* relu_out;       /* NULL, or n floats (16-byte aligned, not aliasing a or b of ANY pair of the call) that receive
                            * max(a + b, 0) as nn.ReLU computes it (NaN kept): the NEXT block's shortcut, re-made in pass 2
                            * instead of kept from pass 1 */
    size_t n;
    int32_t row_a;
    int32_t row_sum;
} fq_pair_seg;
int fq_hist2048_pair_seg(const fq_pair_seg* segs, int nseg, const float* interval, int64_t* hist,
                         fq_stream_t stream);

/* ---- per-channel rows (extension: the reference calibrates per tensor, distribution_collector.py:40-42) ---- */
/* One histogram row per (tensor, channel): tensor [N][C][HW] (dense NCHW; HW = 1 for [N][F]) feeds rows
 * row0 .. row0 + C - 1, channel c being the N planes at (n*C + c)*HW.  Same arithmetic per row as the
 * segmented entry points above (fq_absmax_seg / fq_hist2048_seg), the tensor is read in place. */
typedef struct fq_chan_seg {
    const float* ptr;      /* device, 4-byte aligned */
    int32_t N, C;
    int64_t HW;
    int32_t row0;
    int32_t reserved;      /* must be 0 */
} fq_chan_seg;
int fq_absmax_chan(const fq_chan_seg* segs, int nseg, float* max_inout, fq_stream_t stream);
/* ORDERING CONTRACT of fq_hist2048_chan: a workgroup that is the only one of its launch to feed a group of rows adds its
 * counts to them with plain 16-byte read-add-writes, not atomics (the "owner flush": 7x7 planes 2.0 -> 3.05 TB/s).  Every
 * writer of `hist` rows [row0, row0 + C) of the segments of one call must therefore be ORDERED with that call -- same
 * stream, or an event between them.  A launch on another stream that adds to the same rows at the same time (a statistics
 * side stream, an atomic producer) loses counts.  Within one call the library checks that no two segments share rows
 * and falls back to atomics for the call when they do.  FQ_CHAN_OWN_FLUSH=0 in the environment: atomics everywhere. */
int fq_hist2048_chan(const fq_chan_seg* segs, int nseg, const float* interval, int64_t* hist,
                     fq_stream_t stream);

/* ---- element-wise quantisation ops (new_quantity_op.py) ------------------------------------ */
/* bitwidth is 8 or 16 (QUANTIZE_BIT, new_quantity_op.py:8): clamp range [-128,127] / [-32768,32767].
 * In-place (y == x) is allowed for all of them. */

/* QuanDequan.forward, new_quantity_op.py:246-257: clamp(rint(x * 2^bit)) / 2^bit */
int fq_quandequan_f32(const float* x, float* y, size_t n, int bit, int bitwidth, fq_stream_t stream);
/* Quantity.forward, :52-58: clamp(rint(x * 2^ib)) (round half to even) */
int fq_quantity_f32(const float* x, float* y, size_t n, int ib, int bitwidth, fq_stream_t stream);
/* DeQuantity.forward, :66-68: x / 2^ob */
int fq_dequantity_f32(const float* x, float* y, size_t n, int ob, fq_stream_t stream);
/* Sp.forward, :76-91: clamp(x) */
int fq_sp_f32(const float* x, float* y, size_t n, int bitwidth, fq_stream_t stream);
/* RightShift.forward, :17-44: v = x / 2^rs; clamp(trunc(v + (v > 0 ? 0.5 : -0.5))) (half away) */
int fq_rightshift_f32(const float* x, float* y, size_t n, int rs, int bitwidth, fq_stream_t stream);
/* NewAdd.forward, :171-174: clamp(a + b) */
int fq_add_sat_f32(const float* a, const float* b, float* y, size_t n, int bitwidth,
                   fq_stream_t stream);
/* NewConv2d / NewLinear tail fused, :127-132: for acc[outer][C][inner]
 *   y = clamp( RightShift(acc, rs) + qbias[c] ) / 2^ob      (qbias fp32[C], integer valued) */
int fq_recon_epilogue_f32(const float* acc, const float* qbias, float* y, size_t outer, size_t C,
                          size_t inner, int rs, int ob, int bitwidth, fq_stream_t stream);
/* weight quantiser, pytorch_quantizer.py:656-657,:663: (int32) clip(around(w * 2^bit), -128, 127) */
int fq_quantize_param_i32(const float* w, int32_t* q, size_t n, int bit, fq_stream_t stream);

/* ---- integer contraction of NewConv2d / NewLinear on the matrix cores -------------------------- */

/* The bias add of a float convolution with the calibration's running abs-max (distribution_collector.py:70-78) taken
 * on the way out:  y[n][c][hw] += bias[c]  in place (what torch's Conv2d does after the MIOpen convolution, same
 * single fp32 rounding) and  *max_inout = max(*max_inout, max |y|)  -- pass 1 then does not read this tensor again.
 * relu_out (may be NULL): also writes max(y, 0) there (NaN kept, torch's clamp_min) -- the nn.ReLU that follows, served in
 * the same pass.  y: fp32 [N][C][HW] contiguous, bias: fp32 [C], max_inout: one fp32 (>= 0) in device memory, N*C*HW < 2^32. */
int fq_bias_add_absmax_f32(float* y, const float* bias, int N, int C, int HW, float* max_inout, float* relu_out,
                           fq_stream_t stream);

/* Eltwise.forward of the float model (fabu_layer.py:16-19, x + y) with the calibration's running abs-max
 * (distribution_collector.py:70-78) taken on the way out:  z[i] = x[i] + y[i]  and  *max_inout = max(*max_inout, max |z|).
 * relu_out (may be NULL): also writes max(z, 0) there.  x, y, z, relu_out: fp32, 16-byte aligned, n elements (z may alias
 * x or y); max_inout: one fp32 (>= 0) in device memory. */
int fq_add_absmax_f32(const float* x, const float* y, float* z, size_t n, float* max_inout, float* relu_out, fq_stream_t stream);

/* The same two producers in calibration pass 2 (distribution_collector.py:127-135 taken on the way out): identical
 * outputs (y += bias[c] in place / z = x + y, relu_out as above), and every output value is counted into the 2048-bin
 * histogram row `hist_row` (int64[2048], accumulated into) with the bin width `*interval` (one fp32 in device memory) --
 * exactly what fq_hist2048_seg would add for this tensor, without reading it a second time. */
int fq_bias_add_hist_f32(float* y, const float* bias, int N, int C, int HW, const float* interval, int64_t* hist_row,
                         float* relu_out, fq_stream_t stream);
int fq_add_hist_f32(const float* x, const float* y, float* z, size_t n, const float* interval, int64_t* hist_row,
                    float* relu_out, fq_stream_t stream);

/* The float 1x1 convolutions of the calibration forward on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
 * an fmaf chain over ci = 0 .. Cin-1), with the statistic the calibration takes of the output folded into the epilogue --
 * the convolution itself is the producer, so the bias-add pass above disappears for these layers:
 *   y[n][co][oh][ow] = sum_ci W[co][ci] * x[n][ci][oh*stride][ow*stride] + bias[co]      (padding 0, groups 1)
 * wt: the weights TRANSPOSED, fp32 [Cin][Cout], 16-byte aligned, Cout % 4 == 0; bias: fp32 [Cout] or NULL;
 * x: fp32 [N][Cin][Hin][Win] contiguous, < 2^30 elements (FQ_ERR_UNSUPPORTED beyond); y: fp32 [N][Cout][Hout][Wout],
 * Hout = (Hin-1)/stride + 1, < 2^30 elements;
 * relu_out (may be NULL): max(y, 0) as well (the nn.ReLU behind the convolution); with relu_out given y may be NULL: y is then
 *   not written (its statistic is still taken) -- for callers that know nothing but that ReLU reads the convolution's output
 *   (the same holds for fq_conv_kxk_f32 and fq_conv_stem_f32);
 * exactly one of {max_inout, hist_row} may be given (both NULL: plain convolution):
 *   max_inout: *max_inout = max(*max_inout, max |y|)                 (distribution_collector.py:70-78)
 *   hist_row + interval: y counted into int64[2048] with bin width *interval   (distribution_collector.py:127-135)
 * Replaces, inside the float forward the reference runs at pytorch_quantizer.py:288-296, torch's Conv2d for these layers;
 * not bit-identical to the library's convolution (different, fixed summation order), deterministic from run to run.
 * workspace / workspace_bytes (all six entry points of this family; may be NULL / 0): device scratch of
 *   fq_conv_f32_workspace_bytes() bytes, 16-byte aligned, ZERO-FILLED ONCE by the caller before its first use (the kernels
 *   leave its counters at zero, so it can be handed to launch after launch); one workspace per stream whose launches may
 *   overlap.  With it, a launch whose tile count leaves a partly filled last round over the 256 CUs cuts the tiles of that
 *   round along the reduction into slices that meet in the workspace (the tail split, DESIGN.md section 3: 784-tile launches
 *   18-20 % faster).  A split tile's value is the sum of <= 16 partial fma chains added in slice order instead of one chain:
 *   deterministic, but WHICH tiles are split depends on the tile count, i.e. on the batch size N -- the same images pushed
 *   through in different batch sizes may differ in the last bit of those outputs.  NULL (or FQ_CONV_TAIL_SPLIT=0 in the
 *   environment): never split; every output is one chain over ci = 0 .. Cin-1 whatever N is. */
size_t fq_conv_f32_workspace_bytes(void);
int fq_conv1x1_f32(const float* x, const float* wt, const float* bias, float* y, float* relu_out, int N, int Cin, int Hin,
                   int Win, int Cout, int stride, float* max_inout, const float* interval, int64_t* hist_row,
                   void* workspace, size_t workspace_bytes, fq_stream_t stream);

/* The 1x1 convolutions above with every fp32 operand as THREE bf16 values on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16: 14 x
 * the rate of the fp32 MFMA on this chip).  v = hi + mid + lo with hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid) is exact
 * (8 + 8 + 8 significant bits); of the nine bf16 products of w x the six largest are accumulated in fp32 (the three left out are
 * below 2^-23 of the product).  Measured against fp64: 1.0-1.3e-7 of sum |w||x| -- the fp32 fma chain of fq_conv1x1_f32: 1.4-2.1e-7
 * (profiles/r04_bf16x3_probe.txt; tests/test_gpu_float_forward_kernels.py holds both to the same bound); integer-valued operands
 * below 2^8 are exact as before; a non-finite input gives NaN.  Same operands, epilogues, statistics, workspace / tail split and
 * error codes as the entry points they shadow, except:
 *   wsb: the weights packed by fq_conv1x1_sb_pack from fp32 [Cout][Cin] (NOT transposed) into bf16 [Cin / 16][plane: hi, mid, lo]
 *        [k % 16 / 8][Cout][8] -- the order the kernel's lanes load their MFMA operands in, straight from global memory --
 *        fq_conv1x1_sb_packed_bytes(Cin, Cout) bytes, 16-byte aligned;  Cin % 16 == 0, Cout % 4 == 0 (fq_conv1x1_sb_supported). */
size_t fq_conv1x1_sb_packed_bytes(int Cin, int Cout);
int fq_conv1x1_sb_supported(int Cin, int Cout);
int fq_conv1x1_sb_pack(const float* w_kc, void* wsb, int Cin, int Cout, fq_stream_t stream);
int fq_conv1x1_sb_f32(const float* x, const void* wsb, const float* bias, float* y, float* relu_out, int N, int Cin, int Hin, int Win,
                      int Cout, int stride, float* max_inout, const float* interval, int64_t* hist_row, void* workspace,
                      size_t workspace_bytes, fq_stream_t stream);
int fq_conv1x1_sb_qd_f32(const float* x, const void* wsb, const float* bias, float* y, int N, int Cin, int Hin, int Win, int Cout,
                         int stride, int bit, int bitwidth, void* workspace, size_t workspace_bytes, fq_stream_t stream);
int fq_conv1x1_sb_add_f32(const float* x, const void* wsb, const float* bias, const float* res, float* y, float* sum, float* relu_out,
                          int N, int Cin, int Hin, int Win, int Cout, int stride, float* max_y, float* max_sum, void* workspace,
                          size_t workspace_bytes, fq_stream_t stream);
int fq_conv1x1_sb_add_hist_f32(const float* x, const void* wsb, const float* bias, const float* res, float* relu_out, int N, int Cin,
                               int Hin, int Win, int Cout, int stride, const float* interval_y, int64_t* hist_y,
                               const float* interval_sum, int64_t* hist_sum, void* workspace, size_t workspace_bytes,
                               fq_stream_t stream);

/* The same kernel for R x S convolutions with zero padding (ResNet's 3x3 layers, stride 1 and 2): the reduction runs tap by
 * tap over the same x rows, shifted -- a 1x1 convolution per tap whose per-thread pixel offset (or "outside the image:
 * zero") is worked out once per tap.  wt: fp32 [(r*S + s)*Cin + ci][Cout] = W[co][ci][r][s], 16-byte aligned;
 * Cin % 16 == 0, Cout % 4 == 0, dilation 1, groups 1 (FQ_ERR_UNSUPPORTED otherwise);
 * Hout = (Hin + 2 pad - R)/stride + 1.  Epilogue contract as fq_conv1x1_f32.  With this, fq_conv1x1_f32 and fq_conv_stem_f32
 * every convolution of a ResNet's float forward runs on this library: the calibration forward is deterministic
 * (the convolution library's Winograd kernels are not reproducible from call to call) and does not touch that library. */
int fq_conv_kxk_f32(const float* x, const float* wt, const float* bias, float* y, float* relu_out, int N, int Cin, int Hin,
                    int Win, int Cout, int R, int S, int stride, int pad, float* max_inout, const float* interval,
                    int64_t* hist_row, void* workspace, size_t workspace_bytes, fq_stream_t stream);

/* The stride-1, pad-1 3x3 convolutions of the same forward as Winograd F(2x2, 3x3) on the fp32 matrix cores: 16 products per
 * 2x2 output tile, input channel and output channel instead of 36 (fq_conv_kxk_f32's direct sum), the transforms fused into
 * the kernel, same epilogue contract as fq_conv1x1_f32 (bias, relu_out, y may be NULL with relu_out given, exactly one of
 * {max_inout, hist_row + interval} or neither).  Replaces, inside the float forward the reference runs at
 * pytorch_quantizer.py:288-296, torch's Conv2d for these layers (whose library kernels are Winograd forms as well).
 * u: the weights transformed and packed by fq_conv3x3_wino_f32_pack -- fq_conv3x3_wino_f32_packed_floats(Cin, Cout) floats,
 *   16-byte aligned, U = G g Gt computed in fp64 and rounded once, laid out [ci / 8][position 0..15][ci % 2][co][ci % 8 / 2];
 *   w_kcrs: fp32 [Cout][Cin][3][3] on the device.
 * x: fp32 [N][Cin][H][W], y / relu_out: fp32 [N][Cout][H][W]; Cin % 8 == 0, Cout % 64 == 0, H * W >= 4, x and y below 2^31
 *   BYTES (fq_conv3x3_wino_f32_supported; FQ_ERR_UNSUPPORTED otherwise: callers keep fq_conv_kxk_f32 there).  bias: 16-byte
 *   aligned or FQ_ERR_UNSUPPORTED.  No byte outside [x, x + 4 N Cin H W) is read: x needs no slack behind it.
 * Numerics: per output an fmaf chain over ci = 0 .. Cin-1 for each of the 16 positions, the fixed additions of the output
 *   transform, then the bias: deterministic, independent of N (no tail split, no workspace), NOT the direct sum -- it differs
 *   from fq_conv_kxk_f32 by a few units in the last place of the LARGEST term (tests/test_gpu_conv_wino.py states the bound). */
int fq_conv3x3_wino_f32_supported(int N, int Cin, int Hin, int Win, int Cout);
size_t fq_conv3x3_wino_f32_packed_floats(int Cin, int Cout);
int fq_conv3x3_wino_f32_pack(const float* w_kcrs, float* u, int Cin, int Cout, fq_stream_t stream);
int fq_conv3x3_wino_f32(const float* x, const float* u, const float* bias, float* y, float* relu_out, int N, int Cin, int Hin,
                        int Win, int Cout, float* max_inout, const float* interval, int64_t* hist_row, fq_stream_t stream);

/* The float stem convolution of the calibration forward (ResNet-50/101's conv1: 7x7, stride 2, 3 -> Cout <= 64 channels,
 * any padding) on the fp32 matrix cores, same epilogue contract as fq_conv1x1_f32 (bias, relu_out, and exactly one of
 * {max_inout, hist_row + interval} or neither).  wp: the weights PACKED as fp32 [fq_conv_stem_f32_packed_rows()][64],
 * 16-byte aligned: W[co][c][r][s] at row (c*R + r)*8 + s, column co; zero for s >= S and co >= Cout (the tap axis is
 * padded to 8 so that a k-pair of the MFMA is an (even, odd) column pair of the stride-2 input).
 * x: fp32 [N][Cin][H][W]; y: fp32 [N][Cout][Hout][Wout], Hout = (H + 2 pad - R)/stride + 1; both < 2^30 elements.
 * FQ_ERR_UNSUPPORTED for any other (Cin, R, S, stride) or Cout > 64: callers keep the library convolution there. */
int fq_conv_stem_f32_packed_rows(int Cin, int R, int S);
int fq_conv_stem_f32(const float* x, const float* wp, const float* bias, float* y, float* relu_out, int N, int Cin, int H,
                     int W, int Cout, int R, int S, int stride, int pad, float* max_inout, const float* interval,
                     int64_t* hist_row, fq_stream_t stream);

/* TestConv.forward / TestLinear.forward (new_quantity_op.py:283-292, :248-256) in ONE kernel: the float convolution above
 * with QuanDequan(bit) applied to each value as it leaves the accumulator,
 *   y = clamp(rint((conv(x) + bias) * 2^bit), lo, hi) / 2^bit,   [lo, hi] = the integer range of bitwidth (8 or 16),
 * instead of the reference's two passes (the convolution's store, then an 8 B/element read-modify-write).  The value
 * QuanDequan sees is exactly what fq_conv1x1_f32 / fq_conv_kxk_f32 / fq_conv3x3_wino_f32 / fq_conv_stem_f32 would have stored, so the result equals
 * fq_quandequan_f32 of their output bit for bit.  Same operand contracts as the plain entry points. */
int fq_conv1x1_qd_f32(const float* x, const float* wt, const float* bias, float* y, int N, int Cin, int Hin, int Win,
                      int Cout, int stride, int bit, int bitwidth, void* workspace, size_t workspace_bytes, fq_stream_t stream);
int fq_conv_kxk_qd_f32(const float* x, const float* wt, const float* bias, float* y, int N, int Cin, int Hin, int Win,
                       int Cout, int R, int S, int stride, int pad, int bit, int bitwidth, void* workspace,
                       size_t workspace_bytes, fq_stream_t stream);
int fq_conv3x3_wino_qd_f32(const float* x, const float* u, const float* bias, float* y, int N, int Cin, int Hin, int Win, int Cout,
                           int bit, int bitwidth, fq_stream_t stream);     /* (fq_conv3x3_wino_f32's operands and limits) */
int fq_conv_stem_qd_f32(const float* x, const float* wp, const float* bias, float* y, int N, int Cin, int H, int W,
                        int Cout, int R, int S, int stride, int pad, int bit, int bitwidth, fq_stream_t stream);

/* The last 1x1 convolution of a residual block together with the `Eltwise` that consumes it and the ReLU behind that
 * (fabu_layer.py:5-11 called from the model the reference runs at pytorch_quantizer.py:288-296), calibration pass 1, in ONE
 * kernel:  v = conv1x1(x) + bias  (abs-max folded into *max_y; stored to y unless y is NULL),  s = v + res  (abs-max folded
 * into *max_sum; stored to sum unless sum is NULL),  relu_out = max(s, 0)  (NaN stays NaN).  res, y, sum, relu_out: fp32
 * [N][Cout][Hout][Wout].  Bit for bit what fq_conv1x1_f32 (max form) followed by fq_add_absmax_f32 leave; it exists because
 * those two move 20 bytes per element and this moves 8 when neither v nor s is kept for pass 2.
 * FQ_ERR_UNSUPPORTED unless Cin % 16 == 0 and Cout % 128 == 0 (callers keep the two kernels there). */
int fq_conv1x1_add_f32(const float* x, const float* wt, const float* bias, const float* res, float* y, float* sum,
                       float* relu_out, int N, int Cin, int Hin, int Win, int Cout, int stride, float* max_y, float* max_sum,
                       void* workspace, size_t workspace_bytes, fq_stream_t stream);
/* ... and in pass 2: v and s are counted into the 2048-bin rows hist_y / hist_sum (bin widths *interval_y / *interval_sum,
 * the rule of fq_hist2048_seg) while they pass through the registers and are not written at all; relu_out = max(s, 0).
 * Bit for bit the rows fq_conv1x1_f32 (histogram form) followed by fq_add_hist_f32 leave. */
int fq_conv1x1_add_hist_f32(const float* x, const float* wt, const float* bias, const float* res, float* relu_out, int N, int Cin,
                            int Hin, int Win, int Cout, int stride, const float* interval_y, int64_t* hist_y,
                            const float* interval_sum, int64_t* hist_sum, void* workspace, size_t workspace_bytes,
                            fq_stream_t stream);

/* The pooling layers of the float calibration forward (nn.MaxPool2d / a global nn.AvgPool2d inside the model the
 * reference runs at pytorch_quantizer.py:288-296), bit for bit what torch computes:
 * fq_maxpool2d_f32: y[plane][oy][ox] = max over the window clipped to the image (NaN propagates); x fp32 [planes][H][W],
 *   y fp32 [planes][Ho][Wo], Ho = (H + 2 ph - kh)/sh + 1 (floor mode, no dilation).
 * fq_avgpool_global_f32: y[plane] = (x[plane][0] + x[plane][1] + ... in this order, fp32) / HW; HW <= 144. */
int fq_maxpool2d_f32(const float* x, float* y, int planes, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                     fq_stream_t stream);
int fq_avgpool_global_f32(const float* x, float* y, int planes, int HW, fq_stream_t stream);

/* Quantity.forward (new_quantity_op.py:52-58) fused with the layout change the MFMA kernel wants:
 *   y[n][hw][c] = (int8) clamp(rint(x[n][c][hw] * 2^ib), -128, 127),  c in [C, Cpad) = 0
 * x: fp32 [N][C][HW] (NCHW), y: int8 [N][HW][Cpad] (NHWC), Cpad >= C, Cpad % 4 == 0 (use a multiple
 * of 16 for fq_conv2d_i8).  HW == 1 covers Linear inputs [N][F]. */
int fq_quantize_i8_nhwc(const float* x_nchw, int8_t* y_nhwc, int N, int C, int HW, int Cpad, int ib,
                        fq_stream_t stream);

/* Stem variant of fq_quantize_i8_nhwc for C <= 4 inputs (e.g. 7x7 stride-2 on RGB): the kernel WIDTH is
 * folded into the channel axis,
 *   y[n][ih][q][s*C + c] = Quantity(x[n][c][ih][q*stride_w - pad_w + s*dil_w])   (0 outside the image
 *   and for folded channels >= S*C),   Q = (W + 2*pad_w - dil_w*(S-1) - 1)/stride_w + 1,
 * y: int8 [N][H][Q][Cpad2], Cpad2 >= S*C, Cpad2 % 16 == 0.  fq_conv2d_i8 is then called on y with
 * W := Q, C := Cpad2, S := 1, stride_w := 1, pad_w := 0, dil_w := 1 and weights folded the same way
 * ([K][R][1][Cpad2]): same integers, R*Cpad2 reduction bytes instead of R*S*16. */
int fq_quantize_i8_unfold_w(const float* x_nchw, int8_t* y, int N, int C, int H, int W, int S,
                            int stride_w, int pad_w, int dil_w, int Cpad2, int ib, fq_stream_t stream);

/* The whole stem NewConv2d (new_quantity_op.py:124-133 on the network input, C <= 4) followed by nn.ReLU when
 * relu != 0, in one kernel and without the width-unfolded copy of the image:
 *   q[n][p][q][k] = Sp( RightShift( sum_{r,s,c} w[k][c][r][s] * Quantity_ib(x)[n][c][p*sh-ph+r][q*sw-pw+s], rs ) + qbias[k] )
 *   (max(.,0) with relu), the integers in front of DeQuantity(ob): value = q * 2^-ob.
 * x: fp32 NCHW.  w_stem: int8 [R][64][32], byte 4*s + c of row (r, k) = the quantised weight[k][c][r][s], every other
 * byte 0 (rows k >= K zero).  q_nhwc: int8 [N][P][Q][Kpad], Kpad % 16 == 0, K <= Kpad <= 64, channels >= K written
 * as zeros.  Limits: C <= 4, S <= 8, R <= 8, K <= 64, dilation 1, 1 <= rs <= 16, input patch of an 8x16 output tile
 * ((7*sh + R) x (15*sw + S) pixels) <= 1024 pixels; FQ_ERR_INVALID_ARG otherwise (callers then use
 * fq_quantize_i8_unfold_w + fq_conv2d_i8_resident, which compute the same integers). */
int fq_conv2d_i8_stem(const float* x_nchw, const int8_t* w_stem, const float* qbias, int8_t* q_nhwc, int Kpad, int relu,
                      int N, int C, int H, int W, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                      int pad_w, int ib, int rs, int ob, fq_stream_t stream);

/* NewConv2d.forward / NewLinear.forward after Quantity (new_quantity_op.py:126-132, :199-204):
 *   acc[n][k][p][q] = sum_{r,s,c} w[k][r][s][c] * x[n][p*stride-pad+r*dil][q*stride-pad+s*dil][c]   (int32, exact)
 *   y = clamp( RightShift(acc, rs) + qbias[k] ) / 2^ob                                    (fp32 NCHW)
 * implicit GEMM on v_mfma_i32_32x32x32_i8.  x: int8 NHWC [N][H][W][C], w: int8 [K][R][S][C], both
 * 16-byte aligned, C % 16 == 0 (zero-pad channels; FQ_ERR_UNSUPPORTED otherwise), groups == 1.
 * qbias: fp32[K] integer valued, of ANY magnitude (a bias beyond the output range saturates the output exactly as the reference's
 *   fp32 BiasAdd + Sp does; tests/test_gpu_conv_i8.py).  y: fp32 [N][K][P][Q].  A Linear layer is H = W = R = S = 1.
 * The reference computes acc with an fp32 convolution, exact while |partial sums| < 2^24; in that
 * regime the results are identical, beyond it this kernel stays exact. */
int fq_conv2d_i8(const int8_t* x_nhwc, const int8_t* w_krsc, const float* qbias, float* y_nchw,
                 int N, int H, int W, int C, int K, int R, int S, int stride_h, int stride_w,
                 int pad_h, int pad_w, int dil_h, int dil_w, int rs, int ob, int bitwidth,
                 fq_stream_t stream);

/* ---- resident integer activations (ReconModel without the fp32 module-boundary round trips) ------ */
/* Between two integer layers the reference moves fp32 NCHW: DeQuantity (new_quantity_op.py:66-68), nn.ReLU,
 * then the next layer's Quantity(ib) (:52-58) -- which recovers exactly the integer the previous tail held.
 * These entry points keep that integer in HBM instead (int8 / int16 NHWC, channels padded to 16 with
 * zeros); the values they stand for, q * 2^-g, are bit-identical to the reference's fp32 tensors. */

/* fq_conv2d_i8 with selectable outputs:
 *   y_nchw (may be NULL): fp32 [N][K][P][Q] as fq_conv2d_i8
 *   q_nhwc (may be NULL): int8 [N][P][Q][Kpad], the integer BEFORE DeQuantity,
 *                         clamp(RightShift(acc, rs) + qbias[k]) = Quantity(y, ib = ob); channels [K, Kpad) = 0;
 *                         Kpad >= K, Kpad % 16 == 0, 16-byte aligned; N * P * Q * Kpad < 2^30 elements (32-bit offsets
 *                         into the integer tensors, also for the residual / sum of fq_conv2d_i8_add_resident:
 *                         FQ_ERR_UNSUPPORTED beyond)
 *   relu != 0: a following nn.ReLU is fused into both outputs (max(., 0) commutes with the scale 2^-ob).
 * bitwidth is 8.  At least one output must be given. */
int fq_conv2d_i8_resident(const int8_t* x_nhwc, const int8_t* w_krsc, const float* qbias, float* y_nchw,
                          int8_t* q_nhwc, int Kpad, int relu, int N, int H, int W, int C, int K, int R,
                          int S, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                          int rs, int ob, fq_stream_t stream);

/* fq_conv2d_i8_resident followed by fq_add_resident in one kernel: the convolution's int8 result (no ReLU of
 * its own, grid ob) is operand x of NewAdd and never reaches HBM; `res` is operand y (int8 / int16 NHWC
 * [N][P][Q][Kpad], grid g_res).  Outputs as fq_add_resident: wide (int16, grid g_wide = max(0, ob, g_res) <= 8,
 * may be NULL), narrow (int8 = Quantity(ib), may be NULL), relu fuses the nn.ReLU after the add. */
int fq_conv2d_i8_add_resident(const int8_t* x_nhwc, const int8_t* w_krsc, const float* qbias, const void* res,
                              int res_bytes, int g_res, int16_t* wide, int g_wide, int8_t* narrow, int ib,
                              int relu, int Kpad, int N, int H, int W, int C, int K, int R, int S, int stride_h,
                              int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int rs, int ob,
                              fq_stream_t stream);

/* The tail of a bottleneck block and the head of the next one in ONE kernel (round 4):
 *   NewConv2d.forward of conv3 (1x1, C -> K3; new_quantity_op.py:124-133)  ->  NewAdd.forward with the shortcut (:166-174)
 *   -> nn.ReLU -> the next block's conv1: NewConv2d.forward again (1x1, K3 -> C2, Quantity(ib) on the sum, its own tail and
 *   ReLU) -- what fq_conv2d_i8_add_resident followed by fq_conv2d_i8_resident compute, bit for bit, without the int8
 *   re-quantisation of the sum ever reaching HBM (it is the second convolution's operand, staged in LDS).
 * x_nhwc int8 [M][C]; w3_krsc int8 [K3][C]; qbias3 fp32 [K3] integer valued; rs3 / ob3 as in fq_conv2d_i8_resident;
 * res / res_bytes / g_res / wide / g_wide / narrow / ib / relu exactly as in fq_conv2d_i8_add_resident ([M][K3] tensors;
 *   narrow may be NULL when nothing but the fused conv1 reads it; wide may be NULL);
 * w1_krsc int8 [C2][K3], qbias1 fp32 [C2], rs1, relu1, q1_nhwc int8 [M][C2]: the next conv1 (its input bit is `ib`);
 *   C2 = 0 (w1 / qbias1 / q1 NULL): conv3 + NewAdd alone.
 * 1x1, stride 1, no padding: spatial shape does not matter, M = N * H * W pixels.
 * fq_block_tail_i8_supported: 1 for C in {64, 128, 256}, K3 in {128 .. 1024} a multiple of 128, C2 in {0, 64, 128} (0 only behind
 * C = 256) and shifts in 1 .. 16 (the integer tails); the grids (ob3, g_res, res_bytes, ib) are part of the query so that a later
 * kernel may specialise on them -- today every grid fq_conv2d_i8_add_resident takes is taken.  fq_block_tail_i8 returns
 * FQ_ERR_UNSUPPORTED otherwise and callers keep the two launches. */
int fq_block_tail_i8_supported(int C, int K3, int C2, int rs3, int rs1, int ob3, int g_res, int res_bytes, int ib);
int fq_block_tail_i8(const int8_t* x_nhwc, const int8_t* w3_krsc, const float* qbias3, int rs3, int ob3, const void* res,
                     int res_bytes, int g_res, int16_t* wide, int g_wide, int8_t* narrow, int ib, int relu,
                     const int8_t* w1_krsc, const float* qbias1, int rs1, int relu1, int8_t* q1_nhwc, long M, int C, int K3,
                     int C2, fq_stream_t stream);

/* fq_block_tail_i8 for the FIRST block of a stage (round 5), whose shortcut is not a tensor but a projection: a 1x1
 * NewConv2d.forward (new_quantity_op.py:124-133; stride 1 or 2, no padding, no ReLU) of the block's input.  As its own launch
 * that convolution writes K3 bytes per pixel which the tail reads straight back; here the workgroup computes its 128 pixels of
 * the projection -- contraction, RightShift(rsp) + BiasAdd + Sp, int8 on the grid obp -- slice by slice in front of conv3's slice:
 * the same integers as fq_conv2d_i8_resident followed by fq_block_tail_i8 with res = its output (res_bytes 1, g_res = obp),
 * bit for bit (tests/test_gpu_block_tail.py), CP bytes per pixel read instead of K3, nothing written.
 * xp_nhwc int8 [N][Hp][Wp][CP]: the projection's input; wp_krsc int8 [K3][CP]; qbiasp fp32 [K3] integer valued; stride_p in
 *   {1, 2} with H = (Hp - 1) / stride_p + 1, W likewise; every other argument as in fq_block_tail_i8 with M = N * H * W.
 * fq_block_tail_proj_i8_supported: today C = CP = 64, C2 in {0, 64} (ResNet's first stage), K3 and the shifts as above;
 * FQ_ERR_UNSUPPORTED otherwise and callers keep the projection as its own launch.  fq_conv2d_i8_last_variant: 12. */
int fq_block_tail_proj_i8_supported(int C, int K3, int C2, int CP, int rs3, int rs1, int rsp, int stride_p);
int fq_block_tail_proj_i8(const int8_t* x_nhwc, const int8_t* w3_krsc, const float* qbias3, int rs3, int ob3,
                          const int8_t* xp_nhwc, const int8_t* wp_krsc, const float* qbiasp, int rsp, int obp, int stride_p,
                          int Hp, int Wp, int16_t* wide, int g_wide, int8_t* narrow, int ib, int relu, const int8_t* w1_krsc,
                          const float* qbias1, int rs1, int relu1, int8_t* q1_nhwc, int N, int H, int W, int C, int K3, int C2,
                          int CP, fq_stream_t stream);

/* NewAdd.forward (new_quantity_op.py:171-174) on resident operands, with the nn.ReLU and the Quantity of
 * the consumers fused.  x, y: int8 (x_bytes = 1) or int16 (x_bytes = 2) arrays of n elements in the same
 * flat NHWC layout, standing for x * 2^-gx and y * 2^-gy.
 *   s         = clamp(x * 2^-gx + y * 2^-gy, relu ? 0 : -128, 127)       (the reference's fp32 expression)
 *   wide[i]   = (int16) (s * 2^g_wide),  g_wide = max(0, gx, gy) <= 8      exact: feeds the next residual add
 *   narrow[i] = (int8) clamp(rint(s * 2^ib), -128, 127)                   = Quantity(ib) of the next conv
 * wide / narrow may be NULL (not both).  n % 16 == 0, all pointers 16-byte aligned.
 * FQ_ERR_UNSUPPORTED if the exact sum does not fit int16 (g_wide > 8): use the fp32 path there. */
int fq_add_resident(const void* x, int x_bytes, int gx, const void* y, int y_bytes, int gy, int16_t* wide,
                    int g_wide, int8_t* narrow, int ib, int relu, size_t n, fq_stream_t stream);

/* Leaving the resident domain (DeQuantity, :66-68, fused with the layout change):
 *   y[n][c][hw] = q[n][hw][c] * 2^-g,   q: int8 / int16 (q_bytes 1 / 2) NHWC [N][HW][Cpad], y: fp32 NCHW. */
int fq_dequant_nhwc_to_nchw(const void* q_nhwc, int q_bytes, int g, float* y_nchw, int N, int C, int HW,
                            int Cpad, fq_stream_t stream);

/* nn.MaxPool2d (dilation 1, ceil_mode False) on a resident int8 NHWC activation: max commutes with the
 * monotone de-quantisation, so y = maxpool(x) on the integers; padding acts as -inf (2*pad <= kernel).
 * x: int8 [N][H][W][Cpad], y: int8 [N][P][Q][Cpad], P = (H + 2*ph - kh)/sh + 1, Cpad % 16 == 0. */
int fq_maxpool_i8_nhwc(const int8_t* x, int8_t* y, int N, int H, int W, int Cpad, int kh, int kw, int sh,
                       int sw, int ph, int pw, fq_stream_t stream);

/* Global average pooling (nn.AvgPool2d whose window is the whole plane) on a resident activation:
 *   y[n][c] = ((float)(sum_hw q[n][hw][c]) * 2^-g) / HW      -- fp32 [N][C]; every partial sum of torch's fp32
 * accumulation is exact here (HW * 2^15 < 2^24, else FQ_ERR_UNSUPPORTED), so this is the same number. */
int fq_avgpool_global_nhwc(const void* q_nhwc, int q_bytes, int g, float* y, int N, int C, int HW, int Cpad,
                           fq_stream_t stream);

/* ---- input files ------------------------------------------------------------------------------- */

/* HOST helper for PRE_PROCESS.IMG = 2 (pytorch_quantizer.py:276-280: np.load of one CHW fp32 image per calibration item):
 * n .npy files that share one header (same dtype, order and shape: `header` / `header_bytes` are the bytes in front of the
 * payload of any one of them) are read into consecutive slots of dst -- fp32 [n][elems_per_file], host memory the caller
 * owns (pinned, so that the H2D copy that follows is asynchronous) -- by `threads` host threads.  ok_out[i] = 1 when file
 * i had exactly that header and payload, else 0 (the caller falls back to its general reader for that slot). */
int fq_read_npy_batch_f32(const char* const* paths, int n, const void* header, size_t header_bytes, float* dst,
                          size_t elems_per_file, int threads, int* ok_out);

/* ---- output files ------------------------------------------------------------------------------ */

/* HOST helper: write an int32 array as nested JSON lists, byte-identical to Python's
 * json.dump(arr.tolist(), fh, indent=indent) -- the format of weight/<param>.json, bias/<param>.json,
 * new_weight/, new_bias/ (pytorch_quantizer.py:663-669, rewriter.py:57-59).  data/shape are host. */
int fq_json_dump_i32(const char* path, const int32_t* data, int ndim, const int64_t* shape, int indent);

#ifdef __cplusplus
}
#endif
#endif /* FQ_H */
