// fq_stem.hip -- the stem convolution of the integer-simulation model in one kernel (gfx950).
//
// NewConv2d on the network input (new_quantity_op.py:104-163 with <= 4 input channels, e.g. 7x7 stride 2 on
// RGB): Quantity(ib) of the fp32 NCHW image, the int8 contraction, RightShift + BiasAdd + Sp (+ the ReLU that
// follows) and the int8 NHWC hand-off to the next integer layer.  The general kernels need 16-channel
// groups, so the stem used to go through a width-unfolded int8 copy of the image (fq_quantize_i8_unfold_w:
// 103 MB written and read 3.5 times for ResNet-50 at batch 128, 58 + 74 us); here the image is read once.
//
//   * persistent workgroups over 8 x 16 output tiles; the input patch of a tile ((8-1)*stride + R rows,
//     (16-1)*stride + S columns) is fetched as fp32 (next tile's loads fly under this tile's MFMAs),
//     quantised, and kept in LDS as one dword per pixel: bytes 0..C-1 = channels, the rest zero;
//   * reduction axis of one filter row = 8 taps x 4 bytes = the 32 bytes of one v_mfma_i32_32x32x32_i8:
//     a pixel's operand fragment is 16 consecutive LDS bytes starting at its leftmost tap -- no im2col
//     copy; taps >= S and channel bytes >= C meet zero weights;
//   * weights [R][64][32] int8 (packed on the host) sit in LDS for the lifetime of the workgroup;
//   * each wave owns a 2 x 16 pixel block and all 64 output channels (2 MFMAs per filter row); the integer
//     tail runs on the accumulators, the bytes are transposed through LDS and leave as 16-byte stores.
#include "fq_common.h"
#include "fq_int_tail.h"

namespace fq {

constexpr int kStemBlock = 256;
constexpr int kStemTH = 8, kStemTW = 16;       // output tile; wave w owns rows 2w, 2w+1
constexpr int kStemPix = 4;                    // patch pixels per thread: patches of up to 1024 pixels
constexpr int kStemMaxR = 8;                   // filter rows (LDS for the weights: R * 2 KB)
constexpr int kStemK = 64;                     // output channels computed (two 32-row MFMA blocks)
constexpr int kStemPatchWords = 1536;          // LDS words of the patch, including the over-read margin
constexpr int kStemOS = kStemK + 16;           // LDS row stride of the output transpose, bytes

struct StemParams {
    const float* x;          // [N][C][H][W]
    const int8_t* w;         // [R][64][32]: byte 4*s + c of row r = weight[k][c][r][s], zero elsewhere
    const float* qbias;      // [K], integer valued
    int8_t* q;               // [N][P][Q][Kpad]
    int N, C, H, W, K, R, S, P, Q, Kpad;
    int sh, sw, ph, pw;
    int PR, PC, PCS;         // patch rows, patch columns, LDS row stride in pixels (>= (TW-1)*sw + 8)
    int tiles_x, tiles_y;
    unsigned ntiles;
    unsigned xcd_chunk;      // tiles per XCD (ceil(ntiles / 8)), 0: tiles dealt round robin (FQ_STEM_XCD=0)
    float scale;             // 2^ib
    int rs, half_rs, ilo, ihi, slo, shi;
};

// kC = input channels (compile time: the quantise and fetch loops touch exactly kC planes).  The waves-per-SIMD target
// keeps the accumulators out of the AGPRs (no v_accvgpr_read per output in the tail).
template <int kC>
__global__ __launch_bounds__(kStemBlock) __attribute__((amdgpu_waves_per_eu(4))) void stem_conv_i8_kernel(const StemParams p) {
    __shared__ __attribute__((aligned(16))) int8_t sW[kStemMaxR * kStemK * 32];
    __shared__ __attribute__((aligned(16))) unsigned sPatch[kStemPatchWords];
    __shared__ __attribute__((aligned(16))) int8_t sOut[4][32 * kStemOS];
    __shared__ int sBiasI[kStemK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;

    // weights and bias: once per workgroup
    {
        const v4i* __restrict__ src = reinterpret_cast<const v4i*>(p.w);
        v4i* dst = reinterpret_cast<v4i*>(sW);
        for (int i = tid; i < p.R * kStemK * 2; i += kStemBlock) dst[i] = src[i];
        if (tid < kStemK) sBiasI[tid] = tid < p.K ? (int)p.qbias[tid] : 0;
        for (int i = tid; i < kStemPatchWords; i += kStemBlock) sPatch[i] = 0u;    // the over-read margin stays zero
    }

    // this thread's patch pixels (the same for every tile)
    int prow[kStemPix], pcol[kStemPix];
    const int npix = p.PR * p.PC;
#pragma unroll
    for (int j = 0; j < kStemPix; ++j) {
        const int idx = tid + kStemBlock * j;
        prow[j] = idx < npix ? idx / p.PC : -1;
        pcol[j] = idx < npix ? idx - prow[j] * p.PC : 0;
    }
    const long plane = (long)p.H * p.W;

    // Fetch = loads only.  Nothing here may USE a loaded value (a select on it would make the compiler wait for each
    // load where it is issued) and nothing is conditional (hipcc waits vmcnt(0) at control-flow joins): a pixel outside
    // the image or a channel >= C reads element 0 of an existing plane, and `okbits` says at quantise time what to keep.
    float raw[kStemPix][kC];
    unsigned okbits = 0;
    int nx_tx = 0, nx_ty = 0;                      // tile coordinates of the fetched tile: reused by its store phase
    long nx_n = 0;
    auto fetch = [&](unsigned tile) {
        const int tx = (int)(tile % (unsigned)p.tiles_x);
        const unsigned t2 = tile / (unsigned)p.tiles_x;
        const int ty = (int)(t2 % (unsigned)p.tiles_y);
        const long n = (long)(t2 / (unsigned)p.tiles_y);
        nx_tx = tx; nx_ty = ty; nx_n = n;
        const int ih0 = ty * kStemTH * p.sh - p.ph, iw0 = tx * kStemTW * p.sw - p.pw;
        const float* __restrict__ img = p.x + n * kC * plane;
        okbits = 0;
#pragma unroll
        for (int j = 0; j < kStemPix; ++j) {
            const int ih = ih0 + prow[j], iw = iw0 + pcol[j];
            const bool ok = prow[j] >= 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            okbits |= ok ? 1u << j : 0u;
            const long off = ok ? (long)ih * p.W + iw : 0;
#pragma unroll
            for (int c = 0; c < kC; ++c) raw[j][c] = img[c * plane + off];
        }
    };

    // operand fragment addresses of this lane: pixel (2*wave + m/16, m%16) of the tile, m = lane & 31
    const int oy_l = 2 * wave + ((lane & 31) >> 4), ox_l = lane & 15;
    const int frag0 = oy_l * p.sh * p.PCS + ox_l * p.sw + 4 * half;        // word index for filter row 0
    const int wfrag0 = (lane & 31) * 32 + 16 * half;                       // byte offset inside one [64][32] weight row block

    // Which tiles this workgroup takes.  Workgroup g runs on XCD g % 8 (observed placement: for speed only), and each XCD has an L2
    // of its own: with tiles dealt g, g + grid, ... the neighbours of a tile -- whose input patches overlap its own by half (a 21 x 37
    // patch for 16 x 32 fresh pixels) -- ran on the other seven XCDs and every patch came over the fabric whole: 472 MB fetched for a
    // 154 MB image (rocprofv3 FETCH_SIZE, round 5).  Here an XCD owns a contiguous eighth of the tile sequence (x fastest, then y,
    // then the image) and its resident workgroups walk it side by side, so a patch's overlap with its neighbours' is an L2 hit.
    const unsigned G = gridDim.x;
    unsigned tile, tile_end, tile_step;
    if (p.xcd_chunk != 0 && (G & 7u) == 0) {
        const unsigned xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
        tile = xcd * p.xcd_chunk + local;
        tile_end = min((xcd + 1u) * p.xcd_chunk, p.ntiles);
        tile_step = G >> 3;
    } else {
        tile = blockIdx.x; tile_end = p.ntiles; tile_step = G;
    }
    if (tile < tile_end) fetch(tile);
    __syncthreads();
    for (; tile < tile_end; tile += tile_step) {
        const int tx = nx_tx, ty = nx_ty;               // of `tile` (set when it was fetched)
        const long n = nx_n;
        // a. quantise the fetched patch into LDS
#pragma unroll
        for (int j = 0; j < kStemPix; ++j) {
            unsigned word = 0;
#pragma unroll
            for (int c = 0; c < kC; ++c) word |= q8(raw[j][c], p.scale) << (8 * c);
            if (prow[j] >= 0) sPatch[prow[j] * p.PCS + pcol[j]] = (okbits >> j) & 1u ? word : 0u;
        }
        __syncthreads();
        // b. the next tile's loads fly under the matrix work (the last tile fetches itself again: no branch)
        const unsigned next = tile + tile_step;
        fetch(next < tile_end ? next : tile);

        // c. contraction: one MFMA per filter row and 32-channel block
        v16i acc[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0;
        for (int r = 0; r < p.R; ++r) {
            const unsigned* pf = &sPatch[frag0 + r * p.PCS];
            v4i fb;
            fb[0] = (int)pf[0]; fb[1] = (int)pf[1]; fb[2] = (int)pf[2]; fb[3] = (int)pf[3];
            const int8_t* wr = &sW[r * (kStemK * 32) + wfrag0];
            const v4i fa0 = *reinterpret_cast<const v4i*>(wr);
            const v4i fa1 = *reinterpret_cast<const v4i*>(wr + 32 * 32);
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa0, fb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa1, fb, acc[1], 0, 0, 0);
        }

        // d. tail on the accumulators (D row = channel (r&3) + 8*(r>>2) + 4*half, D column = this lane's pixel),
        //    bytes to LDS as [pixel][64 channels]
        int8_t* so = &sOut[wave][(lane & 31) * kStemOS];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kl = a * 32 + 8 * g + 4 * half + e;
                    v[e] = conv_tail_i(acc[a][4 * g + e], sBiasI[kl], p);
                }
                *reinterpret_cast<unsigned*>(&so[a * 32 + 8 * g + 4 * half]) = pack4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();          // the tile's bytes are in LDS; every wave is done reading the patch

        // e. 16-byte stores: 4 lanes cover the 64 channels of one pixel
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = lane + 64 * j;
            const int pix = idx >> 2, ch = (idx & 3) * 16;
            const int oy = ty * kStemTH + 2 * wave + (pix >> 4), ox = tx * kStemTW + (pix & 15);
            if (oy < p.P && ox < p.Q && ch < p.Kpad) {
                const v4i o = *reinterpret_cast<const v4i*>(&sOut[wave][pix * kStemOS + ch]);
                *reinterpret_cast<v4i*>(p.q + ((n * p.P + oy) * p.Q + ox) * p.Kpad + ch) = o;
            }
        }
    }
}

}  // namespace fq

// fp32 NCHW image -> int8 NHWC activations of the stem convolution (see include/fq.h)
extern "C" int fq_conv2d_i8_stem(const float* x_nchw, const int8_t* w_stem, const float* qbias, int8_t* q_nhwc, int Kpad, int relu,
                                 int N, int C, int H, int W, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                                 int pad_w, int ib, int rs, int ob, fq_stream_t stream) {
    using namespace fq;
    (void)ob;                                             // the output integers stand for q * 2^-ob; nothing to scale here
    if (N < 0 || C < 1 || C > 4 || H < 1 || W < 1 || K < 1 || K > kStemK || R < 1 || R > kStemMaxR || S < 1 || S > 8)
        return FQ_ERR_INVALID_ARG;
    if (stride_h < 1 || stride_w < 1 || pad_h < 0 || pad_w < 0) return FQ_ERR_INVALID_ARG;
    if (Kpad < K || (Kpad & 15) || Kpad > kStemK) return FQ_ERR_INVALID_ARG;
    if (rs < 1 || rs > 16) return FQ_ERR_INVALID_ARG;    // the integer tail; other shifts take the general kernels
    StemParams p;
    p.P = (H + 2 * pad_h - R) / stride_h + 1;
    p.Q = (W + 2 * pad_w - S) / stride_w + 1;
    if (p.P <= 0 || p.Q <= 0) return FQ_ERR_INVALID_ARG;
    p.PR = (kStemTH - 1) * stride_h + R;
    p.PC = (kStemTW - 1) * stride_w + S;
    p.PCS = (kStemTW - 1) * stride_w + 8;
    if (p.PCS < p.PC) p.PCS = p.PC;
    p.PCS |= 1;                                           // odd row stride: rows of a pixel block land in different banks
    // the last fragment read ends at word (PR-1)*PCS + (TW-1)*sw + 8
    if (p.PR * p.PC > kStemPix * kStemBlock || (p.PR - 1) * p.PCS + (kStemTW - 1) * stride_w + 8 > kStemPatchWords)
        return FQ_ERR_INVALID_ARG;
    if (N == 0) return FQ_OK;
    if (!x_nchw || !w_stem || !qbias || !q_nhwc) return FQ_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(q_nhwc) & 15u) || (reinterpret_cast<uintptr_t>(w_stem) & 15u)) return FQ_ERR_INVALID_ARG;
    p.x = x_nchw; p.w = w_stem; p.qbias = qbias; p.q = q_nhwc;
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.R = R; p.S = S; p.Kpad = Kpad;
    p.sh = stride_h; p.sw = stride_w; p.ph = pad_h; p.pw = pad_w;
    p.tiles_x = (p.Q + kStemTW - 1) / kStemTW;
    p.tiles_y = (p.P + kStemTH - 1) / kStemTH;
    const long ntiles = (long)N * p.tiles_x * p.tiles_y;
    if (ntiles > 0x7fffffffL) return FQ_ERR_INVALID_ARG;
    p.ntiles = (unsigned)ntiles;
    static const bool xcd_order = [] { const char* e = getenv("FQ_STEM_XCD"); return !(e && e[0] == '0'); }();
    p.xcd_chunk = xcd_order ? (unsigned)((ntiles + 7) / 8) : 0u;
    p.scale = ldexpf(1.0f, ib);
    p.rs = rs; p.half_rs = 1 << (rs - 1);
    p.ilo = -128; p.ihi = 127;
    p.slo = relu ? 0 : -128; p.shi = 127;
    static const int per_cu = [] { const char* e = getenv("FQ_STEM_WG_PER_CU"); return e ? atoi(e) : 0; }();
    long grid = (long)kCUs * (per_cu > 0 ? per_cu : 4);
    if (grid > ntiles) grid = ntiles;
    const dim3 g((unsigned)grid), b(kStemBlock);
    hipStream_t st = as_stream(stream);
    switch (C) {
        case 1: hipLaunchKernelGGL(stem_conv_i8_kernel<1>, g, b, 0, st, p); break;
        case 2: hipLaunchKernelGGL(stem_conv_i8_kernel<2>, g, b, 0, st, p); break;
        case 3: hipLaunchKernelGGL(stem_conv_i8_kernel<3>, g, b, 0, st, p); break;
        default: hipLaunchKernelGGL(stem_conv_i8_kernel<4>, g, b, 0, st, p); break;
    }
    note_conv_variant(kVarStem, 64);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
}
