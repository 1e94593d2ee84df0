// GPU probe for DESIGN.md section 6c ("split-bf16 matrix products"): how accurate is an fp32 GEMM whose operands are split into
// three bf16 values each and multiplied on v_mfma_f32_32x32x16_bf16, and what does that instruction deliver?
//   accuracy: one wave, a 32 x 32 tile, K = 64 .. 4096, operands uniform in (-1, 1): max over the tile of |result - fp64| / sum |a||b|
//             for  fp32 MFMA (v_mfma_f32_32x32x2_f32: the fma chain the product's kernels use),  bf16 x 3 / x 6 / x 9 products
//   rate:     independent MFMAs, register operands, 1..4 waves per SIMD
// hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/bf16x3_probe scripts/bf16x3_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef short s8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float v, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    hi = bf16_rne(v);
    const float r1 = v - bf16_f32(hi);
    mid = bf16_rne(r1);
    const float r2 = r1 - bf16_f32(mid);
    lo = bf16_rne(r2);
}

// A: [32][K] row major, B: [K][32]; out[method][32][32];  methods: 0 fp32 MFMA, 1 bf16x3, 2 bf16x6, 3 bf16x9
__global__ __launch_bounds__(64) void accuracy_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f16v c32, c3, c6, c9;
    for (int e = 0; e < 16; ++e) { c32[e] = 0.f; c3[e] = 0.f; c6[e] = 0.f; c9[e] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 2) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], B[(k0 + h) * 32 + r], c32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        s8v a[3], b[3];
        for (int i = 0; i < 8; ++i) {
            unsigned short x0, x1, x2;
            split3(A[r * K + k0 + 8 * h + i], x0, x1, x2);
            a[0][i] = (short)x0; a[1][i] = (short)x1; a[2][i] = (short)x2;
            split3(B[(k0 + 8 * h + i) * 32 + r], x0, x1, x2);
            b[0][i] = (short)x0; b[1][i] = (short)x1; b[2][i] = (short)x2;
        }
#define MM(acc, i, j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a[i]), __builtin_bit_cast(bf8v, b[j]), acc, 0, 0, 0)
        // smallest terms first
        MM(c9, 2, 2); MM(c9, 1, 2); MM(c9, 2, 1);
        MM(c9, 0, 2); MM(c9, 2, 0); MM(c9, 1, 1); MM(c9, 0, 1); MM(c9, 1, 0); MM(c9, 0, 0);
        MM(c6, 0, 2); MM(c6, 2, 0); MM(c6, 1, 1); MM(c6, 0, 1); MM(c6, 1, 0); MM(c6, 0, 0);
        MM(c3, 0, 1); MM(c3, 1, 0); MM(c3, 0, 0);
#undef MM
    }
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        out[0 * 1024 + row * 32 + r] = c32[e];
        out[1 * 1024 + row * 32 + r] = c3[e];
        out[2 * 1024 + row * 32 + r] = c6[e];
        out[3 * 1024 + row * 32 + r] = c9[e];
    }
}

template <int MODE>   // 0: fp32 32x32x2, 1: bf16 32x32x16
__global__ __launch_bounds__(256) void rate_kernel(float* sink, int iters, float seed) {
    f16v c0, c1, c2, c3;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
    s8v a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    const float fa = seed, fb = seed * 3;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
void rate(int per_cu, float* sink) {
    const int iters = 4096, grid = 256 * per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, sink, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, sink, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 4 * (MODE == 0 ? 4096.0 : 32768.0);     // waves x MFMAs x 2 * 32 * 32 * k
    printf("%s  %d waves per SIMD: %.3f ms  %.1f TFLOP/s\n", MODE == 0 ? "v_mfma_f32_32x32x2_f32  " : "v_mfma_f32_32x32x16_bf16", per_cu, ms, flop / ms / 1e9);
}

int main() {
    float* sink; hipMalloc(&sink, 4);
    for (int K = 64; K <= 4096; K *= 4) {
        std::vector<float> A(32 * K), B(K * 32);
        srand(7 + K);
        for (auto& v : A) v = 2.0f * rand() / RAND_MAX - 1.0f;
        for (auto& v : B) v = 2.0f * rand() / RAND_MAX - 1.0f;
        float *dA, *dB, *dO;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 4 * 1024 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dO, K);
        std::vector<float> O(4 * 1024);
        hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
        double worst[4] = {0, 0, 0, 0};
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0, bound = 0;
                for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; bound += fabs((double)A[i * K + k] * B[k * 32 + j]); }
                for (int m = 0; m < 4; ++m) worst[m] = fmax(worst[m], fabs(O[m * 1024 + i * 32 + j] - ref) / bound);
            }
        printf("K = %4d   max |err| / sum|a||b|:  fp32 MFMA %.2e   bf16 x3 %.2e   x6 %.2e   x9 %.2e\n", K, worst[0], worst[1], worst[2], worst[3]);
        hipFree(dA); hipFree(dB); hipFree(dO);
    }
    for (int w = 1; w <= 4; ++w) rate<0>(w, sink);
    for (int w = 1; w <= 4; ++w) rate<1>(w, sink);
    return 0;
}
