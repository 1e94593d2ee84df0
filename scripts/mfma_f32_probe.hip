// GPU probe: what share of the fp32 matrix cycles does a wave's instruction stream reach?  (ceiling for fq_conv1x1_f32)
//   mode 0: 4 independent v_mfma_f32_32x32x2_f32 per iteration, register operands only
//   mode 1: + the 1x1 kernel's LDS traffic: 2 x ds_read2_b32 per 4 MFMAs, operands one group ahead (no barriers, no global loads)
//   mode 2: mode 1 + a workgroup barrier every 4 groups (K step of 8)
// swept over 1..4 workgroups (of 4 waves) per CU.   hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/mfma_f32_probe scripts/mfma_f32_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void mfma_kernel(float* sink, int iters, float seed) {
    __shared__ float lds[3 * 8 * 256];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, r = lane & 31u, h = lane >> 5;
    for (int i = threadIdx.x; i < 3 * 8 * 256; i += 256) lds[i] = seed + (float)i * 1e-9f;
    __syncthreads();
    f16v c00, c01, c10, c11;
    for (int e = 0; e < 16; ++e) { c00[e] = 0.f; c01[e] = 0.f; c10[e] = 0.f; c11[e] = 0.f; }
    const float* wrow = lds + h * 128 + (wave >> 1) * 64 + r;
    const float* xrow = lds + 3 * 8 * 128 + h * 128 + (wave & 1) * 64 + r;
    float a0 = seed, a1 = seed * 2, b0 = seed * 3, b1 = seed * 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
            if (MODE >= 1) {
                const int off = ((it + g) & 3) * 2 * 128;
                na0 = wrow[off]; na1 = wrow[off + 32]; nb0 = xrow[off]; nb1 = xrow[off + 32];
                __builtin_amdgcn_sched_barrier(0);
            }
            c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c00[e] + c01[e] + c10[e] + c11[e];
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
void run(int per_cu, float* sink) {
    const int iters = 4096, grid = 256 * per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(mfma_kernel<MODE>, dim3(grid), dim3(256), 0, 0, sink, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(mfma_kernel<MODE>, dim3(grid), dim3(256), 0, 0, sink, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 * iters * 16 * 4096.0;          // waves x MFMAs x 2*32*32*2
    printf("mode %d  %d workgroups per CU (%d waves per SIMD): %.3f ms  %.1f TFLOP/s = %.1f %% of 157.3\n", MODE, per_cu, per_cu, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main() {
    float* sink; hipMalloc(&sink, 4);
    for (int w = 1; w <= 4; ++w) run<0>(w, sink);
    for (int w = 1; w <= 4; ++w) run<1>(w, sink);
    for (int w = 1; w <= 4; ++w) run<2>(w, sink);
    return 0;
}
