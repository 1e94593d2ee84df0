// Probe (GPU box): operand lane maps of v_mfma_i32_32x32x32_i8 and v_mfma_i32_16x16x64_i8 on gfx950,
// checked with exact integer data (asymmetric A and B).  hipcc --offload-arch=gfx950 -o probe mfma_i8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// hypothesis h: 0 = lane holds 16 consecutive k (k = 16*(l>>5)+j); 1 = two 8-runs (k = 8*(l>>5)+j, 16+8*(l>>5)+j-8)
__device__ int kmap32(int h, int l, int j) {
    if (h == 0) return 16 * (l >> 5) + j;
    return (j < 8) ? 8 * (l >> 5) + j : 16 + 8 * (l >> 5) + (j - 8);
}
__global__ void k32(const int8_t* A, const int8_t* B, int* D, int h) {   // A[32][32] row-major (i,k), B[32][32] (k,j)
    int l = threadIdx.x;
    int8_t a[16], b[16];
    for (int j = 0; j < 16; ++j) { int k = kmap32(h, l, j); a[j] = A[(l & 31) * 32 + k]; b[j] = B[k * 32 + (l & 31)]; }
    v4i av, bv; memcpy(&av, a, 16); memcpy(&bv, b, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); int col = l & 31; D[row * 32 + col] = c[r]; }
}
__device__ int kmap16(int h, int l, int j) {          // 16x16x64: 4 lane groups of 16
    if (h == 0) return 16 * (l >> 4) + j;
    return (j < 8) ? 8 * (l >> 4) + j : 32 + 8 * (l >> 4) + (j - 8);
}
__global__ void k16(const int8_t* A, const int8_t* B, int* D, int h) {   // A[16][64], B[64][16]
    int l = threadIdx.x;
    int8_t a[16], b[16];
    for (int j = 0; j < 16; ++j) { int k = kmap16(h, l, j); a[j] = A[(l & 15) * 64 + k]; b[j] = B[k * 16 + (l & 15)]; }
    v4i av, bv; memcpy(&av, a, 16); memcpy(&bv, b, 16);
    typedef int v4 __attribute__((ext_vector_type(4)));
    v4 c = {0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { int row = (l >> 4) * 4 + r; int col = l & 15; D[row * 16 + col] = c[r]; }
}
int main() {
    int8_t hA[2048], hB[2048]; srand(1);
    for (int i = 0; i < 2048; ++i) { hA[i] = (int8_t)(rand() % 255 - 127); hB[i] = (int8_t)(rand() % 255 - 127); }
    int8_t *dA, *dB; int* dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    int hD[1024];
    for (int h = 0; h < 2; ++h) {
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD, h); hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 32 + j]; bad += s != hD[i * 32 + j]; }
        printf("mfma_i32_32x32x32_i8 hypothesis %d: %d mismatches\n", h, bad);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, h); hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int s = 0; for (int k = 0; k < 64; ++k) s += hA[i * 64 + k] * hB[k * 16 + j]; bad += s != hD[i * 16 + j]; }
        printf("mfma_i32_16x16x64_i8 hypothesis %d: %d mismatches\n", h, bad);
    }
    return 0;
}
