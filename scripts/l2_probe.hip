// GPU probe: how many bytes per clock can one CU pull from L2 with 16-byte-per-lane loads?
// Each workgroup (256 threads) loops over a small, L2-resident buffer; variants: plain global_load_dwordx4
// into registers, buffer_load_dwordx4 ... lds (LDS-DMA).  Prints GB/s chip-wide and B/clk/CU (2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int rsrc_words __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void reg_loads(const v4i* __restrict__ buf, size_t nvec_per_wg, int iters, int* sink) {
    const v4i* p = buf + (size_t)(blockIdx.x % 8) * nvec_per_wg;       // 8 distinct windows: L2 resident, larger than L1
    v4i acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (size_t i = threadIdx.x; i < nvec_per_wg; i += 256) {
            const v4i v = p[i];
            acc ^= v;
        }
    }
    if (acc[0] == 0x12345678) sink[0] = acc[1];
}

__global__ __launch_bounds__(256) void dma_loads(const char* __restrict__ buf, unsigned bytes_total, unsigned bytes_per_wg, int iters, int* sink) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const unsigned long long a = (unsigned long long)buf;
    rsrc_words r = {(int)(unsigned)(a & 0xffffffffu), (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes_total, 0x00020000};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const unsigned base = (blockIdx.x % 8) * bytes_per_wg;
    for (int it = 0; it < iters; ++it) {
        for (unsigned off = 0; off < bytes_per_wg; off += 4096) {     // 4 waves x 1 KB per instruction
            const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)&lds[(off & 32767u) / 4096 * 4096 + wave * 1024];
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                         :: "s"(lds_base), "v"(base + off + wave * 1024 + lane * 16), "s"(r) : "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (lds[threadIdx.x] == 0x7f && sink) sink[1] = 1;
}

int main(int argc, char** argv) {
    const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 2;
    const int iters = 40;
    const size_t per_wg = argc > 2 ? (size_t)atoi(argv[2]) * 1024 : 262144;   // bytes each workgroup sweeps per iteration (> 32 KB L1)
    char* buf; int* sink;
    hipMalloc(&buf, 8 * per_wg); hipMemset(buf, 1, 8 * per_wg); hipMalloc(&sink, 64);
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(reg_loads, dim3(grid), dim3(256), 0, 0, (const v4i*)buf, per_wg / 16, iters, sink);
            else hipLaunchKernelGGL(dma_loads, dim3(grid), dim3(256), 0, 0, buf, (unsigned)(8 * per_wg), (unsigned)per_wg, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double bytes = (double)grid * per_wg * iters;
        printf("%s, %d WG/CU: %.3f ms  %.1f TB/s  %.1f B/clk/CU @2.4GHz\n", variant == 0 ? "global_load_dwordx4 -> VGPR" : "buffer_load_dwordx4 -> LDS ",
               wgs_per_cu, best, bytes / best / 1e9, bytes / (best * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
