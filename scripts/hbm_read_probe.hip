// GPU probe: what read-only rate does HBM give a streaming reduction on this box?  (ceiling for abs-max / histogram)
// Reads one 8 GiB buffer with 16-byte loads and a trivial reduction; sweeps workgroups per CU, loads in flight per
// lane, contiguous-region-per-workgroup vs. interleaved 16 KB chunks, plain vs. non-temporal loads.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/hbm_read_probe scripts/hbm_read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));

template <int U, bool kInterleave, bool kNT>
__global__ __launch_bounds__(256) void read_kernel(const f4v* __restrict__ x, size_t nvec, float* sink) {
    float m = 0.f;
    const size_t chunk = (size_t)U * 256;
    size_t base, step, end;
    if (kInterleave) { base = blockIdx.x * chunk; step = (size_t)gridDim.x * chunk; end = nvec; }
    else { const size_t per = (nvec / gridDim.x) / chunk * chunk; base = blockIdx.x * per; step = chunk; end = base + per; }
    for (; base + chunk <= end; base += step) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = kNT ? __builtin_nontemporal_load(&x[base + threadIdx.x + u * 256]) : x[base + threadIdx.x + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
    }
    if (m == 12345.678f) sink[0] = m;
}

__global__ void fill_kernel(unsigned* x, size_t n) {      // cheap hash: values differ from word to word
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = (h & 0x807fffffu) | 0x3f000000u;          // +-[0.5, 1)
    }
}

template <int U, bool I, bool NT>
static void run(const f4v* x, size_t nvec, float* sink, int per_cu) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * per_cu;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((read_kernel<U, I, NT>), dim3(grid), dim3(256), 0, 0, x, nvec, sink);
    float best = 1e9, tot = 0;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a); hipLaunchKernelGGL((read_kernel<U, I, NT>), dim3(grid), dim3(256), 0, 0, x, nvec, sink); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); tot += ms; if (ms < best) best = ms;
    }
    printf("U=%d %-11s %-3s wg/cu=%-3d  mean %.3f ms = %.0f GB/s (best %.0f)\n", U, I ? "interleaved" : "contiguous", NT ? "nt" : "-",
           per_cu, tot / 5, nvec * 16.0 / (tot / 5) / 1e6, nvec * 16.0 / best / 1e6);
}

int main() {
    const size_t bytes = (size_t)8 << 30, nvec = bytes / 16;
    f4v* x; float* sink;
    hipMalloc(&x, bytes); hipMalloc(&sink, 4);
    hipMemset(x, 0x3c, bytes);
    const bool random_fill = getenv("PROBE_RANDOM") != nullptr;
    if (random_fill) hipLaunchKernelGGL(fill_kernel, dim3(65536), dim3(256), 0, 0, (unsigned*)x, bytes / 4);
    printf("fill: %s\n", random_fill ? "hashed values" : "constant bytes");
    for (int per_cu : {4, 16}) {
        run<4, false, false>(x, nvec, sink, per_cu);
        run<4, true, false>(x, nvec, sink, per_cu);
        run<4, true, true>(x, nvec, sink, per_cu);
        run<4, false, true>(x, nvec, sink, per_cu);
        run<8, false, false>(x, nvec, sink, per_cu);
        run<8, true, false>(x, nvec, sink, per_cu);
        run<2, true, false>(x, nvec, sink, per_cu);
        run<2, true, true>(x, nvec, sink, per_cu);
    }
    return 0;
}
