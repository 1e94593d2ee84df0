// Exhaustive check (GPU box) of the 3-instruction quotient used by the fast histogram path:
//     y = RN(1/b);  q0 = RN(a*y);  r = fma(-q0, b, a);  q1 = fma(r, y, q0)
// against the IEEE correctly rounded a/b, for ALL 2^23 x 2^23 significand pairs a, b in [1, 2).
// Absent overflow / underflow every fp32 division is an exact power-of-two scaling of one of these
// pairs, so this covers every normal-range case.  Prints the number of pairs where q1 != a/b and,
// of those, where trunc() of the quotient would differ for some quotient exponent 0..10 (the only
// thing the histogram consumes).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o verify_fastdiv verify_fastdiv.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void check(uint32_t a_begin, uint32_t a_count, unsigned long long* counters, uint32_t* examples) {
    const uint32_t mb = blockIdx.x * blockDim.x + threadIdx.x;          // all 2^23 divisors
    const float b = __uint_as_float(0x3f800000u | mb);
    const float y = 1.0f / b;                                           // IEEE
    unsigned long long bad = 0, bad_trunc = 0;
    for (uint32_t i = 0; i < a_count; ++i) {
        const float a = __uint_as_float(0x3f800000u | (a_begin + i));
        const float ref = a / b;
        const float q0 = a * y;
        const float r = __builtin_fmaf(-q0, b, a);
        const float q1 = __builtin_fmaf(r, y, q0);
        if (q1 != ref) {
            ++bad;
            // quotient in (0.5, 2): compare integer parts after scaling by 2^e, e = 0..11
            bool t = false;
            for (int e = 0; e <= 11; ++e) {
                const float s = (float)(1u << e);
                t |= (int)(ref * s) != (int)(q1 * s);
                t |= (int)(ref * s * 2.0f) != (int)(q1 * s * 2.0f);
            }
            if (t) {
                ++bad_trunc;
                unsigned long long slot = atomicAdd(&counters[2], 1ULL);
                if (slot < 16) { examples[2 * slot] = __float_as_uint(a); examples[2 * slot + 1] = __float_as_uint(b); }
            }
        }
    }
    if (bad) atomicAdd(&counters[0], bad);
    if (bad_trunc) atomicAdd(&counters[1], bad_trunc);
}

int main(int argc, char** argv) {
    // optional: fraction of the a-range to sweep (1 = exhaustive 2^46 pairs)
    const uint32_t total_a = 1u << 23;
    uint32_t step = 1u << 15;
    uint32_t limit = argc > 1 ? (uint32_t)atol(argv[1]) : total_a;
    unsigned long long* d_cnt; uint32_t* d_ex;
    hipMalloc(&d_cnt, 3 * sizeof(unsigned long long)); hipMemset(d_cnt, 0, 3 * sizeof(unsigned long long));
    hipMalloc(&d_ex, 32 * sizeof(uint32_t)); hipMemset(d_ex, 0, 32 * sizeof(uint32_t));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    for (uint32_t a0 = 0; a0 < limit; a0 += step) {
        hipLaunchKernelGGL(check, dim3((1u << 23) / 256), dim3(256), 0, 0, a0, step, d_cnt, d_ex);
        if ((a0 / step) % 32 == 31) hipDeviceSynchronize();
    }
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3]; uint32_t ex[32];
    hipMemcpy(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(ex, d_ex, sizeof(ex), hipMemcpyDeviceToHost);
    printf("pairs checked: %llu x 8388608 ; q1 != a/b: %llu ; trunc would differ: %llu ; %.1f s\n",
           (unsigned long long)limit, h[0], h[1], ms / 1e3);
    for (int i = 0; i < 16 && (unsigned long long)i < h[2]; ++i) printf("  example a=0x%08x b=0x%08x\n", ex[2 * i], ex[2 * i + 1]);
    return 0;
}
