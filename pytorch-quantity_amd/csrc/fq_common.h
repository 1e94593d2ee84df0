// fq_common.h -- shared host-side plumbing for libfq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/fq.h"

namespace fq {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return FQ_ERR_HIP;
}

#define FQ_HIP_CHECK(expr)                         \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) return ::fq::hip_fail(e__); \
    } while (0)

// Launch-error check that does not synchronise.
#define FQ_LAUNCH_CHECK()                          \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return ::fq::hip_fail(e__); \
    } while (0)

inline hipStream_t as_stream(fq_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kCUs = 256;          // MI355X

inline bool valid_bitwidth(int bw) { return bw == 8 || bw == 16; }

}  // namespace fq
