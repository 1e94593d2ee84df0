// fq_common.h -- shared host-side plumbing for libfq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/fq.h"

namespace fq {

extern thread_local int g_last_hip_error;
// Which kernel the last fq_conv2d_i8* call of this thread launched (fq_conv2d_i8_last_variant, include/fq.h): a diagnostic
// of the same kind as the error code above -- thread-local, written by the dispatch, read by tests and bench.py so that
// "the kernels the bench times were the ones the golden check ran" is an assertion and not a belief.
extern thread_local int g_last_conv_variant;
enum ConvVariant { kVarNone = 0, kVarC64Halo = 1, kVarStream = 2, kVarHalo8 = 3, kVarHalo = 4, kVarDma2 = 5, kVarDma3 = 6,
                   kVarTileC128 = 7, kVarTileC64 = 8, kVarTileGeneral = 9, kVarStem = 10, kVarLinearWave = 13 };
inline void note_conv_variant(int variant, int tk) { g_last_conv_variant = variant | (tk << 8); }

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

// More than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize, and that attribute is PER DEVICE: a
// process that drives a second GPU must opt in there too (a function-local "done once" flag, what round 3 had, made the launch
// on the second device fail after the dispatch had already committed to the kernel).  One flag per kernel AND device; the
// call is cheap, a racing second thread merely repeats it.
constexpr int kMaxDevices = 64;
inline bool ensure_dynamic_lds(const void* kernel, int bytes, bool (&done)[kMaxDevices]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return false;
    if (dev < kMaxDevices && done[dev]) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (dev < kMaxDevices) done[dev] = true;
    return true;
}

}  // namespace fq
