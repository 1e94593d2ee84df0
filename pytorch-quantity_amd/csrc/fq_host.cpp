// fq_host.cpp -- host-only entry points of libfq_hip.so: version/status, and the two scalar formulas
// of the reference that must run on the host libm to reproduce CPython's math.log(x, 2) bit for bit.
#include <cmath>

#include "fq_common.h"

extern "C" int fq_version(void) { return FQ_VERSION; }

extern "C" const char* fq_status_string(int status) {
    switch (status) {
        case FQ_OK: return "FQ_OK";
        case FQ_ERR_INVALID_ARG: return "FQ_ERR_INVALID_ARG";
        case FQ_ERR_HIP: return "FQ_ERR_HIP (see fq_last_hip_error)";
        case FQ_ERR_WORKSPACE: return "FQ_ERR_WORKSPACE";
        case FQ_ERR_UNSUPPORTED: return "FQ_ERR_UNSUPPORTED";
        default: return "FQ_ERR_UNKNOWN";
    }
}

extern "C" int fq_last_hip_error(void) { return fq::g_last_hip_error; }

// quantizer.py:86-90.  (threshold_bin + 0.5) is a Python float, interval an np.float32: the product
// is fp32.  math.log(x, 2) is log(x)/log(2) in float64 on the C library's log.
extern "C" int fq_bits_from_threshold(const int32_t* thr, const float* interval, int rows, int32_t* bits_out,
                                      float* thr_val_out) {
    if (rows < 0) return FQ_ERR_INVALID_ARG;
    if (rows && (!thr || !interval || !bits_out)) return FQ_ERR_INVALID_ARG;
    for (int r = 0; r < rows; ++r) {
        volatile float tv = ((float)thr[r] + 0.5f) * interval[r];
        if (thr_val_out) thr_val_out[r] = tv;
        const double l = std::log((double)tv) / std::log(2.0);
        bits_out[r] = (int32_t)(8 - 1 - std::ceil(l));
    }
    return FQ_OK;
}

// pytorch_quantizer.py:651-653
extern "C" int fq_bits_from_absmax(const float* absmax, int n, int32_t* bits_out) {
    if (n < 0) return FQ_ERR_INVALID_ARG;
    if (n && (!absmax || !bits_out)) return FQ_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) {
        if (!(absmax[i] > 0.0f)) return FQ_ERR_INVALID_ARG;      // the reference raises (log of 0)
        const double l = std::log((double)absmax[i]) / std::log(2.0);
        bits_out[i] = (int32_t)(8 - 1 - std::ceil(l));
    }
    return FQ_OK;
}
