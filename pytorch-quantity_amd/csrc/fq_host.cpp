// fq_host.cpp -- host-only entry points of libfq_hip.so: version/status, and the two scalar formulas
// of the reference that must run on the host libm to reproduce CPython's math.log(x, 2) bit for bit.
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

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
extern "C" int fq_conv2d_i8_last_variant(void) { return fq::g_last_conv_variant; }

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

// ---------------------------------------------------------------------------------------------
// Quantised-parameter JSON writer: byte-identical to Python's json.dump(nested_lists, fh, indent=N)
// (reference pytorch_quantizer.py:663-669, rewriter.py:57-59), written straight from the int32
// array.  HOST pointers.  Returns FQ_ERR_INVALID_ARG if the file cannot be opened / written.
// ---------------------------------------------------------------------------------------------
#include <cstdio>
#include <string>
#include <vector>

namespace {

struct JsonWriter {
    std::string buf;
    FILE* fh;
    bool ok = true;
    int indent;
    const int32_t* data;
    const int64_t* shape;
    int ndim;
    std::vector<std::string> pads;

    void flush() {
        if (!buf.empty()) {
            if (fwrite(buf.data(), 1, buf.size(), fh) != buf.size()) ok = false;
            buf.clear();
        }
    }
    static inline void put_int(std::string& s, int32_t v) {
        char tmp[12];
        int n = 0;
        uint32_t u = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
        do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) s.push_back('-');
        while (n) s.push_back(tmp[--n]);
    }
    // emits the list at nesting depth `level` starting at flat element offset `off`; returns elements consumed
    int64_t emit(int level, int64_t off) {
        const int64_t n = shape[level];
        if (n == 0) { buf += "[]"; return 0; }
        int64_t stride = 1;
        for (int d = level + 1; d < ndim; ++d) stride *= shape[d];
        buf += "[\n";
        const std::string& pad = pads[level + 1];
        for (int64_t i = 0; i < n; ++i) {
            buf += pad;
            if (level == ndim - 1) put_int(buf, data[off + i]);
            else emit(level + 1, off + i * stride);
            if (i + 1 < n) buf += ",\n";
            if (buf.size() > (1u << 22)) flush();
        }
        buf += "\n";
        buf += pads[level];
        buf += "]";
        return n * stride;
    }
};

}  // namespace

extern "C" int fq_json_dump_i32(const char* path, const int32_t* data, int ndim, const int64_t* shape, int indent) {
    if (!path || ndim < 0 || ndim > 16 || indent < 0 || indent > 64) return FQ_ERR_INVALID_ARG;
    if (ndim > 0 && !shape) return FQ_ERR_INVALID_ARG;
    int64_t total = 1;
    for (int d = 0; d < ndim; ++d) { if (shape[d] < 0) return FQ_ERR_INVALID_ARG; total *= shape[d]; }
    if (total > 0 && !data) return FQ_ERR_INVALID_ARG;
    FILE* fh = fopen(path, "wb");
    if (!fh) return FQ_ERR_INVALID_ARG;
    JsonWriter w;
    w.fh = fh; w.indent = indent; w.data = data; w.shape = shape; w.ndim = ndim;
    for (int d = 0; d <= ndim + 1; ++d) w.pads.emplace_back((size_t)(indent * d), ' ');
    w.buf.reserve(1u << 23);
    if (ndim == 0) JsonWriter::put_int(w.buf, data[0]);
    else w.emit(0, 0);
    w.flush();
    const bool ok = w.ok && fclose(fh) == 0;
    return ok ? FQ_OK : FQ_ERR_INVALID_ARG;
}

// ---- input side: PRE_PROCESS.IMG = 2 (pytorch_quantizer.py:276-280: np.load of one CHW image per calibration item) ----
// n .npy files that share one header (written by one np.save loop: same dtype / order / shape) read straight into
// consecutive slots of a caller-owned (pinned) host buffer: open, compare the header bytes, read the payload.  One call per
// batch from Python: the interpreter lock is released for all of it (the same loop in Python needs the lock three times per
// file, and loses it to the thread that is launching kernels).
static void read_npy_range(const char* const* paths, int lo, int hi, const void* header, size_t header_bytes, float* dst,
                           size_t elems, int* ok) {
    std::vector<char> head(header_bytes);
    for (int i = lo; i < hi; ++i) {
        ok[i] = 0;
        const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
        if (fd < 0) continue;
        bool good = header_bytes == 0 || ((size_t)pread(fd, head.data(), header_bytes, 0) == header_bytes &&
                                          std::memcmp(head.data(), header, header_bytes) == 0);
        char* out = reinterpret_cast<char*>(dst + (size_t)i * elems);
        size_t want = elems * sizeof(float), got = 0;
        while (good && got < want) {
            const ssize_t r = pread(fd, out + got, want - got, (off_t)(header_bytes + got));
            if (r <= 0) good = false;
            else got += (size_t)r;
        }
        if (good) {                                           // nothing may follow the payload
            char extra;
            good = pread(fd, &extra, 1, (off_t)(header_bytes + want)) == 0;
        }
        close(fd);
        ok[i] = good ? 1 : 0;
    }
}

extern "C" int fq_read_npy_batch_f32(const char* const* paths, int n, const void* header, size_t header_bytes, float* dst,
                                     size_t elems_per_file, int threads, int* ok_out) {
    if (n < 0 || (n && (!paths || !dst || !ok_out)) || (header_bytes && !header)) return FQ_ERR_INVALID_ARG;
    if (n == 0) return FQ_OK;
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    if (threads == 1) {
        read_npy_range(paths, 0, n, header, header_bytes, dst, elems_per_file, ok_out);
        return FQ_OK;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        const int lo = (int)((long)n * t / threads), hi = (int)((long)n * (t + 1) / threads);
        pool.emplace_back(read_npy_range, paths, lo, hi, header, header_bytes, dst, elems_per_file, ok_out);
    }
    for (auto& th : pool) th.join();
    return FQ_OK;
}
