// GPU probe: what does a GB of FRESH device memory cost a cold process, by API?
//   hipMalloc                        -- what torch's caching allocator calls when its pool has to grow (28 ms/GB)
//   hipMallocAsync                   -- the stream-ordered pool, release threshold raised so that frees stay in the pool
//   hipMemAddressReserve + hipMemCreate / hipMemMap / hipMemSetAccess in granules -- the virtual-memory API
// Each figure is ms per GB for the allocation call(s) alone and for allocation + first touch (a kernel writing one dword per
// 4 KB page), and -- for the pool -- what a SECOND allocation of the same size costs once the first was freed into the pool.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/_bin/alloc_probe scripts/alloc_probe.hip     Run: scripts/_bin/alloc_probe [GB]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);               \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

__global__ void touch(unsigned* p, size_t pages) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < pages; i += stride) p[i * 1024] = 1u;          // one dword per 4 KB
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int touch_all(void* p, size_t bytes) {
    hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, static_cast<unsigned*>(p), bytes / 4096);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    return 0;
}

static bool want(const char* mode, const char* name) { return !mode || !strcmp(mode, "all") || !strcmp(mode, name); }

int main(int argc, char** argv) {
    // usage: alloc_probe [GB] [all | malloc | async | vmm]: ONE API per process is the clean measurement -- the driver hands
    // memory an earlier part of the SAME process has freed back much faster than fresh memory (run with `all`, the 1 GB VMM
    // pieces right after the 1 GB hipMalloc chunks cost 0.02 ms/GB; in a process of their own, 60)
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 16;
    const char* mode = argc > 2 ? argv[2] : nullptr;
    const size_t bytes = gb << 30;
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    printf("device memory: %.1f GB free of %.1f GB; probe size %zu GB\n", fr / 1073741824.0, tot / 1073741824.0, gb);

    // ---- hipMalloc, in chunks of 1 GB and as one block
    for (int form = 0; form < 2 && want(mode, "malloc"); ++form) {
        const size_t chunk = form == 0 ? ((size_t)1 << 30) : bytes;
        std::vector<void*> ps;
        CK(hipDeviceSynchronize());
        double t0 = now_ms();
        for (size_t off = 0; off < bytes; off += chunk) {
            void* p = nullptr;
            CK(hipMalloc(&p, chunk));
            ps.push_back(p);
        }
        CK(hipDeviceSynchronize());
        double t1 = now_ms();
        for (void* p : ps)
            if (touch_all(p, chunk)) return 1;
        double t2 = now_ms();
        for (void* p : ps) CK(hipFree(p));
        CK(hipDeviceSynchronize());
        double t3 = now_ms();
        printf("hipMalloc %-22s: alloc %7.2f ms/GB   alloc + first touch %7.2f ms/GB   free %6.2f ms/GB\n",
               form == 0 ? "(1 GB chunks)" : "(one block)", (t1 - t0) / gb, (t2 - t0) / gb, (t3 - t2) / gb);
    }

    // ---- hipMallocAsync: default pool, release threshold = everything
    if (want(mode, "async")) {
        hipMemPool_t pool;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = UINT64_MAX;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        hipStream_t st;
        CK(hipStreamCreate(&st));
        for (int round = 0; round < 2; ++round) {
            std::vector<void*> ps;
            CK(hipDeviceSynchronize());
            double t0 = now_ms();
            for (size_t off = 0; off < bytes; off += (size_t)1 << 30) {
                void* p = nullptr;
                CK(hipMallocAsync(&p, (size_t)1 << 30, st));
                ps.push_back(p);
            }
            CK(hipStreamSynchronize(st));
            double t1 = now_ms();
            for (void* p : ps)
                if (touch_all(p, (size_t)1 << 30)) return 1;
            double t2 = now_ms();
            for (void* p : ps) CK(hipFreeAsync(p, st));
            CK(hipStreamSynchronize(st));
            double t3 = now_ms();
            printf("hipMallocAsync %-17s: alloc %7.2f ms/GB   alloc + first touch %7.2f ms/GB   free %6.2f ms/GB\n",
                   round == 0 ? "(pool grows)" : "(from the pool)", (t1 - t0) / gb, (t2 - t0) / gb, (t3 - t2) / gb);
        }
        uint64_t reserved = 0;
        CK(hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &reserved));
        printf("  pool holds %.1f GB after the frees\n", reserved / 1073741824.0);
        CK(hipMemPoolTrimTo(pool, 0));
        CK(hipStreamDestroy(st));
    }

    // ---- the virtual-memory API: one reservation, physical granules mapped into it
    if (want(mode, "vmm")) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("VMM: recommended granularity %zu KB\n", gran >> 10);
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        for (size_t piece_mb : {(size_t)64, (size_t)1024, (size_t)2048}) {
            const size_t piece = piece_mb << 20;
            if (piece % gran) continue;
            void* va = nullptr;
            CK(hipDeviceSynchronize());
            double t0 = now_ms();
            CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
            std::vector<hipMemGenericAllocationHandle_t> hs;
            double t_create = 0, t_map = 0, t_acc = 0;
            for (size_t off = 0; off < bytes; off += piece) {
                hipMemGenericAllocationHandle_t h;
                double a = now_ms();
                CK(hipMemCreate(&h, piece, &prop, 0));
                double b = now_ms();
                CK(hipMemMap(static_cast<char*>(va) + off, piece, 0, h, 0));
                double c = now_ms();
                CK(hipMemSetAccess(static_cast<char*>(va) + off, piece, &acc, 1));
                double d = now_ms();
                t_create += b - a; t_map += c - b; t_acc += d - c;
                hs.push_back(h);
            }
            CK(hipDeviceSynchronize());
            double t1 = now_ms();
            if (touch_all(va, bytes)) return 1;
            double t2 = now_ms();
            for (size_t i = 0; i < hs.size(); ++i) {
                CK(hipMemUnmap(static_cast<char*>(va) + i * piece, piece));
                CK(hipMemRelease(hs[i]));
            }
            CK(hipMemAddressFree(va, bytes));
            CK(hipDeviceSynchronize());
            double t3 = now_ms();
            printf("VMM %5zu MB pieces          : alloc %7.2f ms/GB (create %.2f, map %.2f, set access %.2f)   alloc + first touch %7.2f "
                   "ms/GB   unmap + release %6.2f ms/GB\n",
                   piece_mb, (t1 - t0) / gb, t_create / gb, t_map / gb, t_acc / gb, (t2 - t0) / gb, (t3 - t2) / gb);
        }
    }
    return 0;
}
