// Does a large pageable-host -> device hipMemcpyAsync into stream-ordered (hipMallocAsync) memory stay ordered with the
// hipFreeAsync / hipMallocAsync calls around it?  Mimics fs_upload_orbit twice in a row with GB-sized buffers.
// Build: hipcc --offload-arch=gfx950 -O2 -o async_alloc_probe async_alloc_probe.hip ; usage: ./async_alloc_probe [MiB] [sync_alloc]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_prepare(const uint4 *in, uint4 *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint4 v = in[i]; out[i] = make_uint4(v.x + 1u, v.y, v.z, v.w ^ 0x55u); }
}
__global__ void k_check(const uint4 *out, size_t n, unsigned seed, unsigned long long *bad)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned x = (unsigned)i * 2654435761u + seed;
        const uint4 v = out[i];
        if (v.x != x + 1u || v.y != (unsigned)(i >> 3) || v.w != (seed ^ 0x55u)) {
            if (atomicAdd(bad, 1ull) == 0ull) {
                bad[1] = i;
                bad[2] = ((unsigned long long)v.w << 32) | v.x;
                bad[3] = ((unsigned long long)v.z << 32) | v.y;
            }
        }
    }
}
int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 1300;
    const int mode_early = argc > 2 ? atoi(argv[2]) : 0;
    const int mode = mode_early; // 1: hipMalloc/hipFree; 2: page-locked source; 4: synchronise after the frees; 8: no third buffer; 16: release threshold = max; 32: blocking stream
    const bool sync_alloc = (mode & 1) != 0;
    const size_t n = mib * (1u << 20) / 16;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, (mode_early & 32) ? hipStreamDefault : hipStreamNonBlocking));
    if (mode & 16) { // keep freed memory in the pool (no trimming at synchronisations)
        hipMemPool_t pool = nullptr;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t keep = ~0ull;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    }
    std::vector<uint4> host_v((mode & 2) ? 1 : n);
    uint4 *host_p = host_v.data();
    if (mode & 2)
        CK(hipHostMalloc((void **)&host_p, n * 16, hipHostMallocDefault));
    struct { uint4 *p; uint4 *data() { return p; } uint4 &operator[](size_t i) { return p[i]; } } host{host_p};
    unsigned long long *bad;
    CK(hipMalloc(&bad, 32));
    uint4 *raw = nullptr, *out = nullptr, *aux = nullptr;
    for (int round = 0; round < 4; round++) {
        const unsigned seed = 1000u + (unsigned)round;
        for (size_t i = 0; i < n; i++)
            host[i] = make_uint4((unsigned)i * 2654435761u + seed, (unsigned)(i >> 3), 0u, seed);
        if (out) { CK(sync_alloc ? hipFree(out) : hipFreeAsync(out, s)); out = nullptr; }
        if (aux) { CK(sync_alloc ? hipFree(aux) : hipFreeAsync(aux, s)); aux = nullptr; }
        if (mode & 4)
            CK(hipStreamSynchronize(s));
        CK(sync_alloc ? hipMalloc(&raw, n * 16) : hipMallocAsync((void **)&raw, n * 16, s));
        CK(sync_alloc ? hipMalloc(&out, n * 16) : hipMallocAsync((void **)&out, n * 16, s));
        CK(hipMemcpyAsync(raw, host.data(), n * 16, hipMemcpyDefault, s));
        hipLaunchKernelGGL(k_prepare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw, out, n);
        if (!(mode & 8)) {
            CK(sync_alloc ? hipMalloc(&aux, 2 * n * 16) : hipMallocAsync((void **)&aux, 2 * n * 16, s));
            CK(hipMemsetAsync(aux, 0, 2 * n * 16, s));
        }
        CK(hipStreamSynchronize(s));
        CK(sync_alloc ? hipFree(raw) : hipFreeAsync(raw, s));
        CK(hipMemsetAsync(bad, 0, 32, s));
        hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, seed, bad);
        unsigned long long hbv[4] = {0, 0, 0, 0};
        CK(hipGetLastError());
        CK(hipMemcpyAsync(hbv, bad, 32, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        const unsigned long long hb = hbv[0];
        if (hb)
            printf("   first bad entry %llu: got x %08x y %08x z %08x w %08x (w ^ 0x55 = seed %u), expected seed %u\n", hbv[1],
                   (unsigned)hbv[2], (unsigned)hbv[3], (unsigned)(hbv[3] >> 32), (unsigned)(hbv[2] >> 32),
                   (unsigned)(hbv[2] >> 32) ^ 0x55u, seed);
        printf("mode %d round %d  %zu MiB  %s  bad entries %llu  (raw %p out %p aux %p)\n", mode, round, mib, sync_alloc ? "hipMalloc" : "hipMallocAsync", hb, (void *)raw, (void *)out, (void *)aux);
    }
    return 0;
}
