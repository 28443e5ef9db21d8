// glds_probe.hip -- does one global_load_lds_dwordx4 per wave land 64 x 16 B contiguously at the wave-uniform LDS
// address in M0, and is it readable after s_waitcnt vmcnt(0)?  (the staging primitive of the kLds variant of
// k_lav2_hdr32_fast).  hipcc --offload-arch=gfx950 -O3 glds_probe.hip -o glds_probe && ./glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k(const float4 *src, float4 *out, int use_builtin)
{
    __shared__ float4 buf[4 * 2 * 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float4 *wbuf = buf + wave * 128u;
    const uint32_t lds_base =
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4 *)wbuf);
    for (uint32_t ch = 0; ch < 2; ch++) {
        const float4 *s = src + (blockIdx.x * 4 + wave) * 128u + ch * 64u + lane;
        const uint32_t dst = lds_base + (ch << 10);
        if (use_builtin) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)s,
                                             (__attribute__((address_space(3))) void *)(wbuf + ch * 64u), 16, 0, 0);
        } else {
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(s), "s"(dst)
                         : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // broadcast reads: every lane reads entry (lane % 7) * 9 % 128 ... and its own
    for (uint32_t i = 0; i < 2; i++)
        out[((blockIdx.x * 4 + wave) * 2 + i) * 64u + lane] = wbuf[i * 64u + lane];
}

int main()
{
    const int blocks = 8, n = blocks * 4 * 128;
    std::vector<float4> h(n);
    for (int i = 0; i < n; i++)
        h[i] = float4{(float)i, (float)(i * 2), (float)(i * 3), (float)(i * 5)};
    float4 *d, *o;
    hipMalloc(&d, n * sizeof(float4));
    hipMalloc(&o, n * sizeof(float4));
    hipMemcpy(d, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
    for (int b = 0; b < 2; b++) {
        hipMemset(o, 0xff, n * sizeof(float4));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, o, b);
        std::vector<float4> r(n);
        hipMemcpy(r.data(), o, n * sizeof(float4), hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int i = 0; i < n; i++)
            if (r[i].x != h[i].x || r[i].y != h[i].y || r[i].z != h[i].z || r[i].w != h[i].w) {
                if (first < 0)
                    first = i;
                bad++;
            }
        printf("%s: %d of %d entries wrong (first %d: got %g %g %g %g)\n", b ? "builtin" : "asm", bad, n, first,
               first >= 0 ? r[first].x : 0.f, first >= 0 ? r[first].y : 0.f, first >= 0 ? r[first].z : 0.f,
               first >= 0 ? r[first].w : 0.f);
    }
    return 0;
}
