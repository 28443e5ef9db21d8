// What the shader clock is while a kernel of N waves runs: clock64() (shader-clock counter) against wall_clock64() (constant
// 100 MHz) around a dependent chain of packed FMAs.  A frame that ends on a few hundred long waves (C2's never-escaping pixels,
// the last waves of a rank) runs at whatever the power management gives an almost idle chip.
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/sclk_probe tools/microbench/sclk_probe.hip ; run: /tmp/sclk_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void k(unsigned long long *out, int iters)
{
    f2 a = {1.0f + threadIdx.x * 1e-7f, 0.5f}, b = {0.999999f, 1.000001f}, c = {1e-9f, -1e-9f};
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++)
            a = __builtin_elementwise_fma(a, b, c);
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = w1 - w0;
        out[3 * blockIdx.x + 1] = c1 - c0;
        out[3 * blockIdx.x + 2] = (unsigned long long)(a.x + a.y);
    }
}

int main()
{
    unsigned long long *d, h[3 * 4096];
    hipMalloc(&d, sizeof(h));
    const int grids[] = {1, 64, 256, 1024, 4096};
    for (int g : grids) {
        const int iters = 400000;
        hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d, 1000); // warm
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 3 * g * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const double wall_s = h[0] / 100e6, mhz = h[1] / wall_s / 1e6;
        printf("{\"workgroups_of_4_waves\": %d, \"kernel_ms\": %.2f, \"counter_ticks_per_us\": %.1f, \"ns_per_dependent_pk_fma\": %.3f}\n", g,
               wall_s * 1e3, mhz, wall_s * 1e9 / ((double)iters * 16));
    }
    return 0;
}
