// Does a wave64 vector instruction cost less when part of EXEC is zero?  Same instruction stream (8 independent VOP3P / VOP2
// instructions per repetition, 8 waves per SIMD) under EXEC = all 64 lanes, the lower 32, the lower 16, and 32 scattered lanes.
// Build: hipcc --offload-arch=gfx950 -O3 -o exec_mask_rate exec_mask_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
#define BODY                                                                                                            \
    "v_pk_mul_f32 v[14:15], v[10:11], v[12:13]\n v_pk_add_f32 v[16:17], v[10:11], v[12:13]\n v_ldexp_f32 v18, v10, v20\n"  \
    "v_max3_i32 v19, v20, v21, v22\n v_add_f32 v24, v10, v11\n v_sub_u32 v25, v20, v21\n v_pk_mul_f32 v[26:27], v[12:13], v[10:11]\n" \
    "v_max_i32 v18, v21, v22\n"

__global__ void __launch_bounds__(256) k(float *out, int iters, unsigned long long mask)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    int i0 = threadIdx.x & 7, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    for (int it = 0; it < iters; it++) {
        REP16(asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %6\n" BODY BODY "s_mov_b64 exec, s[20:21]"
                           : "+{v[10:11]}"(*(double *)&a0), "+{v[12:13]}"(*(double *)&a2), "+{v20}"(i0), "+{v21}"(i1),
                             "+{v22}"(i2), "+{v23}"(i3)
                           : "s"(mask)
                           : "v14", "v15", "v16", "v17", "v18", "v19", "v24", "v25", "v26", "v27", "s20", "s21");)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3;
}

int main()
{
    float *out;
    const int blocks = 256 * 8, iters = 2000;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x5555555555555555ull, 0xFFFFFFFF00000000ull, 0x1ull};
    const char *names[] = {"all 64 lanes", "lower 32", "lower 16", "32 scattered (every second lane)", "upper 32", "lane 0 only"};
    for (int m = 0; m < 6; m++) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10, masks[m]);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, masks[m]);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)iters * 16 * 16 * 8;
        printf("{\"exec\": \"%s\", \"ms\": %.3f, \"cycles_per_wave64_instruction_at_2.4GHz\": %.3f}\n", names[m], ms,
               ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
    return 0;
}
