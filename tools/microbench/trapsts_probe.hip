// Do the sticky IEEE exception bits of TRAPSTS (EXCP field, bits 8:0) accumulate on gfx950 without enabling traps?
// If so, "did anything underflow / overflow / go denormal during this run of steps" is one s_getreg per run instead of
// per-step range tests.  Build: hipcc --offload-arch=gfx950 -O1 -o trapsts_probe trapsts_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(float a, float b, float c, unsigned *out)
{
    unsigned t0, t1, t2, t3, t4, mode;
    float r1, r2, r3, r4;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t0));
    asm volatile("v_mul_f32 %0, %1, %2\n s_nop 4" : "=v"(r1) : "v"(a), "v"(a)); // 1.5 * 1.5: inexact? (exact) -> nothing
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t1));
    asm volatile("v_mul_f32 %0, %1, %2\n s_nop 4" : "=v"(r2) : "v"(b), "v"(b)); // 2^-100 * 2^-100: underflow
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t2));
    asm volatile("v_mul_f32 %0, %1, %2\n s_nop 4" : "=v"(r3) : "v"(c), "v"(c)); // 2^100 * 2^100: overflow
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t3));
    asm volatile("v_pk_mul_f32 %0, %1, %1\n s_nop 4" : "=v"(*(double *)&r4) : "v"((double)b)); // packed op on garbage
    asm volatile("v_add_f32 %0, %1, %2\n s_nop 4" : "=v"(r4) : "v"(1.0f), "v"(0x1p-30f)); // inexact
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t4));
    if (threadIdx.x == 0) {
        out[0] = mode, out[1] = t0, out[2] = t1, out[3] = t2, out[4] = t3, out[5] = t4;
        out[6] = __float_as_uint(r1), out[7] = __float_as_uint(r2), out[8] = __float_as_uint(r3), out[9] = __float_as_uint(r4);
    }
}

int main()
{
    unsigned *out;
    hipHostMalloc((void **)&out, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 1.5f, 0x1p-100f, 0x1p100f, out);
    hipDeviceSynchronize();
    printf("{\"mode\": \"0x%08x\", \"trapsts_cleared\": \"0x%08x\", \"after_exact_mul\": \"0x%08x\", \"after_underflow\": \"0x%08x\", "
           "\"after_overflow\": \"0x%08x\", \"after_inexact_add\": \"0x%08x\", \"r\": [\"0x%08x\", \"0x%08x\", \"0x%08x\", \"0x%08x\"]}\n",
           out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], out[8], out[9]);
    return 0;
}
