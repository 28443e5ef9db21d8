// Issue cost of the instruction kinds the hand-written BLA loop (kernels_bla_fast.hip) is made of: cycles per wave64
// instruction per SIMD with 8 waves per SIMD, eight independent instructions per repetition.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates2 valu_rates2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X

#define KERNEL(NAME, TEXT)                                                                                              \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters)                                                  \
    {                                                                                                                   \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                                                  \
        int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;                                                    \
        for (int it = 0; it < iters; it++) {                                                                            \
            REP16(asm volatile(TEXT : "+{v[10:11]}"(*(double *)&a0), "+{v[12:13]}"(*(double *)&a2), "+{v20}"(i0),       \
                               "+{v21}"(i1), "+{v22}"(i2), "+{v23}"(i3)                                                 \
                               :                                                                                        \
                               : "v14", "v15", "v16", "v17", "v18", "v19", "v24", "v25", "v26", "v27", "s20", "s21", "s22", \
                                 "s23", "s24", "s25", "s26", "s27", "vcc");)                                            \
        }                                                                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3;                             \
    }

// registers: v10,v11,v12,v13 floats (pairs v[10:11], v[12:13]); v20..v23 ints; v14..v19, v24..v27 scratch destinations
KERNEL(k_add, "v_add_f32 v14, v10, v11\n v_add_f32 v15, v10, v11\n v_add_f32 v16, v10, v11\n v_add_f32 v17, v10, v11\n"
              "v_add_f32 v18, v10, v11\n v_add_f32 v19, v10, v11\n v_add_f32 v24, v10, v11\n v_add_f32 v25, v10, v11")
KERNEL(k_ldexp, "v_ldexp_f32 v14, v10, v20\n v_ldexp_f32 v15, v11, v21\n v_ldexp_f32 v16, v12, v22\n v_ldexp_f32 v17, v13, v23\n"
                "v_ldexp_f32 v18, v10, v21\n v_ldexp_f32 v19, v11, v22\n v_ldexp_f32 v24, v12, v23\n v_ldexp_f32 v25, v13, v20")
KERNEL(k_cmp64, "v_cmp_lt_i64 s[20:21], v[10:11], v[12:13]\n v_cmp_lt_i64 s[22:23], v[12:13], v[10:11]\n"
                "v_cmp_lt_i64 s[24:25], v[10:11], v[20:21]\n v_cmp_lt_i64 s[26:27], v[12:13], v[22:23]\n"
                "v_cmp_lt_i64 s[20:21], v[20:21], v[12:13]\n v_cmp_lt_i64 s[22:23], v[22:23], v[10:11]\n"
                "v_cmp_lt_i64 s[24:25], v[10:11], v[22:23]\n v_cmp_lt_i64 s[26:27], v[12:13], v[20:21]")
KERNEL(k_cmp32, "v_cmp_lt_i32 s[20:21], v20, v21\n v_cmp_lt_i32 s[22:23], v21, v22\n v_cmp_lt_i32 s[24:25], v22, v23\n"
                "v_cmp_lt_i32 s[26:27], v23, v20\n v_cmp_lt_i32 s[20:21], v20, v22\n v_cmp_lt_i32 s[22:23], v21, v23\n"
                "v_cmp_lt_i32 s[24:25], v22, v20\n v_cmp_lt_i32 s[26:27], v23, v21")
KERNEL(k_pkmov, "v_pk_mov_b32 v[14:15], v[10:11], v[10:11] op_sel:[0,1]\n v_pk_mov_b32 v[16:17], v[12:13], v[12:13] op_sel:[0,1]\n"
                "v_pk_mov_b32 v[18:19], v[10:11], v[12:13] op_sel:[0,1]\n v_pk_mov_b32 v[24:25], v[12:13], v[10:11] op_sel:[0,1]\n"
                "v_pk_mov_b32 v[14:15], v[20:21], v[20:21] op_sel:[0,1]\n v_pk_mov_b32 v[16:17], v[22:23], v[22:23] op_sel:[0,1]\n"
                "v_pk_mov_b32 v[18:19], v[20:21], v[22:23] op_sel:[0,1]\n v_pk_mov_b32 v[24:25], v[22:23], v[20:21] op_sel:[0,1]")
KERNEL(k_mov, "v_mov_b32 v14, v10\n v_mov_b32 v15, v11\n v_mov_b32 v16, v12\n v_mov_b32 v17, v13\n v_mov_b32 v18, v20\n"
              "v_mov_b32 v19, v21\n v_mov_b32 v24, v22\n v_mov_b32 v25, v23")
KERNEL(k_max3f, "v_max3_f32 v14, |v10|, |v11|, |v12|\n v_min3_f32 v15, |v10|, |v11|, |v12|\n v_max3_f32 v16, |v13|, |v11|, v14\n"
                "v_min3_f32 v17, |v13|, |v10|, v15\n v_max3_f32 v18, |v10|, |v11|, |v12|\n v_min3_f32 v19, |v10|, |v11|, |v12|\n"
                "v_max3_f32 v24, |v13|, |v11|, v18\n v_min3_f32 v25, |v13|, |v10|, v19")
KERNEL(k_int3, "v_max3_i32 v14, v20, v21, v22\n v_min3_i32 v15, v20, v21, v22\n v_add3_u32 v16, v20, v21, v22\n"
               "v_lshl_add_u32 v17, v20, 1, v21\n v_and_or_b32 v18, v20, v21, 1.0\n v_bfe_u32 v19, v20, 23, 8\n"
               "v_mad_u32_u24 v24, v20, 12, -12\n v_mul_u32_u24 v25, 48, v21")
KERNEL(k_cnd, "v_cndmask_b32_e64 v14, v20, 3, s[20:21]\n v_cndmask_b32_e64 v15, v21, 2, s[22:23]\n v_cndmask_b32_e64 v16, v22, 1, s[24:25]\n"
              "v_cndmask_b32_e64 v17, v23, 0, s[26:27]\n v_cndmask_b32_e64 v18, v20, 3, s[20:21]\n v_cndmask_b32_e64 v19, v21, 2, s[22:23]\n"
              "v_cndmask_b32_e64 v24, v22, 1, s[24:25]\n v_cndmask_b32_e64 v25, v23, 0, s[26:27]")
KERNEL(k_pkmul, "v_pk_mul_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 v[16:17], v[10:11], v[12:13] op_sel:[1,1] op_sel_hi:[1,0]\n"
                "v_pk_add_f32 v[18:19], v[10:11], v[12:13] neg_lo:[0,1] neg_hi:[0,0]\n v_pk_add_f32 v[24:25], v[10:11], v[12:13]\n"
                "v_pk_mul_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 v[16:17], v[10:11], v[12:13] op_sel:[1,1] op_sel_hi:[1,0]\n"
                "v_pk_add_f32 v[18:19], v[10:11], v[12:13] neg_lo:[0,1] neg_hi:[0,0]\n v_pk_add_f32 v[24:25], v[10:11], v[12:13]")
KERNEL(k_sub, "v_sub_u32 v14, v20, v21\n v_sub_u32 v15, v21, v22\n v_sub_u32 v16, v22, v23\n v_sub_u32 v17, v23, v20\n"
              "v_max_i32 v18, v20, v21\n v_max_i32 v19, v21, v22\n v_lshlrev_b32 v24, 1, v22\n v_add_u32_e64 v25, v23, v20 clamp")

template <class K> void run(const char *name, K kern)
{
    float *out;
    const int blocks = 256 * 8;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 16 * 8 * 8;
    printf("{\"kind\": \"%s\", \"ms\": %.3f, \"cycles_per_wave64_instruction_at_2.4GHz\": %.3f}\n", name, ms,
           ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(out);
}

int main()
{
    run("v_add_f32", k_add);
    run("v_ldexp_f32", k_ldexp);
    run("v_cmp_lt_i64 -> sgpr", k_cmp64);
    run("v_cmp_lt_i32 -> sgpr", k_cmp32);
    run("v_pk_mov_b32", k_pkmov);
    run("v_mov_b32", k_mov);
    run("v_max3_f32 / v_min3_f32 |abs|", k_max3f);
    run("max3_i32, min3_i32, add3, lshl_add, and_or, bfe, mad_u24, mul_u24", k_int3);
    run("v_cndmask_b32_e64 sgpr mask", k_cnd);
    run("v_pk_mul_f32 / v_pk_add_f32 with swizzles", k_pkmul);
    run("sub, max_i32, lshlrev, add clamp", k_sub);
    return 0;
}
