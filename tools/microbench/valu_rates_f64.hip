// Issue cost of the binary64 instruction kinds the HDRFloat<double> LAv2 kernel (kernels_hdr64.hip) is made of: cycles per wave64
// instruction per SIMD with 8 waves per SIMD, eight independent instructions per repetition (same harness as valu_rates2.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates_f64 valu_rates_f64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X

#define KERNEL(NAME, TEXT)                                                                                              \
    __global__ void __launch_bounds__(256) NAME(double *out, int iters)                                                 \
    {                                                                                                                   \
        double a0 = 1.0 + threadIdx.x * 1e-3, a1 = a0 + 0.25, a2 = a0 + 0.5, a3 = a0 + 0.75;                           \
        int i0 = threadIdx.x & 7, i1 = i0 + 1, i2 = i0 - 2, i3 = i0 - 3;                                                \
        for (int it = 0; it < iters; it++) {                                                                            \
            REP16(asm volatile(TEXT : "+{v[10:11]}"(a0), "+{v[12:13]}"(a1), "+{v[14:15]}"(a2), "+{v[16:17]}"(a3),      \
                               "+{v20}"(i0), "+{v21}"(i1), "+{v22}"(i2), "+{v23}"(i3)                                   \
                               :                                                                                        \
                               : "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",  \
                                 "v44", "v45", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc");)        \
        }                                                                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3;                             \
    }

#define D8(OP, A, B) OP " v[30:31], " A ", " B "\n" OP " v[32:33], " B ", " A "\n" OP " v[34:35], " A ", v[14:15]\n" OP " v[36:37], " B ", v[16:17]\n" \
                     OP " v[38:39], v[14:15], " A "\n" OP " v[40:41], v[16:17], " B "\n" OP " v[42:43], v[14:15], v[16:17]\n" OP " v[44:45], v[16:17], v[14:15]"
KERNEL(k_mul, D8("v_mul_f64", "v[10:11]", "v[12:13]"))
KERNEL(k_add, D8("v_add_f64", "v[10:11]", "v[12:13]"))
KERNEL(k_fma, "v_fma_f64 v[30:31], v[10:11], v[12:13], v[14:15]\n v_fma_f64 v[32:33], v[12:13], v[10:11], v[16:17]\n"
              "v_fma_f64 v[34:35], v[10:11], v[14:15], v[12:13]\n v_fma_f64 v[36:37], v[12:13], v[16:17], v[10:11]\n"
              "v_fma_f64 v[38:39], v[14:15], v[10:11], v[16:17]\n v_fma_f64 v[40:41], v[16:17], v[12:13], v[14:15]\n"
              "v_fma_f64 v[42:43], v[14:15], v[16:17], v[10:11]\n v_fma_f64 v[44:45], v[16:17], v[14:15], v[12:13]")
KERNEL(k_ldexp, "v_ldexp_f64 v[30:31], v[10:11], v20\n v_ldexp_f64 v[32:33], v[12:13], v21\n v_ldexp_f64 v[34:35], v[14:15], v22\n"
                "v_ldexp_f64 v[36:37], v[16:17], v23\n v_ldexp_f64 v[38:39], v[10:11], v21\n v_ldexp_f64 v[40:41], v[12:13], v22\n"
                "v_ldexp_f64 v[42:43], v[14:15], v23\n v_ldexp_f64 v[44:45], v[16:17], v20")
KERNEL(k_cmp, "v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n v_cmp_lt_f64 vcc, v[12:13], v[14:15]\n v_cmp_lt_f64 vcc, v[14:15], v[16:17]\n"
              "v_cmp_lt_f64 vcc, v[16:17], v[10:11]\n v_cmp_ge_f64 vcc, v[10:11], v[14:15]\n v_cmp_ge_f64 vcc, v[12:13], v[16:17]\n"
              "v_cmp_ge_f64 vcc, v[14:15], v[10:11]\n v_cmp_ge_f64 vcc, v[16:17], v[12:13]")
KERNEL(k_cmps, "v_cmp_lt_f64 s[20:21], v[10:11], v[12:13]\n v_cmp_lt_f64 s[22:23], v[12:13], v[14:15]\n v_cmp_lt_f64 s[24:25], v[14:15], v[16:17]\n"
               "v_cmp_lt_f64 s[26:27], v[16:17], v[10:11]\n v_cmp_ge_f64 s[20:21], v[10:11], v[14:15]\n v_cmp_ge_f64 s[22:23], v[12:13], v[16:17]\n"
               "v_cmp_ge_f64 s[24:25], v[14:15], v[10:11]\n v_cmp_ge_f64 s[26:27], v[16:17], v[12:13]")
KERNEL(k_mov64, "v_mov_b64 v[30:31], v[10:11]\n v_mov_b64 v[32:33], v[12:13]\n v_mov_b64 v[34:35], v[14:15]\n v_mov_b64 v[36:37], v[16:17]\n"
                "v_mov_b64 v[38:39], v[10:11]\n v_mov_b64 v[40:41], v[12:13]\n v_mov_b64 v[42:43], v[14:15]\n v_mov_b64 v[44:45], v[16:17]")
KERNEL(k_cnd, "v_cndmask_b32_e32 v30, v10, v12, vcc\n v_cndmask_b32_e32 v31, v11, v13, vcc\n v_cndmask_b32_e32 v32, v14, v16, vcc\n"
              "v_cndmask_b32_e32 v33, v15, v17, vcc\n v_cndmask_b32_e32 v34, v12, v10, vcc\n v_cndmask_b32_e32 v35, v13, v11, vcc\n"
              "v_cndmask_b32_e32 v36, v16, v14, vcc\n v_cndmask_b32_e32 v37, v17, v15, vcc")
KERNEL(k_cnd_cmp, "v_cmp_lt_i32_e32 vcc, v20, v21\n v_cndmask_b32_e32 v30, v10, v12, vcc\n v_cndmask_b32_e32 v31, v11, v13, vcc\n v_cndmask_b32_e32 v32, v14, v16, vcc\n"
                  "v_cndmask_b32_e32 v33, v15, v17, vcc\n v_cndmask_b32_e32 v34, v12, v10, vcc\n v_cndmask_b32_e32 v35, v13, v11, vcc\n"
                  "v_cndmask_b32_e32 v36, v16, v14, vcc")
KERNEL(k_cnd64, "v_cndmask_b32_e64 v30, v10, v12, s[20:21]\n v_cndmask_b32_e64 v31, v11, v13, s[20:21]\n v_cndmask_b32_e64 v32, v14, v16, s[22:23]\n"
                "v_cndmask_b32_e64 v33, v15, v17, s[22:23]\n v_cndmask_b32_e64 v34, v12, v10, s[24:25]\n v_cndmask_b32_e64 v35, v13, v11, s[24:25]\n"
                "v_cndmask_b32_e64 v36, v16, v14, s[26:27]\n v_cndmask_b32_e64 v37, v17, v15, s[26:27]")
KERNEL(k_cnd64vcc, "v_cndmask_b32_e64 v30, v10, v12, vcc\n v_cndmask_b32_e64 v31, v11, v13, vcc\n v_cndmask_b32_e64 v32, v14, v16, vcc\n"
                   "v_cndmask_b32_e64 v33, v15, v17, vcc\n v_cndmask_b32_e64 v34, v12, v10, vcc\n v_cndmask_b32_e64 v35, v13, v11, vcc\n"
                   "v_cndmask_b32_e64 v36, v16, v14, vcc\n v_cndmask_b32_e64 v37, v17, v15, vcc")
KERNEL(k_cnd32i, "v_cndmask_b32_e32 v30, v20, v21, vcc\n v_cndmask_b32_e32 v31, v21, v22, vcc\n v_cndmask_b32_e32 v32, v22, v23, vcc\n"
                 "v_cndmask_b32_e32 v33, v23, v20, vcc\n v_cndmask_b32_e32 v34, v20, v22, vcc\n v_cndmask_b32_e32 v35, v21, v23, vcc\n"
                 "v_cndmask_b32_e32 v36, v22, v20, vcc\n v_cndmask_b32_e32 v37, v23, v21, vcc")
KERNEL(k_addc, "v_add_co_u32_e32 v30, vcc, v20, v21\n v_addc_co_u32_e32 v31, vcc, v21, v22, vcc\n v_add_co_u32_e32 v32, vcc, v22, v23\n"
               "v_addc_co_u32_e32 v33, vcc, v23, v20, vcc\n v_add_co_u32_e32 v34, vcc, v20, v22\n v_addc_co_u32_e32 v35, vcc, v21, v23, vcc\n"
               "v_add_co_u32_e32 v36, vcc, v22, v20\n v_addc_co_u32_e32 v37, vcc, v23, v21, vcc")
KERNEL(k_clamp3, "v_add_u32_e32 v30, 0x77, v20\n v_ashrrev_i32_e32 v30, 31, v30\n v_lshl_add_u32 v31, v30, 12, v20\n v_add_u32_e32 v32, 0x77, v21\n"
                 "v_ashrrev_i32_e32 v32, 31, v32\n v_lshl_add_u32 v33, v32, 12, v21\n v_med3_i32 v34, v20, v21, v22\n v_min_i32_e32 v35, v22, v23")
KERNEL(k_int, "v_sub_u32_e32 v30, v20, v21\n v_max_i32_e32 v31, v21, v22\n v_bfe_u32 v32, v11, 20, 11\n v_bfe_u32 v33, v13, 20, 11\n"
              "v_cmp_lt_i32_e32 vcc, v22, v23\n v_add3_u32 v34, v20, v21, v22\n v_max_u32_e32 v35, v22, v23\n v_lshlrev_b32_e32 v36, 1, v20")
KERNEL(k_frexp, "v_frexp_exp_i32_f64 v30, v[10:11]\n v_frexp_exp_i32_f64 v31, v[12:13]\n v_frexp_mant_f64 v[32:33], v[14:15]\n"
                "v_frexp_mant_f64 v[34:35], v[16:17]\n v_frexp_exp_i32_f64 v36, v[14:15]\n v_frexp_exp_i32_f64 v37, v[16:17]\n"
                "v_frexp_mant_f64 v[38:39], v[10:11]\n v_frexp_mant_f64 v[40:41], v[12:13]")

template <class K> void run(const char *name, K kern)
{
    double *out;
    const int blocks = 256 * 8;
    hipMalloc(&out, blocks * 256 * sizeof(double));
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
    run("v_mul_f64", k_mul);
    run("v_add_f64", k_add);
    run("v_fma_f64", k_fma);
    run("v_ldexp_f64", k_ldexp);
    run("v_cmp_*_f64 -> vcc", k_cmp);
    run("v_cmp_*_f64 -> sgpr pair", k_cmps);
    run("v_mov_b64", k_mov64);
    run("v_cndmask_b32_e32 (vcc)", k_cnd);
    run("v_cmp_lt_i32 vcc + 7 v_cndmask_b32_e32 (vcc)", k_cnd_cmp);
    run("v_cndmask_b32_e64 (sgpr pair)", k_cnd64);
    run("v_cndmask_b32_e64 (vcc as the mask)", k_cnd64vcc);
    run("v_cndmask_b32_e32 (vcc), integer-register sources", k_cnd32i);
    run("v_add_co_u32 / v_addc_co_u32 (vcc)", k_addc);
    run("add, ashrrev, lshl_add x2, med3_i32, min_i32", k_clamp3);
    run("sub, max_i32, bfe x2, cmp_i32, add3, max_u32, lshlrev", k_int);
    run("v_frexp_exp_i32_f64 / v_frexp_mant_f64", k_frexp);
    return 0;
}
