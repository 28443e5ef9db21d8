// VALU issue-rate probe for gfx950: cycles per wave64 instruction per SIMD at full occupancy for the instruction kinds the
// perturbation kernels are made of.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run on the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
#define REP256(X) REP16(REP16(X))

template <int KIND> __global__ void __launch_bounds__(256) k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { // v_add_f32, 8 independent chains
            REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1\n"
                               "v_add_f32 %5, %5, %1\n v_add_f32 %6, %6, %1\n v_add_f32 %7, %7, %1\n v_add_f32 %8, %8, %1"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0.x));)
        } else if (KIND == 1) { // v_pk_add_f32, 4 chains x2
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
        } else if (KIND == 2) { // v_add_u32
            REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                               "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0));)
        } else if (KIND == 3) { // v_lshl_add_u32 / v_max3_i32 / v_med3_i32 / v_add3_u32 (VOP3 integer)
            REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_max3_i32 %1, %1, %4, %0\n v_med3_i32 %2, %2, %4, %1\n v_add3_u32 %3, %3, %4, %2\n"
                               "v_lshl_add_u32 %0, %0, 1, %4\n v_max3_i32 %1, %1, %4, %0\n v_med3_i32 %2, %2, %4, %1\n v_add3_u32 %3, %3, %4, %2"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0));)
        } else if (KIND == 4) { // v_pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
        } else if (KIND == 5) { // v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %1, %1\n v_fma_f32 %3, %3, %1, %1\n v_fma_f32 %4, %4, %1, %1\n"
                               "v_fma_f32 %5, %5, %1, %1\n v_fma_f32 %6, %6, %1, %1\n v_fma_f32 %7, %7, %1, %1\n v_fma_f32 %8, %8, %1, %1"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0.x));)
        } else if (KIND == 6) { // v_cmp_lt_i32 to SGPR pair + s_or (the wave-vote pattern)
            REP16(asm volatile("v_cmp_lt_i32 s[20:21], %0, %1\n v_cmp_lt_i32 s[22:23], %1, %2\n s_or_b64 s[20:21], s[20:21], s[22:23]\n"
                               "v_cmp_lt_i32 s[22:23], %2, %3\n v_cmp_lt_i32 s[24:25], %3, %0\n s_or_b64 s[22:23], s[22:23], s[24:25]\n"
                               "v_cmp_lt_i32 s[20:21], %0, %1\n v_cmp_lt_i32 s[22:23], %1, %2\n s_or_b64 s[20:21], s[20:21], s[22:23]\n"
                               "v_cmp_lt_i32 s[22:23], %2, %3\n v_cmp_lt_i32 s[24:25], %3, %0\n s_or_b64 s[22:23], s[22:23], s[24:25]"
                               : : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "s20", "s21", "s22", "s23", "s24", "s25");)
        } else if (KIND == 7) { // 8 x (v_add_f32 + an independent s_add_u32): is scalar work issued alongside vector work?
            REP16(asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %2, %2, %1\n s_add_u32 s21, s21, 1\n"
                               "v_add_f32 %3, %3, %1\n s_add_u32 s22, s22, 1\n v_add_f32 %4, %4, %1\n s_add_u32 s23, s23, 1\n"
                               "v_add_f32 %5, %5, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %6, %6, %1\n s_add_u32 s21, s21, 1\n"
                               "v_add_f32 %7, %7, %1\n s_add_u32 s22, s22, 1\n v_add_f32 %8, %8, %1\n s_add_u32 s23, s23, 1"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0.x)
                               :
                               : "s20", "s21", "s22", "s23", "scc");)
        } else if (KIND == 8) { // 8 x (v_add_f32 + s_nop 0)
            REP16(asm volatile("v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %2, %2, %1\n s_nop 0\n"
                               "v_add_f32 %3, %3, %1\n s_nop 0\n v_add_f32 %4, %4, %1\n s_nop 0\n"
                               "v_add_f32 %5, %5, %1\n s_nop 0\n v_add_f32 %6, %6, %1\n s_nop 0\n"
                               "v_add_f32 %7, %7, %1\n s_nop 0\n v_add_f32 %8, %8, %1\n s_nop 0"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0.x));)
        } else if (KIND == 9) { // 8 x (v_add_f32 + 2 independent s_add_u32)
            REP16(asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n s_add_u32 s24, s24, 1\n v_add_f32 %2, %2, %1\n s_add_u32 s21, s21, 1\n s_add_u32 s25, s25, 1\n"
                               "v_add_f32 %3, %3, %1\n s_add_u32 s22, s22, 1\n s_add_u32 s26, s26, 1\n v_add_f32 %4, %4, %1\n s_add_u32 s23, s23, 1\n s_add_u32 s27, s27, 1\n"
                               "v_add_f32 %5, %5, %1\n s_add_u32 s20, s20, 1\n s_add_u32 s24, s24, 1\n v_add_f32 %6, %6, %1\n s_add_u32 s21, s21, 1\n s_add_u32 s25, s25, 1\n"
                               "v_add_f32 %7, %7, %1\n s_add_u32 s22, s22, 1\n s_add_u32 s26, s26, 1\n v_add_f32 %8, %8, %1\n s_add_u32 s23, s23, 1\n s_add_u32 s27, s27, 1"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0.x)
                               :
                               : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");)
        } else if (KIND == 10) { // dependent chain: 8 v_pk_fma_f32 on ONE register (latency, hidden by the other 7 waves?)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n"
                               "v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1"
                               : "+v"(p0) : "v"(p1));)
        } else if (KIND == 11) { // v_pk_fma_f32, 4 independent chains
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
        } else if (KIND == 12) { // v_pk_mul_f32 with the op_sel swizzles of the complex product
            REP16(asm volatile("v_pk_mul_f32 %0, %4, %0 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %4, %1 op_sel:[1,1] op_sel_hi:[1,0]\n"
                               "v_pk_mul_f32 %2, %4, %2 op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %4, %3 op_sel:[1,1] op_sel_hi:[1,0]\n"
                               "v_pk_mul_f32 %0, %4, %0 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %4, %1 op_sel:[1,1] op_sel_hi:[1,0]\n"
                               "v_pk_mul_f32 %2, %4, %2 op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %4, %3 op_sel:[1,1] op_sel_hi:[1,0]"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
        } else if (KIND == 13) { // v_cmp_nle_f32_e64 into SGPR pairs (the wave-vote compares)
            REP16(asm volatile("v_cmp_nle_f32_e64 s[20:21], %0, %1\n v_cmp_nle_f32_e64 s[22:23], %1, %2\n v_cmp_nle_f32_e64 s[24:25], %2, %3\n v_cmp_nle_f32_e64 s[26:27], %3, %0\n"
                               "v_cmp_nle_f32_e64 s[20:21], %0, %1\n v_cmp_nle_f32_e64 s[22:23], %1, %2\n v_cmp_nle_f32_e64 s[24:25], %2, %3\n v_cmp_nle_f32_e64 s[26:27], %3, %0"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                               :
                               : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (KIND == 14) { // v_max_f32_e64 with |abs| modifiers
            REP16(asm volatile("v_max_f32_e64 %0, |%0|, |%4|\n v_max_f32_e64 %1, |%1|, |%4|\n v_max_f32_e64 %2, |%2|, |%4|\n v_max_f32_e64 %3, |%3|, |%4|\n"
                               "v_max_f32_e64 %0, |%0|, |%4|\n v_max_f32_e64 %1, |%1|, |%4|\n v_max_f32_e64 %2, |%2|, |%4|\n v_max_f32_e64 %3, |%3|, |%4|"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (KIND == 15) { // v_cmp_nle_f32 e32 (VOPC, writes vcc)
            REP16(asm volatile("v_cmp_nle_f32 vcc, %0, %1\n v_cmp_nle_f32 vcc, %1, %2\n v_cmp_nle_f32 vcc, %2, %3\n v_cmp_nle_f32 vcc, %3, %0\n"
                               "v_cmp_nle_f32 vcc, %0, %1\n v_cmp_nle_f32 vcc, %1, %2\n v_cmp_nle_f32 vcc, %2, %3\n v_cmp_nle_f32 vcc, %3, %0"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                               :
                               : "vcc");)
        } else if (KIND == 16) { // the scaled step's vector instructions, as a unit (13 per rep, 2 independent copies)
            REP16(asm volatile("v_pk_fma_f32 %0, %1, %4, %0\n v_pk_mul_f32 %2, %1, %0 op_sel_hi:[0,1]\n v_pk_mul_f32 %0, %1, %0 op_sel:[1,1] op_sel_hi:[1,0]\n"
                               "v_pk_add_f32 %0, %2, %0 neg_lo:[0,1] neg_hi:[0,0]\n v_pk_add_f32 %1, %4, %0\n"
                               "v_max_f32_e64 %5, |%6|, |%7|\n v_min_f32_e64 %7, |%6|, |%5|\n v_mul_f32 %8, %5, %6\n v_cmp_nle_f32_e64 s[20:21], %8, %7\n"
                               "v_mul_f32 %8, 0x2b800000, %5\n v_cmp_nge_f32_e64 s[22:23], %7, %8\n v_add_u32 %8, 0xac800000, %5\n v_cmp_gt_u32_e64 s[24:25], %9, %8"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0)
                               :
                               : "s20", "s21", "s22", "s23", "s24", "s25");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x +
                                                 p2.y + p3.x + p3.y + i0 + i1 + i2 + i3;
}

template <int KIND> void run(const char *name, int valu_per_rep)
{
    float *out;
    const int blocks = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipMalloc(&out, blocks * 256 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 16 * valu_per_rep * 8; // 8 waves per SIMD
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("{\"kind\": \"%s\", \"ms\": %.3f, \"cycles_per_wave64_valu_at_2.4GHz\": %.3f}\n", name, ms, cyc / insts_per_simd);
    hipFree(out);
}

int main()
{
    run<0>("v_add_f32", 8);
    run<5>("v_fma_f32", 8);
    run<1>("v_pk_add_f32", 8);
    run<4>("v_pk_mul_f32", 8);
    run<2>("v_add_u32", 8);
    run<3>("vop3_int(lshl_add,max3,med3,add3)", 8);
    run<6>("v_cmp->sgpr (8 per rep) + s_or", 8);
    run<7>("v_add_f32 + 1 s_add_u32 each", 8);
    run<9>("v_add_f32 + 2 s_add_u32 each", 8);
    run<8>("v_add_f32 + s_nop 0 each", 8);
    run<10>("v_pk_fma_f32 dependent chain", 8);
    run<11>("v_pk_fma_f32", 8);
    run<12>("v_pk_mul_f32 op_sel swizzles", 8);
    run<13>("v_cmp_nle_f32_e64 -> sgpr", 8);
    run<15>("v_cmp_nle_f32 e32 -> vcc", 8);
    run<14>("v_max_f32_e64 |a|,|b|", 8);
    run<16>("scaled step, 13 vector instructions (per instruction)", 13);
    return 0;
}
