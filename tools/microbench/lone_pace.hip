// What a wave that has its SIMD to itself pays per step of C2's untested loop, and how much that depends on WHERE it runs.
// 64 workgroups of one wave each (they land on 64 different CUs); every wave runs the same dependent chain -- per step the five
// packed operations of the scaled perturbation step, sixteen steps per body -- in three forms: 0 = arithmetic alone; 1 = plus
// the body's two 64-byte scalar loads (a 128 KB table walked in order) waited for at once; 2 = the same loads requested one body
// ahead.  Per wave: ns per step (constant 100 MHz clock), shader MHz, XCC / SE / CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o lone_pace lone_pace.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define STEP(E)                                                                                                         \
    "v_pk_fma_f32 v[56:57], v[48:49], v[40:41], " E "\n\t"                                                              \
    "v_pk_mul_f32 v[58:59], v[48:49], v[56:57] op_sel_hi:[0,1]\n\t"                                                     \
    "v_pk_mul_f32 v[56:57], v[48:49], v[56:57] op_sel:[1,1] op_sel_hi:[1,0]\n\t"                                        \
    "v_pk_add_f32 v[58:59], v[58:59], v[56:57] neg_lo:[0,1] neg_hi:[0,0]\n\t"                                           \
    "v_pk_add_f32 v[48:49], v[58:59], v[42:43]\n\t"                                                                     \
    "v_min3_f32 v61, |v48|, |v49|, v61\n\t"
#define STEPS8(A, B, C, D, E, F, G, H) STEP(A) STEP(B) STEP(C) STEP(D) STEP(E) STEP(F) STEP(G) STEP(H)
#define LOWER STEPS8("s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]")
#define UPPER STEPS8("s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]", "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]")

// the block tests of the real loop, once per four steps: as they are (compare -> scalar or -> branch), and deferred (the
// violation accumulated in a vector register, one compare and branch per body)
#define TEST_BRANCHY(B)                                                                                                 \
    "v_max_f32_e64 v60, |v48|, |v49|\n\t"                                                                               \
    "v_max_i32_e32 v62, v60, v44\n\t"                                                                                   \
    "v_add_u32_e32 v62, v62, v45\n\t"                                                                                   \
    "v_cmp_lt_i32_e64 s[76:77], " B ", v62\n\t"                                                                         \
    "v_cmp_lt_f32_e32 vcc, 0x46800000, v60\n\t"                                                                         \
    "s_or_b64 s[76:77], s[76:77], vcc\n\t"                                                                              \
    "s_cbranch_scc1 .Lout_%=\n\t"
#define TEST_DEFERRED(B)                                                                                                \
    "v_max_f32_e64 v60, |v48|, |v49|\n\t"                                                                               \
    "s_sub_u32 s78, 0, " B "\n\t"                                                                                       \
    "v_max_i32_e32 v62, v60, v44\n\t"                                                                                   \
    "v_subrev_u32_e32 v46, 0x46800000, v60\n\t"                                                                         \
    "v_add3_u32 v62, v62, v45, s78\n\t"                                                                                 \
    "v_max3_i32 v63, v63, v62, v46\n\t"
#define STEPS4(A, B, C, D) STEP(A) STEP(B) STEP(C) STEP(D)

__global__ void __launch_bounds__(64) k(const float2 *table, uint32_t table_bytes_mask, int bodies, int mode, unsigned long long *out)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w = {1e-8f * (threadIdx.x + 1), 2e-8f}, se = {1.0f, 1.0f}, dc = {1e-9f, -1e-9f};
    float mn = 1e30f;
    const unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    uint32_t off = 0;
    if (mode == 0) {
        asm volatile("s_load_dwordx16 s[36:51], %[tb], 0x0\n\ts_load_dwordx16 s[52:67], %[tb], 0x40\n\ts_waitcnt lgkmcnt(0)\n"
                     ".La_%=:\n\t" LOWER UPPER "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 .La_%=\n\t"
                     : "+{v[48:49]}"(w), "+{v61}"(mn), [n] "+s"(bodies)
                     : "{v[40:41]}"(se), "{v[42:43]}"(dc), [tb] "s"(table)
                     : "v56", "v57", "v58", "v59", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46",
                       "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61",
                       "s62", "s63", "s64", "s65", "s66", "s67", "scc", "memory");
    } else if (mode == 1) {
        asm volatile(".Lb_%=:\n\t"
                     "s_load_dwordx16 s[36:51], %[tb], %[off]\n\ts_load_dwordx16 s[52:67], %[tb], %[off] offset:0x40\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t" LOWER UPPER
                     "s_add_u32 %[off], %[off], 0x80\n\ts_and_b32 %[off], %[off], %[mask]\n\t"
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 .Lb_%=\n\t"
                     : "+{v[48:49]}"(w), "+{v61}"(mn), [n] "+s"(bodies), [off] "+s"(off)
                     : "{v[40:41]}"(se), "{v[42:43]}"(dc), [tb] "s"(table), [mask] "s"(table_bytes_mask)
                     : "v56", "v57", "v58", "v59", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46",
                       "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61",
                       "s62", "s63", "s64", "s65", "s66", "s67", "scc", "memory");
    } else if (mode == 3 || mode == 4) {
        // the pipelined loads + the block tests; the bound (s79) is never exceeded: the branches are never taken
        int imdc = 0x20000000, esh = -(24 << 23);
        float acc = 0.0f;
        if (mode == 3)
            asm volatile("s_mov_b32 s79, 0x7f000000\n\ts_load_dwordx16 s[36:51], %[tb], %[off]\n"
                         ".Le_%=:\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_load_dwordx16 s[52:67], %[tb], %[off] offset:0x40\n\t"
                         TEST_BRANCHY("s79") STEPS4("s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]")
                         TEST_BRANCHY("s79") STEPS4("s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]")
                         "s_add_u32 %[off], %[off], 0x80\n\ts_and_b32 %[off], %[off], %[mask]\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_load_dwordx16 s[36:51], %[tb], %[off]\n\t"
                         TEST_BRANCHY("s79") STEPS4("s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]")
                         TEST_BRANCHY("s79") STEPS4("s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]")
                         "v_cmp_gt_f32_e32 vcc, 0x23800000, v61\n\t"
                         "s_cbranch_vccnz .Lout_%=\n\t"
                         "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 .Le_%=\n"
                         ".Lout_%=:\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         : "+{v[48:49]}"(w), "+{v61}"(mn), [n] "+s"(bodies), [off] "+s"(off)
                         : "{v[40:41]}"(se), "{v[42:43]}"(dc), [tb] "s"(table), [mask] "s"(table_bytes_mask), "{v44}"(imdc), "{v45}"(esh)
                         : "v56", "v57", "v58", "v59", "v60", "v62", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46",
                           "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61",
                           "s62", "s63", "s64", "s65", "s66", "s67", "s76", "s77", "s79", "vcc", "scc", "memory");
        else
            asm volatile("s_mov_b32 s79, 0x7f000000\n\ts_load_dwordx16 s[36:51], %[tb], %[off]\n\tv_mov_b32_e32 v63, 0x80000000\n"
                         ".Lg_%=:\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_load_dwordx16 s[52:67], %[tb], %[off] offset:0x40\n\t"
                         TEST_DEFERRED("s79") STEPS4("s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]")
                         TEST_DEFERRED("s79") STEPS4("s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]")
                         "s_add_u32 %[off], %[off], 0x80\n\ts_and_b32 %[off], %[off], %[mask]\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_load_dwordx16 s[36:51], %[tb], %[off]\n\t"
                         TEST_DEFERRED("s79") STEPS4("s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]")
                         TEST_DEFERRED("s79") STEPS4("s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]")
                         /* one verdict per body: floor (0x23800000 = 2^-56 > the smallest part) or a block violation */
                         "v_sub_u32_e32 v46, 0x23800000, v61\n\t"
                         "v_max_i32_e32 v46, v46, v63\n\t"
                         "v_cmp_lt_i32_e32 vcc, 0, v46\n\t"
                         "s_cbranch_vccnz .Lout2_%=\n\t"
                         "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 .Lg_%=\n"
                         ".Lout2_%=:\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         : "+{v[48:49]}"(w), "+{v61}"(mn), [n] "+s"(bodies), [off] "+s"(off)
                         : "{v[40:41]}"(se), "{v[42:43]}"(dc), [tb] "s"(table), [mask] "s"(table_bytes_mask), "{v44}"(imdc), "{v45}"(esh)
                         : "v46", "v56", "v57", "v58", "v59", "v60", "v62", "v63", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46",
                           "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61",
                           "s62", "s63", "s64", "s65", "s66", "s67", "s78", "s79", "vcc", "scc", "memory");
        mn += acc;
    } else {
        asm volatile("s_load_dwordx16 s[36:51], %[tb], %[off]\n"
                     ".Lc_%=:\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_load_dwordx16 s[52:67], %[tb], %[off] offset:0x40\n\t" LOWER
                     "s_add_u32 %[off], %[off], 0x80\n\ts_and_b32 %[off], %[off], %[mask]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_load_dwordx16 s[36:51], %[tb], %[off]\n\t" UPPER
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 .Lc_%=\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     : "+{v[48:49]}"(w), "+{v61}"(mn), [n] "+s"(bodies), [off] "+s"(off)
                     : "{v[40:41]}"(se), "{v[42:43]}"(dc), [tb] "s"(table), [mask] "s"(table_bytes_mask)
                     : "v56", "v57", "v58", "v59", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46",
                       "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61",
                       "s62", "s63", "s64", "s65", "s66", "s67", "scc", "memory");
    }
    const unsigned long long t1 = wall_clock64(), c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        uint32_t hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        out[blockIdx.x * 4 + 0] = t1 - t0;
        out[blockIdx.x * 4 + 1] = c1 - c0;
        out[blockIdx.x * 4 + 2] = ((unsigned long long)xcc_id << 32) | hw_id;
        out[blockIdx.x * 4 + 3] = (unsigned long long)(w.x + w.y + mn != 12345.0f);
    }
}

// What ONE vector load costs a wave that is alone on its SIMD when nothing else on the CU has used the vector memory path for a
// while: `gap_bodies` bodies of arithmetic (16 steps each), then a global_load_dwordx4 of a line not touched before and the
// wait for it, timed with the shader clock.  gap_bodies = 0: the loads back to back.
__global__ void __launch_bounds__(64) k_load(const float4 *table, uint32_t mask_entries, int loads, int gap_bodies, unsigned long long *out)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w = {1e-8f * (threadIdx.x + 1), 2e-8f}, se = {1.0f, 1.0f}, dc = {1e-9f, -1e-9f};
    float mn = 1e30f, acc = 0.0f;
    unsigned long long load_cycles = 0, worst = 0;
    uint32_t idx = blockIdx.x * 977u;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < loads; i++) {
        int n = gap_bodies;
        if (n > 0)
            asm volatile("s_load_dwordx16 s[36:51], %[tb], 0x0\n\ts_load_dwordx16 s[52:67], %[tb], 0x40\n\ts_waitcnt lgkmcnt(0)\n"
                         ".Ld_%=:\n\t" LOWER UPPER "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 .Ld_%=\n\t"
                         : "+{v[48:49]}"(w), "+{v61}"(mn), [n] "+s"(n)
                         : "{v[40:41]}"(se), "{v[42:43]}"(dc), [tb] "s"(table)
                         : "v56", "v57", "v58", "v59", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46",
                           "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61",
                           "s62", "s63", "s64", "s65", "s66", "s67", "scc", "memory");
        idx = (idx + 131u) & mask_entries;
        const unsigned long long c0 = __builtin_readcyclecounter();
        float4 v;
        const float4 *p = table + idx;
        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        const unsigned long long c = __builtin_readcyclecounter() - c0;
        load_cycles += c;
        worst = c > worst ? c : worst;
        acc += v.x;
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0;
        out[blockIdx.x * 4 + 1] = load_cycles;
        out[blockIdx.x * 4 + 2] = worst;
        out[blockIdx.x * 4 + 3] = (unsigned long long)(w.x + w.y + mn + acc != 12345.0f);
    }
}

int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 64, bodies = 200000;
    const uint32_t table_bytes = 128 * 1024;
    float2 *table;
    unsigned long long *out;
    (void)hipMalloc(&table, table_bytes + 256);
    std::vector<float2> h((table_bytes + 256) / 8);
    for (size_t i = 0; i < h.size(); i++)
        h[i] = float2{0.5f + 1e-3f * (i % 7), -0.25f};
    (void)hipMemcpy(table, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, waves * 4 * sizeof(unsigned long long));
    std::vector<unsigned long long> r(waves * 4);
    for (int rep = 0; rep < 1; rep++)
        for (int mode = 0; mode < 5; mode++) {
            hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, table, table_bytes - 1, bodies, mode, out);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(r.data(), out, r.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> ns(waves);
            double mhz = 0;
            for (int w = 0; w < waves; w++) {
                ns[w] = r[w * 4] * 10.0 / ((double)bodies * 16);
                mhz += (double)r[w * 4 + 1] / (r[w * 4] / 100.0) / waves;
            }
            std::vector<double> s = ns;
            std::sort(s.begin(), s.end());
            int worst = (int)(std::max_element(ns.begin(), ns.end()) - ns.begin());
            const unsigned hw = (unsigned)(r[worst * 4 + 2] & 0xFFFFFFFFu);
            printf("{\"mode\": %d, \"rep\": %d, \"waves\": %d, \"ns_per_step_min\": %.2f, \"p25\": %.2f, \"median\": %.2f, \"p75\": %.2f, "
                   "\"p95\": %.2f, \"max\": %.2f, \"mean_shader_mhz\": %.0f, \"slowest_at_xcc_se_cu_simd\": [%u, %u, %u, %u]}\n",
                   mode, rep, waves, s[0], s[waves / 4], s[waves / 2], s[3 * waves / 4], s[(int)(waves * 0.95)], s[waves - 1], mhz,
                   (unsigned)(r[worst * 4 + 2] >> 32) & 0xF, (hw >> 13) & 7, (hw >> 8) & 0xF, (hw >> 4) & 3);
        }
    // the isolated vector load
    {
        const uint32_t entries = 1u << 16; // 1 MiB of float4
        float4 *t4;
        (void)hipMalloc(&t4, entries * sizeof(float4));
        (void)hipMemset(t4, 0, entries * sizeof(float4));
        const int gaps[] = {0, 1, 8, 40, 160};
        for (int g = 0; g < 5; g++) {
            const int loads = gaps[g] == 0 ? 20000 : (gaps[g] >= 40 ? 2000 : 8000);
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k_load, dim3(waves), dim3(64), 0, 0, t4, entries - 1, loads, gaps[g], out);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(r.data(), out, r.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> cyc(waves);
            double worst = 0;
            for (int w = 0; w < waves; w++) {
                cyc[w] = (double)r[w * 4 + 1] / loads;
                worst = std::max(worst, (double)r[w * 4 + 2]);
            }
            std::sort(cyc.begin(), cyc.end());
            printf("{\"vector_load_after_bodies_of_arithmetic\": %d, \"gap_us\": %.2f, \"waves\": %d, \"cycles_per_load_min\": %.0f, "
                   "\"median\": %.0f, \"max_wave_mean\": %.0f, \"worst_single_load\": %.0f}\n",
                   gaps[g], gaps[g] * 16 * 11.05e-3, waves, cyc[0], cyc[waves / 2], cyc[waves - 1], worst);
        }
    }
    return 0;
}
