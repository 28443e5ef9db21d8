// pt_step_asm.hpp -- the HDRFloat<double> perturbation step written by hand (round 6), in the manner of la_step_asm.hpp.
//
// k_lav2_hdr64's perturbation loop (the reference's per-pixel loop, restated in kernels_hdr64.hip: pt_step) compiles to ~76 vector
// instructions per step; the step needs ~52.  The difference is bookkeeping the compiler cannot avoid in a loop lanes leave at
// different times: every live value is copied at the latch.  Here an escaped lane -- or one at the iteration cap -- drops out of EXEC
// and keeps its registers; dz moves between two register sets (X, the C++ variable, and Y) and the orbit entry between two more
// (P, the C++ variable, and Q), their roles exchanged from one half-step to the next, so nothing is copied in the loop.
//
// What is done here is the step of a wave whose lanes agree: 2 Z + dz and Z' + dz' with the orbit value on top (or, out of line, the
// general sum), dc 120 binades and more below dz (2 Z + dz) in every lane (dz t + dc IS dz t; else the general sum again), Reduce on normal numbers, both squared
// norms above 2^-1000.  Anything else leaves the statement: status 1 / 3 before the step has changed anything (one compiled step
// follows), status 2 with dz' in X, zoff and the count as they were (the compiled code does
// the tests, the rebase and the count).  Status 0: every lane has escaped or reached the cap.
#pragma once

#include "la_step_asm.hpp" /* FS_LA_GENADD */

#define FS_PT_COPY_YX "v_mov_b64_e32 %[xr], %[yr]\n\tv_mov_b64_e32 %[xi], %[yi]\n\tv_mov_b32_e32 %[xe], %[ye]\n\t"
// The orbit entries live in NAMED registers -- P = v[52:56], Q = v[58:62] (tuples start on even registers): one 16-byte load and one 4-byte load per entry need four
// consecutive registers, which operands cannot promise -- and do not cross the statement's boundary: it loads the entry at zoff when
// it starts, and the compiled code that follows an exit reads the entry it needs itself.

// One half-step.  S: label suffix; DI*: dz in; DO*: dz out; ZH*: the entry the step leaves from; ZN*: the entry it arrives at (loaded
// here); CPI: copies for an exit BEFORE the step (input sets back to X / P), CPO: copies for the status-2 exit (output sets to X / P).
#define FS_PT_HALF(S, DIr, DIi, DIe, DOr, DOi, DOe, ZHr, ZHi, ZHe, ZN4, ZNr, ZNi, ZNe, CPI, CPO)                         \
    ".Lpt_top" S "_%=:\n\t"                                                                                          \
    "global_load_dwordx4 " ZN4 ", %[zoff], %[zb] offset:32\n\t"                                                     \
    "global_load_dword " ZNe ", %[zoff], %[zb] offset:48\n\t" FS_ASM_CNT(0)                                         \
    /* cur = 2 Z + dz */                                                                                            \
    "v_add_u32_e32 %[i0], 1, " ZHe "\n\t"                                                                           \
    "v_sub_u32_e32 %[i1], " DIe ", %[i0]\n\t"                                                                       \
    "v_add_u32_e32 %[i2], 0x77, %[i1]\n\t"                                                                          \
    "v_cmp_gt_u32_e32 vcc, 0x78, %[i2]\n\t"                                                                         \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lpt_Agen" S "_%=\n\t"                                                                           \
    "v_ldexp_f64 %[t0], " DIr ", %[i1]\n\t"                                                                         \
    "v_ldexp_f64 %[t1], " DIi ", %[i1]\n\t"                                                                         \
    "v_add_f64 %[t0], " ZHr ", %[t0]\n\t"                                                                           \
    "v_add_f64 %[t1], " ZHi ", %[t1]\n\t"                                                                           \
    "v_add_u32_e32 %[i1], " DIe ", %[i0]\n\t"                                                                       \
    ".Lpt_Aback" S "_%=:\n\t"                                                                                        \
    /* p = dz cur, exponent clamped */                                                                              \
    "v_mul_f64 %[t2], " DIr ", %[t0]\n\t"                                                                           \
    "v_mul_f64 %[t3], " DIi ", %[t1]\n\t"                                                                           \
    "v_add_f64 %[t2], %[t2], -%[t3]\n\t"                                                                            \
    "v_mul_f64 %[t3], " DIr ", %[t1]\n\t"                                                                           \
    "v_mul_f64 %[t4], " DIi ", %[t0]\n\t"                                                                           \
    "v_add_f64 %[t3], %[t3], %[t4]\n\t"                                                                             \
    "v_max_i32_e32 %[i1], 0xf0000000, %[i1]\n\t"                                                                    \
    /* + dc: 120 binades and more below p in every lane (a deep zoom: dz t + dc IS dz t), or the general sum */    \
    "v_sub_u32_e32 %[i2], %[dce], %[i1]\n\t"                                                                        \
    "v_cmp_ge_i32_e32 vcc, 0xffffff88, %[i2]\n\t"                                                                   \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lpt_Bgen" S "_%=\n\t"                                                                          \
    ".Lpt_Bback" S "_%=:\n\t"                                                                                       \
    /* dz' = Reduce(p) */                                                                                           \
    "v_max_f64 %[t4], |%[t2]|, |%[t3]|\n\t"                                                                         \
    "v_cmp_class_f64_e64 vcc, %[t4], %[cls]\n\t"                                                                    \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lpt_leave3" S "_%=\n\t"                                                                         \
    "v_frexp_exp_i32_f64_e32 %[i2], %[t4]\n\t"                                                                      \
    "v_sub_u32_e32 %[i3], 1, %[i2]\n\t"                                                                             \
    "v_ldexp_f64 " DOr ", %[t2], %[i3]\n\t"                                                                         \
    "v_ldexp_f64 " DOi ", %[t3], %[i3]\n\t"                                                                         \
    "v_add3_u32 " DOe ", %[i1], %[i2], -1\n\t"                                                                      \
    /* complex0 = Z' + dz' -> t0, t1; i2 <- 2 complex0.e, i3 <- 2 (complex0.e - dz'.e) */                           \
    "s_waitcnt vmcnt(0)\n\t"                                                                                        \
    "v_sub_u32_e32 %[i1], " DOe ", " ZNe "\n\t"                                                                     \
    "v_add_u32_e32 %[i2], 0x77, %[i1]\n\t"                                                                          \
    "v_cmp_gt_u32_e32 vcc, 0x78, %[i2]\n\t"                                                                         \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lpt_Cgen" S "_%=\n\t"                                                                           \
    "v_ldexp_f64 %[t0], " DOr ", %[i1]\n\t"                                                                         \
    "v_ldexp_f64 %[t1], " DOi ", %[i1]\n\t"                                                                         \
    "v_add_f64 %[t0], " ZNr ", %[t0]\n\t"                                                                           \
    "v_add_f64 %[t1], " ZNi ", %[t1]\n\t"                                                                           \
    "v_lshlrev_b32_e32 %[i2], 1, " ZNe "\n\t"                                                                       \
    "v_sub_u32_e32 %[i3], " ZNe ", " DOe "\n\t"                                                                     \
    "v_lshlrev_b32_e32 %[i3], 1, %[i3]\n\t"                                                                         \
    ".Lpt_Cback" S "_%=:\n\t"                                                                                        \
    /* n1 = |complex0|^2 -> t2, n2 = |dz'|^2 -> t3; both >= 2^-1000 in every lane, or the compiled tests */        \
    "v_mul_f64 %[t2], %[t0], %[t0]\n\t"                                                                             \
    "v_mul_f64 %[t3], %[t1], %[t1]\n\t"                                                                             \
    "v_add_f64 %[t2], %[t2], %[t3]\n\t"                                                                             \
    "v_mul_f64 %[t3], " DOr ", " DOr "\n\t"                                                                         \
    "v_mul_f64 %[t4], " DOi ", " DOi "\n\t"                                                                         \
    "v_add_f64 %[t3], %[t3], %[t4]\n\t"                                                                             \
    "v_min_f64 %[t4], %[t2], %[t3]\n\t"                                                                             \
    "v_cmp_le_f64_e32 vcc, %[tiny], %[t4]\n\t"                                                                      \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lpt_leave2" S "_%=\n\t"                                                                         \
    "v_ldexp_f64 %[t4], %[t2], %[i2]\n\t"                                                                           \
    "v_cmp_lt_f64_e64 %[mesc], %[c256], %[t4]\n\t" /* escaped: 256 < |z|^2 */                                      \
    "v_ldexp_f64 %[t4], %[t2], %[i3]\n\t"                                                                           \
    "v_cmp_lt_f64_e64 %[mreb], %[t4], %[t3]\n\t" /* |z|^2 < |dz|^2 */                                              \
    "v_add_u32_e32 %[i1], 32, %[zoff]\n\t"                                                                          \
    "v_cmp_le_u32_e64 %[mend], %[maxoff], %[i1]\n\t" /* arrived at the orbit's last entry */                      \
    "s_or_b64 %[mreb], %[mreb], %[mend]\n\t"                                                                        \
    "s_andn2_b64 %[mreb], %[mreb], %[mesc]\n\t" /* (an escaped lane does not rebase) */                            \
    "s_and_b64 %[mreb], %[mreb], exec\n\t"                                                                          \
    "s_cbranch_scc0 .Lpt_noreb" S "_%=\n\t" FS_ASM_CNT(2)                                                            \
    /* rebase: dz' = Reduce(complex0), the orbit from its first entry */                                            \
    "s_mov_b64 %[mend], exec\n\t"                                                                                   \
    "s_mov_b64 exec, %[mreb]\n\t"                                                                                   \
    "v_max_f64 %[t4], |%[t0]|, |%[t1]|\n\t"                                                                         \
    "v_cmp_class_f64_e64 vcc, %[t4], %[cls]\n\t"                                                                    \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_mov_b64 exec, %[mend]\n\t"                                                                                   \
    "s_cbranch_scc1 .Lpt_leave2" S "_%=\n\t"                                                                         \
    "s_mov_b64 exec, %[mreb]\n\t"                                                                                   \
    "v_frexp_exp_i32_f64_e32 %[i1], %[t4]\n\t"                                                                      \
    "v_sub_u32_e32 %[i3], 1, %[i1]\n\t"                                                                             \
    "v_ldexp_f64 " DOr ", %[t0], %[i3]\n\t"                                                                         \
    "v_ldexp_f64 " DOi ", %[t1], %[i3]\n\t"                                                                         \
    "v_ashrrev_i32_e32 " DOe ", 1, %[i2]\n\t"                                                                       \
    "v_add3_u32 " DOe ", " DOe ", %[i1], -1\n\t"                                                                    \
    "v_mov_b64_e32 " ZNr ", %[z0re]\n\t"                                                                            \
    "v_mov_b64_e32 " ZNi ", %[z0im]\n\t"                                                                            \
    "v_mov_b32_e32 " ZNe ", %[z0e]\n\t"                                                                             \
    "v_mov_b32_e32 %[zoff], 0xffffffe0\n\t"                                                                         \
    "s_mov_b64 exec, %[mend]\n\t"                                                                                   \
    ".Lpt_noreb" S "_%=:\n\t"                                                                                        \
    "v_add_u32_e32 %[zoff], 32, %[zoff]\n\t"                                                                        \
    "s_andn2_b64 exec, exec, %[mesc]\n\t" /* escaped lanes leave with their count */                               \
    "v_add_u32_e32 %[iter], 1, %[iter]\n\t"                                                                         \
    "v_cmp_gt_u32_e32 vcc, %[niter], %[iter]\n\t" /* lanes at the cap leave with it */                             \
    "s_and_b64 exec, exec, vcc\n\t"                                                                                 \
    "s_cbranch_execz .Lpt_done_%=\n\t"                                                                               \
    "s_branch .Lpt_next" S "_%=\n\t"                                                                                 \
    /* ---- out of line: the general sums */                                                                        \
    ".Lpt_Agen" S "_%=:\n\t" FS_ASM_CNT(1)                                                                           \
    FS_LA_GENADD(ZHr, ZHi, "%[i0]", DIr, DIi, DIe, "%[t0]", "%[t1]", "%[i3]", "%[i1]", "%[i2]", "%[t2]", "%[t3]")   \
    "v_add_u32_e32 %[i1], " DIe ", %[i3]\n\t"                                                                       \
    "s_branch .Lpt_Aback" S "_%=\n\t"                                                                                \
    ".Lpt_Bgen" S "_%=:\n\t"                                                                                        \
    FS_LA_GENADD("%[t2]", "%[t3]", "%[i1]", "%[dcr]", "%[dci]", "%[dce]", "%[t0]", "%[t1]", "%[i0]", "%[i2]", "%[i3]", "%[t4]", "%[t2]") \
    "v_mov_b64_e32 %[t2], %[t0]\n\t"                                                                                \
    "v_mov_b64_e32 %[t3], %[t1]\n\t"                                                                                \
    "v_mov_b32_e32 %[i1], %[i0]\n\t"                                                                                \
    "s_branch .Lpt_Bback" S "_%=\n\t"                                                                               \
    ".Lpt_Cgen" S "_%=:\n\t" FS_ASM_CNT(3)                                                                           \
    FS_LA_GENADD(ZNr, ZNi, ZNe, DOr, DOi, DOe, "%[t0]", "%[t1]", "%[i0]", "%[i1]", "%[i2]", "%[t2]", "%[t3]")       \
    "v_lshlrev_b32_e32 %[i2], 1, %[i0]\n\t"                                                                         \
    "v_sub_u32_e32 %[i3], %[i0], " DOe "\n\t"                                                                       \
    "v_lshlrev_b32_e32 %[i3], 1, %[i3]\n\t"                                                                         \
    "s_branch .Lpt_Cback" S "_%=\n\t"                                                                                \
    /* ---- out of line: the exits of this half */                                                                  \
    /* (the loads of this half are still on their way at the first two, and in half B they write P: wait before P is restored) */ \
    ".Lpt_leave1" S "_%=:\n\t"                                                                                       \
    "s_waitcnt vmcnt(0)\n\t" CPI                                                                                    \
    "s_branch .Lpt_leave1_%=\n\t"                                                                                    \
    ".Lpt_leave3" S "_%=:\n\t"                                                                                       \
    "s_waitcnt vmcnt(0)\n\t" CPI                                                                                    \
    "s_branch .Lpt_leave3_%=\n\t"                                                                                    \
    ".Lpt_leave2" S "_%=:\n\t" CPO                                                                                   \
    "s_branch .Lpt_leave2_%=\n\t"                                                                                    \
    ".Lpt_next" S "_%=:\n\t"

#define FS_PT_LOOP                                                                                                  \
    "s_mov_b64 %[sx], exec\n\t"                                                                                     \
    "s_mov_b64 exec, %[run]\n\t"                                                                                    \
    "global_load_dwordx4 v[52:55], %[zoff], %[zb]\n\t"                                                              \
    "global_load_dword v56, %[zoff], %[zb] offset:16\n\t"                                                           \
    "s_waitcnt vmcnt(0)\n\t"                                                                                        \
    FS_PT_HALF("A", "%[xr]", "%[xi]", "%[xe]", "%[yr]", "%[yi]", "%[ye]", "v[52:53]", "v[54:55]", "v56", "v[58:61]", "v[58:59]",   \
               "v[60:61]", "v62", "", FS_PT_COPY_YX)                                                                \
    FS_PT_HALF("B", "%[yr]", "%[yi]", "%[ye]", "%[xr]", "%[xi]", "%[xe]", "v[58:59]", "v[60:61]", "v62", "v[52:55]", "v[52:53]",   \
               "v[54:55]", "v56", FS_PT_COPY_YX, "")                                                                \
    "s_branch .Lpt_topA_%=\n\t"                                                                                      \
    ".Lpt_leave1_%=:\n\t"                                                                                            \
    "s_mov_b32 %[st], 1\n\t"                                                                                        \
    "s_branch .Lpt_leave_%=\n\t"                                                                                     \
    ".Lpt_leave3_%=:\n\t"                                                                                            \
    "s_mov_b32 %[st], 3\n\t"                                                                                        \
    "s_branch .Lpt_leave_%=\n\t"                                                                                     \
    ".Lpt_leave2_%=:\n\t"                                                                                            \
    "s_mov_b32 %[st], 2\n\t"                                                                                        \
    ".Lpt_leave_%=:\n\t"                                                                                             \
    "s_waitcnt vmcnt(0)\n\t"                                                                                        \
    "s_mov_b64 %[run], exec\n\t"                                                                                    \
    "s_branch .Lpt_out_%=\n\t"                                                                                       \
    ".Lpt_done_%=:\n\t"                                                                                              \
    "s_mov_b32 %[st], 0\n\t"                                                                                        \
    "s_mov_b64 %[run], 0\n\t"                                                                                       \
    ".Lpt_out_%=:\n\t"                                                                                               \
    "s_mov_b64 exec, %[sx]\n\t"
