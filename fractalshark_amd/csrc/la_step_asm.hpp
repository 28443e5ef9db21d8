// la_step_asm.hpp -- the HDRFloat<double> LA step for a wave whose lanes stand at ONE record, written by hand (round 6).
//
// k_lav2_hdr64's LA loop (LAReference.cpp's per-pixel walk through a stage, restated in kernels_hdr64.hip: la_body) issues ~100
// vector instructions per compiled step for the ~70 it needs: every value that lives across a loop a lane can leave is copied at
// the latch, because the compiler keeps a leaver's state by copying, not by the EXEC mask.  Here a lane that leaves the stage (or
// reaches the iteration cap) just drops out of EXEC with its registers as they are; the record comes through the scalar cache (four
// s_load for its 128 bytes and the next record's Ref) and its fields are scalar operands; dz moves between two register sets, X
// (the C++ variable) and Y, so that a step that cannot be finished here -- lanes at different records, a zero / subnormal product, a
// norm below 2^-1000 in the rebase test -- leaves BEFORE it has changed anything the compiled step reads, or (status 2) with only
// the rebase test left to do.  Same operations in the same order as la_body, which the counting build and every rare case still run.
//
// One statement = a loop of two half-steps (A: dz from X to Y, B: back).  Status: 0 every lane has left the stage or reached the cap;
// 1 (the lanes stand at different records) / 3 (a product Reduce's fast form does not cover) the lanes in `run` need one compiled step; 2 the lanes in `run` have taken the step up to dz' (in X) and j + 1, the rebase test
// is the compiled code's.  `left`: lanes that left the stage (RefIteration set from the record).
//
// Record layout (fs_la_hdr64_u32, 128 bytes, static_assert in kernels_hdr64.hip): Ref 0 / ZCoeff 24 / CCoeff 48 (re, im, e at +0, +8,
// +16), LAThreshold 72 (m, e at +0, +8), StepLength 120, NextStageLAIndex 124.  SGPRs after the loads:
//   s[36:37] Ref.re  s[38:39] Ref.im  s40 Ref.e   s[42:43] ZCoeff.re  s[44:45] ZCoeff.im  s46 ZCoeff.e  s[48:49] CCoeff.re  s[50:51] CCoeff.im
//   s52 CCoeff.e  s[54:55] LAThreshold.m  s56 LAThreshold.e  s60 StepLength  s61 NextStageLAIndex
//   s[64:65] next Ref.re  s[66:67] next Ref.im  s68 next Ref.e
//   operands: sx EXEC at entry, so record offset, m0 / m1 masks, sa exponent of 2 Ref
#pragma once

// (probe build, FS_H64_LA_ASM_DEBUG: tallies of the half-steps begun and of the general sums taken, in scalar registers)
#if FS_H64_LA_ASM_DEBUG
#define FS_ASM_CNT(N) "s_add_u32 %[c" #N "], %[c" #N "], 1\n\t"
#else
#define FS_ASM_CNT(N) ""
#endif

// HDRFloatComplex::plus_mutable for operands whose lanes do NOT agree on the arm: hi + lo 2^(lo.e - hi.e), the shift replaced by
// -4000 (m4k, a vector register: a literal and VCC are one constant-bus operand too many; the addend becomes a zero) from a gap of 120 on -- ldexp(a, sa) + ldexp(b, sb) with sa / sb = e - max(e_a, e_b), one of them 0.
// AR/AI/AE: first operand (scalar or vector), BR/BI/BE vector; OR/OI <- sum, OE <- its exponent; SA/SB, TA/TB scratch.
#define FS_LA_GENADD(AR, AI, AE, BR, BI, BE, OR_, OI, OE, SA, SB, TA, TB)                                           \
    "v_max_i32_e32 " OE ", " AE ", " BE "\n\t"                                                                      \
    "v_sub_u32_e32 " SA ", " AE ", " OE "\n\t"                                                                      \
    "v_sub_u32_e32 " SB ", " BE ", " OE "\n\t"                                                                      \
    "v_cmp_lt_i32_e32 vcc, 0xffffff88, " SA "\n\t"                                                                  \
    "s_nop 1\n\t"                                                                                                   \
    "v_cndmask_b32_e32 " SA ", %[m4k], " SA ", vcc\n\t"                                                         \
    "v_cmp_lt_i32_e32 vcc, 0xffffff88, " SB "\n\t"                                                                  \
    "s_nop 1\n\t"                                                                                                   \
    "v_cndmask_b32_e32 " SB ", %[m4k], " SB ", vcc\n\t"                                                         \
    "v_ldexp_f64 " TA ", " AR ", " SA "\n\t"                                                                        \
    "v_ldexp_f64 " TB ", " BR ", " SB "\n\t"                                                                        \
    "v_add_f64 " OR_ ", " TA ", " TB "\n\t"                                                                         \
    "v_ldexp_f64 " TA ", " AI ", " SA "\n\t"                                                                        \
    "v_ldexp_f64 " TB ", " BI ", " SB "\n\t"                                                                        \
    "v_add_f64 " OI ", " TA ", " TB "\n\t"

// dz of set Y -> set X for the lanes in EXEC
#define FS_LA_COPY_YX "v_mov_b64_e32 %[xr], %[yr]\n\tv_mov_b64_e32 %[xi], %[yi]\n\tv_mov_b32_e32 %[xe], %[ye]\n\t"

// One half-step.  S: label suffix; DIr/DIi/DIe the set dz is read from, DOr/DOi/DOe the set dz' is written to; CPI: FS_LA_COPY_YX when
// the INPUT set is Y (a lane that stops before dz' keeps dz: it has to be in X), CPO: FS_LA_COPY_YX when the OUTPUT set is Y.
#define FS_LA_HALF(S, DIr, DIi, DIe, DOr, DOi, DOe, CPI, CPO)                                                       \
    ".Lla_top" S "_%=:\n\t"                                                                                          \
    "v_lshl_add_u32 %[i0], %[j], 7, %[boff]\n\t"                                                                    \
    "s_nop 0\n\t" /* gfx940+: a readlane of a register the previous vector instruction wrote needs one wait state */ \
    "v_readfirstlane_b32 %[so], %[i0]\n\t"                                                                            \
    "s_nop 1\n\t"                                                                                                   \
    "v_cmp_eq_u32_e32 vcc, %[so], %[i0]\n\t"                                                                          \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lla_leave1" S "_%=\n\t"                                                                         \
    "s_load_dwordx16 s[36:51], %[las], %[so]\n\t"                                                                     \
    "s_load_dwordx8 s[52:59], %[las], %[so] offset:0x40\n\t"                                                          \
    "s_load_dwordx2 s[60:61], %[las], %[so] offset:0x78\n\t"                                                          \
    "s_load_dwordx8 s[64:71], %[las], %[so] offset:0x80\n\t"                                                          \
    "s_waitcnt lgkmcnt(0)\n\t" FS_ASM_CNT(0)                                                                        \
    /* (1) the step would pass the iteration limit: the lane leaves the stage */                                    \
    "v_add_u32_e32 %[i1], s60, %[it]\n\t"                                                                           \
    "v_cmp_ge_u32_e32 vcc, %[nit], %[i1]\n\t"                                                                       \
    "s_and_b64 %[m0], vcc, exec\n\t"                                                                             \
    "s_xor_b64 %[m1], %[m0], exec\n\t"                                                                        \
    "s_cbranch_scc0 .Lla_1ok" S "_%=\n\t"                                                                            \
    "s_mov_b64 exec, %[m1]\n\t"                                                                                  \
    "v_mov_b32_e32 %[refit], s61\n\t" CPI                                                                           \
    "s_or_b64 %[left], %[left], %[m1]\n\t"                                                                       \
    "s_mov_b64 exec, %[m0]\n\t"                                                                                  \
    "s_cbranch_execz .Lla_done_%=\n\t"                                                                               \
    ".Lla_1ok" S "_%=:\n\t"                                                                                          \
    /* (2), (3) p = dz (2 Ref + dz).  First the arm most steps of a deep zoom take: dz 120 binades and more below 2 Ref in every lane */ \
    /* (cur IS 2 Ref: the record's own values are the operands); then 2 Ref on top with the gap below 120; then the general sum */ \
    "s_max_i32 %[sa], s40, 0xefffffff\n\t"                                                                          \
    "s_add_i32 %[sa], %[sa], 1\n\t"                                                                                 \
    "v_subrev_u32_e32 %[i2], %[sa], " DIe "\n\t"                                                                    \
    "v_cmp_ge_i32_e32 vcc, 0xffffff88, %[i2]\n\t"                                                                   \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lla_Aother" S "_%=\n\t"                                                                         \
    "v_mul_f64 %[t2], " DIr ", s[36:37]\n\t"                                                                        \
    "v_mul_f64 %[t3], " DIi ", s[38:39]\n\t"                                                                        \
    "v_add_f64 %[t2], %[t2], -%[t3]\n\t"                                                                            \
    "v_mul_f64 %[t3], " DIr ", s[38:39]\n\t"                                                                        \
    "v_mul_f64 %[t4], " DIi ", s[36:37]\n\t"                                                                        \
    "v_add_f64 %[t3], %[t3], %[t4]\n\t"                                                                             \
    "v_add_u32_e32 %[i2], %[sa], " DIe "\n\t"                                                                       \
    ".Lla_Amulled" S "_%=:\n\t"                                                                                      \
    "v_max_i32_e32 %[i2], 0xf0000000, %[i2]\n\t"                                                                    \
    /* (4) Reduce: max(|re|, |im|) a normal number in every lane, or the compiled step takes over */               \
    "v_max_f64 %[t4], |%[t2]|, |%[t3]|\n\t"                                                                         \
    "v_cmp_class_f64_e64 vcc, %[t4], %[cls]\n\t"                                                                    \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lla_leave3" S "_%=\n\t"                                                                         \
    "v_frexp_exp_i32_f64_e32 %[i3], %[t4]\n\t"                                                                      \
    "v_sub_u32_e32 %[i4], 1, %[i3]\n\t"                                                                             \
    "v_ldexp_f64 %[t2], %[t2], %[i4]\n\t"                                                                           \
    "v_ldexp_f64 %[t3], %[t3], %[i4]\n\t"                                                                           \
    "v_add3_u32 %[i2], %[i2], %[i3], -1\n\t"                                                                        \
    "v_ldexp_f64 %[t4], %[t4], %[i4]\n\t"                                                                           \
    /* (5) usable: |newDz| (Chebyshev) below the record's threshold, else the lane leaves the stage */             \
    "v_cmp_gt_i32_e64 %[m0], s56, %[i2]\n\t"                                                                     \
    "v_cmp_eq_u32_e64 %[m1], s56, %[i2]\n\t"                                                                     \
    "v_cmp_gt_f64_e32 vcc, s[54:55], %[t4]\n\t"                                                                     \
    "s_and_b64 %[m1], %[m1], vcc\n\t"                                                                         \
    "s_or_b64 %[m0], %[m0], %[m1]\n\t"                                                                     \
    "s_xor_b64 %[m1], %[m0], exec\n\t"                                                                        \
    "s_cbranch_scc0 .Lla_5ok" S "_%=\n\t"                                                                            \
    "s_mov_b64 exec, %[m1]\n\t"                                                                                  \
    "v_mov_b32_e32 %[refit], s61\n\t" CPI                                                                           \
    "s_or_b64 %[left], %[left], %[m1]\n\t"                                                                       \
    "s_mov_b64 exec, %[m0]\n\t"                                                                                  \
    "s_cbranch_execz .Lla_done_%=\n\t"                                                                               \
    ".Lla_5ok" S "_%=:\n\t"                                                                                          \
    /* (6) the step is taken */                                                                                     \
    "v_mov_b32_e32 %[it], %[i1]\n\t"                                                                                \
    "v_add_u32_e32 %[nla], 1, %[nla]\n\t"                                                                           \
    /* (7) dz' = newDz ZCoeff + dc CCoeff: the exponents first -- with dc CCoeff 120 binades and more below in every lane (nine */ \
    /* steps of ten at C4's zoom) the second product is never formed */                                             \
    "v_add_u32_e32 %[i2], s46, %[i2]\n\t"                                                                           \
    "v_max_i32_e32 %[i2], 0xf0000000, %[i2]\n\t"                                                                    \
    "v_add_u32_e32 %[i3], s52, %[dce]\n\t"                                                                          \
    "v_max_i32_e32 %[i3], 0xf0000000, %[i3]\n\t"                                                                    \
    "v_sub_u32_e32 %[i4], %[i3], %[i2]\n\t"                                                                         \
    "v_cmp_ge_i32_e32 vcc, 0xffffff88, %[i4]\n\t"                                                                   \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lla_Sother" S "_%=\n\t"                                                                         \
    "v_mul_f64 " DOr ", %[t2], s[42:43]\n\t"                                                                        \
    "v_mul_f64 %[t0], %[t3], s[44:45]\n\t"                                                                          \
    "v_add_f64 " DOr ", " DOr ", -%[t0]\n\t"                                                                        \
    "v_mul_f64 " DOi ", %[t2], s[44:45]\n\t"                                                                        \
    "v_mul_f64 %[t0], %[t3], s[42:43]\n\t"                                                                          \
    "v_add_f64 " DOi ", " DOi ", %[t0]\n\t"                                                                         \
    "v_mov_b32_e32 " DOe ", %[i2]\n\t"                                                                              \
    ".Lla_Sback" S "_%=:\n\t"                                                                                        \
    /* (8) complex0 = next Ref + dz'; i4 <- complex0.e - dz'.e.  dz' 120 binades and more below in every lane: the next Ref itself */ \
    "v_subrev_u32_e32 %[i4], s68, " DOe "\n\t"                                                                      \
    "v_cmp_ge_i32_e32 vcc, 0xffffff88, %[i4]\n\t"                                                                   \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lla_Cother" S "_%=\n\t"                                                                         \
    "v_mov_b64_e32 %[t0], s[64:65]\n\t"                                                                             \
    "v_mov_b64_e32 %[t1], s[66:67]\n\t"                                                                             \
    "v_sub_u32_e32 %[i4], 0, %[i4]\n\t"                                                                             \
    ".Lla_Cback" S "_%=:\n\t"                                                                                        \
    /* (9) j + 1; rebase: |complex0| < |dz'| (Chebyshev norms, both above 2^-1000 or the compiled test decides) or the stage's end */ \
    "v_add_u32_e32 %[j], 1, %[j]\n\t"                                                                               \
    "v_max_f64 %[t2], |%[t0]|, |%[t1]|\n\t"                                                                         \
    "v_max_f64 %[t3], |" DOr "|, |" DOi "|\n\t"                                                                     \
    "v_min_f64 %[t4], %[t2], %[t3]\n\t"                                                                             \
    "v_cmp_le_f64_e32 vcc, %[tiny], %[t4]\n\t"                                                                      \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lla_leave2" S "_%=\n\t"                                                                         \
    "v_ldexp_f64 %[t2], %[t2], %[i4]\n\t"                                                                           \
    "v_cmp_lt_f64_e32 vcc, %[t2], %[t3]\n\t"                                                                        \
    "v_cmp_ge_u32_e64 %[m0], %[j], %[macro]\n\t"                                                                 \
    "s_or_b64 %[m0], %[m0], vcc\n\t"                                                                          \
    "s_and_b64 %[m0], %[m0], exec\n\t"                                                                        \
    "s_cbranch_scc0 .Lla_noreb" S "_%=\n\t"                                                                          \
    "s_mov_b64 %[m1], exec\n\t"                                                                                  \
    "s_mov_b64 exec, %[m0]\n\t"                                                                                  \
    "v_mov_b64_e32 " DOr ", %[t0]\n\t"                                                                              \
    "v_mov_b64_e32 " DOi ", %[t1]\n\t"                                                                              \
    "v_add_u32_e32 " DOe ", " DOe ", %[i4]\n\t"                                                                     \
    "v_mov_b32_e32 %[j], 0\n\t"                                                                                     \
    "s_mov_b64 exec, %[m1]\n\t"                                                                                  \
    ".Lla_noreb" S "_%=:\n\t"                                                                                        \
    /* (10) lanes at the iteration cap stop here (dz' has to be in X) */                                            \
    "v_cmp_gt_u32_e32 vcc, %[nit], %[it]\n\t"                                                                       \
    "s_and_b64 %[m0], vcc, exec\n\t"                                                                             \
    "s_xor_b64 %[m1], %[m0], exec\n\t"                                                                        \
    "s_cbranch_scc0 .Lla_next" S "_%=\n\t"                                                                           \
    "s_mov_b64 exec, %[m1]\n\t" CPO                                                                              \
    "s_mov_b64 exec, %[m0]\n\t"                                                                                  \
    "s_cbranch_execz .Lla_done_%=\n\t"                                                                               \
    "s_branch .Lla_next" S "_%=\n\t"                                                                                 \
    /* ---- out of line: the general sums */                                                                        \
    ".Lla_Aother" S "_%=:\n\t"                                                                                       \
    "v_add_u32_e32 %[i3], 0x77, %[i2]\n\t"                                                                          \
    "v_cmp_gt_u32_e32 vcc, 0x78, %[i3]\n\t"                                                                         \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lla_Agen" S "_%=\n\t"                                                                           \
    "v_ldexp_f64 %[t0], " DIr ", %[i2]\n\t"                                                                         \
    "v_ldexp_f64 %[t1], " DIi ", %[i2]\n\t"                                                                         \
    "v_add_f64 %[t0], s[36:37], %[t0]\n\t"                                                                          \
    "v_add_f64 %[t1], s[38:39], %[t1]\n\t"                                                                          \
    "v_add_u32_e32 %[i2], %[sa], " DIe "\n\t"                                                                       \
    ".Lla_Amul" S "_%=:\n\t"                                                                                         \
    "v_mul_f64 %[t2], " DIr ", %[t0]\n\t"                                                                           \
    "v_mul_f64 %[t3], " DIi ", %[t1]\n\t"                                                                           \
    "v_add_f64 %[t2], %[t2], -%[t3]\n\t"                                                                            \
    "v_mul_f64 %[t3], " DIr ", %[t1]\n\t"                                                                           \
    "v_mul_f64 %[t4], " DIi ", %[t0]\n\t"                                                                           \
    "v_add_f64 %[t3], %[t3], %[t4]\n\t"                                                                             \
    "s_branch .Lla_Amulled" S "_%=\n\t"                                                                              \
    ".Lla_Agen" S "_%=:\n\t" FS_ASM_CNT(1)                                                                           \
    "v_mov_b32_e32 %[i5], %[sa]\n\t"                                                                                \
    FS_LA_GENADD("s[36:37]", "s[38:39]", "%[i5]", DIr, DIi, DIe, "%[t0]", "%[t1]", "%[i3]", "%[i4]", "%[i2]", "%[t4]", "%[t5]") \
    "v_add_u32_e32 %[i2], %[i3], " DIe "\n\t"                                                                       \
    "s_branch .Lla_Amul" S "_%=\n\t"                                                                                 \
    ".Lla_Sother" S "_%=:\n\t"                                                                                       \
    "v_mul_f64 %[t0], %[t2], s[42:43]\n\t"                                                                          \
    "v_mul_f64 %[t1], %[t3], s[44:45]\n\t"                                                                          \
    "v_add_f64 %[t0], %[t0], -%[t1]\n\t"                                                                            \
    "v_mul_f64 %[t1], %[t2], s[44:45]\n\t"                                                                          \
    "v_mul_f64 %[t5], %[t3], s[42:43]\n\t"                                                                          \
    "v_add_f64 %[t1], %[t1], %[t5]\n\t"                                                                             \
    "v_mul_f64 %[t2], %[dcr], s[48:49]\n\t"                                                                         \
    "v_mul_f64 %[t3], %[dci], s[50:51]\n\t"                                                                         \
    "v_add_f64 %[t2], %[t2], -%[t3]\n\t"                                                                            \
    "v_mul_f64 %[t3], %[dcr], s[50:51]\n\t"                                                                         \
    "v_mul_f64 %[t5], %[dci], s[48:49]\n\t"                                                                         \
    "v_add_f64 %[t3], %[t3], %[t5]\n\t"                                                                             \
    "v_add_u32_e32 %[i5], 0x77, %[i4]\n\t"                                                                          \
    "v_cmp_gt_u32_e32 vcc, 0x78, %[i5]\n\t"                                                                         \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lla_Sgen" S "_%=\n\t"                                                                           \
    "v_ldexp_f64 %[t2], %[t2], %[i4]\n\t"                                                                           \
    "v_ldexp_f64 %[t3], %[t3], %[i4]\n\t"                                                                           \
    "v_add_f64 " DOr ", %[t0], %[t2]\n\t"                                                                           \
    "v_add_f64 " DOi ", %[t1], %[t3]\n\t"                                                                           \
    "v_mov_b32_e32 " DOe ", %[i2]\n\t"                                                                              \
    "s_branch .Lla_Sback" S "_%=\n\t"                                                                                \
    ".Lla_Sgen" S "_%=:\n\t" FS_ASM_CNT(2)                                                                           \
    FS_LA_GENADD("%[t0]", "%[t1]", "%[i2]", "%[t2]", "%[t3]", "%[i3]", DOr, DOi, DOe, "%[i4]", "%[i5]", "%[t4]", "%[t5]") \
    "s_branch .Lla_Sback" S "_%=\n\t"                                                                                \
    ".Lla_Cother" S "_%=:\n\t"                                                                                       \
    "v_add_u32_e32 %[i5], 0x77, %[i4]\n\t"                                                                          \
    "v_cmp_gt_u32_e32 vcc, 0x78, %[i5]\n\t"                                                                         \
    "s_cmp_lg_u64 vcc, exec\n\t"                                                                                    \
    "s_cbranch_scc1 .Lla_Cgen" S "_%=\n\t"                                                                           \
    "v_ldexp_f64 %[t0], " DOr ", %[i4]\n\t"                                                                         \
    "v_ldexp_f64 %[t1], " DOi ", %[i4]\n\t"                                                                         \
    "v_add_f64 %[t0], s[64:65], %[t0]\n\t"                                                                          \
    "v_add_f64 %[t1], s[66:67], %[t1]\n\t"                                                                          \
    "v_sub_u32_e32 %[i4], 0, %[i4]\n\t"                                                                             \
    "s_branch .Lla_Cback" S "_%=\n\t"                                                                                \
    ".Lla_Cgen" S "_%=:\n\t" FS_ASM_CNT(3)                                                                           \
    "v_mov_b32_e32 %[i5], s68\n\t"                                                                                  \
    FS_LA_GENADD("s[64:65]", "s[66:67]", "%[i5]", DOr, DOi, DOe, "%[t0]", "%[t1]", "%[i3]", "%[i4]", "%[i2]", "%[t4]", "%[t5]") \
    "v_sub_u32_e32 %[i4], %[i3], " DOe "\n\t"                                                                       \
    "s_branch .Lla_Cback" S "_%=\n\t"                                                                                \
    /* ---- out of line: the exits of this half */                                                                  \
    ".Lla_leave1" S "_%=:\n\t" CPI                                                                                   \
    "s_branch .Lla_leave1_%=\n\t"                                                                                    \
    ".Lla_leave3" S "_%=:\n\t" CPI                                                                                   \
    "s_branch .Lla_leave3_%=\n\t"                                                                                    \
    ".Lla_leave2" S "_%=:\n\t" CPO                                                                                   \
    "s_branch .Lla_leave2_%=\n\t"                                                                                    \
    ".Lla_next" S "_%=:\n\t"

// The statement.  (Half B's ".Lla_next" falls through to the loop's back edge.)
#define FS_LA_UNIFORM_LOOP                                                                                          \
    "s_mov_b64 %[sx], exec\n\t"                                                                                  \
    "s_mov_b64 %[left], 0\n\t"                                                                                      \
    FS_LA_HALF("A", "%[xr]", "%[xi]", "%[xe]", "%[yr]", "%[yi]", "%[ye]", "", FS_LA_COPY_YX)                       \
    FS_LA_HALF("B", "%[yr]", "%[yi]", "%[ye]", "%[xr]", "%[xi]", "%[xe]", FS_LA_COPY_YX, "")                       \
    "s_branch .Lla_topA_%=\n\t"                                                                                      \
    ".Lla_leave1_%=:\n\t"                                                                                            \
    "s_mov_b32 %[st], 1\n\t"                                                                                        \
    "s_mov_b64 %[run], exec\n\t"                                                                                    \
    "s_branch .Lla_out_%=\n\t"                                                                                       \
    ".Lla_leave3_%=:\n\t"                                                                                            \
    "s_mov_b32 %[st], 3\n\t"                                                                                        \
    "s_mov_b64 %[run], exec\n\t"                                                                                    \
    "s_branch .Lla_out_%=\n\t"                                                                                       \
    ".Lla_leave2_%=:\n\t"                                                                                            \
    "s_mov_b32 %[st], 2\n\t"                                                                                        \
    "s_mov_b64 %[run], exec\n\t"                                                                                    \
    "s_branch .Lla_out_%=\n\t"                                                                                       \
    ".Lla_done_%=:\n\t"                                                                                              \
    "s_mov_b32 %[st], 0\n\t"                                                                                        \
    "s_mov_b64 %[run], 0\n\t"                                                                                       \
    ".Lla_out_%=:\n\t"                                                                                               \
    "s_mov_b64 exec, %[sx]\n\t"
