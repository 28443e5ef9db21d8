// scaled_runs.hpp -- what the two HDRFloat<float> perturbation kernels share (round 6: moved out of kernels.hip when each kernel got
// a translation unit of its own -- kernels_lav2_hdr32.hip: k_lav2_hdr32_fast; kernels_perturb.hip: k_perturb_scalar): the step pieces
// of the scaled runs, the hand-scheduled untested loops (FS_FAST_LOOP_*), the run-length votes and the small key / scale helpers.
// One definition of each; the text is the one rounds 3-5 wrote (DESIGN_APPENDIX.md 4.2, 4.3).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hdr_math.hpp"
#include "kernels.h"

namespace {

// max / min of two magnitudes as ONE instruction (source modifiers).  Written as fmaxf(fabsf(a), fabsf(b)) the compiler first
// canonicalises each operand (v_max_f32 |a|, |a| -- quieting a signalling NaN no arithmetic of this file can produce): three
// instructions instead of one at every run entry, run exit and tested step.  A quiet NaN in one operand returns the other, as
// fmaxf / fminf do.
static __device__ __forceinline__ float fs_max_abs(float a, float b)
{
    float r;
    asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
static __device__ __forceinline__ float fs_min_abs(float a, float b)
{
    float r;
    asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

} // namespace

namespace {
// A scaled run may start at entry e of the second companion array: a usable entry (bound not "never") or an exact zero.
__device__ __forceinline__ bool scaled_startable(const float4 e)
{
    return __float_as_int(e.z) != (int)0x80000000 || (e.x == 0.0f && e.y == 0.0f);
}
} // namespace

// ------------------------------------------------------------------------------------------------
// The scalar-cache path of the scaled runs (k_lav2_hdr32_fast and the perturbation-only float path of k_perturb_scalar):
// step pieces of the tested C++ block and the hand-scheduled untested loop.  Names used from the enclosing scope: sE2, dcs,
// Esh, imdc, wv, mxS, pwi, zS, off, zpb, lim8 (and the asm's outputs).
// PF of the loops (FS_FAST_LOOP_FL / _FD below): FS_PF_NONE, or FS_PF_NEXT_BODY = one dword of each 64-byte line of the NEXT body's entries (three:
// entries are 16-byte aligned only), requested right after this body's wait, so that the next body's loads hit the scalar
// cache -- for waves that run alone on their SIMD (C2's interior pixels), where the L2 round trip per body is not hidden.
#define FS_PF_NONE ""
#define FS_PF_NEXT_BODY                                                                                             \
    "s_load_dword %[pf], s[68:69], %[off] offset:0x80\n\t"                                                          \
    "s_load_dword %[pg], s[68:69], %[off] offset:0xc0\n\t"                                                          \
    "s_load_dword %[ph], s[68:69], %[off] offset:0xfc\n\t"
#define FS_STEP_ARITH(W_, Z_, NW_, T)                                                                               \
    const f2 s_##T = __builtin_elementwise_fma(W_, sE2, Z_);                                                        \
    const f2 pa_##T = W_.xx * s_##T;                                                                                \
    const f2 pb_##T = W_.yy * s_##T.yx;                                                                             \
    f2 p_##T;                                                                                                       \
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(p_##T) : "v"(pa_##T), "v"(pb_##T));              \
    NW_ = p_##T + dcs;
#define FS_STEP_BOUND(NW_, T, V, EB)                                                                                \
    const float mx_##T = fs_max_abs(NW_.x, NW_.y);                           \
    V |= __builtin_amdgcn_ballot_w64(__float_as_int(mx_##T) + Esh > __float_as_int(EB));
// One scaled step with its acceptance tests, and the per-lane entry load of the runs whose lanes sit at different orbit
// positions: shared by k_lav2_hdr32_fast and the perturbation-only float path of k_perturb_scalar (one definition; round 4 had
// two identical copies).  Names from the enclosing scope as listed above, plus lane_off / zp for the load.
#define FS_SCALED_STEP(W_, Z_, NW_, NZ_, T, V, FULL, AFTER_ARITH, EX, EY, EB)                                       \
    const f2 s_##T = __builtin_elementwise_fma(W_, sE2, Z_);                                                        \
    const f2 pa_##T = W_.xx * s_##T;                                                                                \
    const f2 pb_##T = W_.yy * s_##T.yx;                                                                             \
    f2 p_##T;                                                                                                       \
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(p_##T) : "v"(pa_##T), "v"(pb_##T));              \
    NW_ = p_##T + dcs;                                                                                              \
    float mx_##T = fs_max_abs(NW_.x, NW_.y);                                 \
    AFTER_ARITH;                                                                                                    \
    NZ_ = (f2){EX, EY};                                                                                             \
    V |= __builtin_amdgcn_ballot_w64(__float_as_int(mx_##T) + Esh > __float_as_int(EB));                            \
    if (FULL) {                                                                                                     \
        FS_STEP_FLOOR(NW_, V)                                                                                       \
        V |= __builtin_amdgcn_ballot_w64(!(mx_##T < FS_FL_HIGH_TRIP));                                              \
    } else {                                                                                                        \
        FS_STEP_FLOOR_FIRST(NW_, V)                                                                                 \
    }
#define FS_SCALED_LOAD(OFS, T, PIN)                                                                                 \
    asm volatile("global_load_dwordx3 %0, %2, %3 offset:" OFS : "=v"(ent_##T), "+v"(PIN) : "v"(lane_off), "s"(zp));
// The untested body.  Registers are named (the halves of a packed pair have no operand syntax): the state w in v[48:49];
// four state pairs v[48:55] in rotation (a trip = two steps: start state, first step, and the next trip's two while the
// verdict is pending); the entries in s[36:67].  A packed result cannot be read by the next instruction, so each trip's
// tests run in the wait states of the following trip's packed arithmetic, and its verdict arrives just before that trip's
// second step is written over the failed trip's start state: everything a failed trip needs is still in its registers,
// and what was computed past it is dropped.
// The loop runs inside the statement: bodies of eight steps while the block test passes and eight steps are left
// (status 0 on the way out: state in v[48:49], max|w| in v60, `off` = 16 bytes per step taken so far, s[64:65] / s67 =
// 2Z / block bound of the entry the state is at); status 1 / 2: the first / second trip of a block failed (start state /
// first step: v48 / v50, v52 / v54; `off` counts the steps before the trip, `eb` = the first arrival's bound).  The
// tests of a body's LAST trip run in the wait states of the next body's first two steps -- or on the way out.
#define FS_PK_F(W, Z) "v_pk_fma_f32 v[56:57], " W ", %[se], " Z "\n\t"
#define FS_PK_MA(W) "v_pk_mul_f32 v[58:59], " W ", v[56:57] op_sel_hi:[0,1]\n\t"
#define FS_PK_MB(W) "v_pk_mul_f32 v[56:57], " W ", v[56:57] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
#define FS_PK_P "v_pk_add_f32 v[58:59], v[58:59], v[56:57] neg_lo:[0,1] neg_hi:[0,0]\n\t"
#define FS_PK_A(NW) "v_pk_add_f32 " NW ", v[58:59], %[dc]\n\t"
#define FS_R0 "v[48:49]"
#define FS_R1 "v[50:51]"
#define FS_R2 "v[52:53]"
#define FS_R3 "v[54:55]"
#define FS_T_X(A, B) "v_max_f32_e64 v60, |" A "|, |" B "|\n\t"

// ------------------------------------------------------------------------------------------------
// Round 4: the FLOOR form of the scaled runs' acceptance tests (k_lav2_hdr32_fast, and k_perturb_scalar's float path -- whose
// simpler version of the argument is given there).  Scale of a run: w = dz 2^-E with E = dz's exponent + 24, i.e. max|w| starts in [2^-24, 2^-23).
//
// Why a scaled step can differ from the reference's HDRFloatComplex step at all (Fractal.cpp:2646-2661: cur = 2Z + dz,
// p = dz cur, q = p + dc, Reduce): both carry out the same IEEE operations on the same real operands (the scale is a
// power of two), so the results agree bit for bit UNLESS
//   (u) an operation underflows -- its result is below 2^-126 in the units it is carried out in and loses bits -- in one
//       of the two arithmetics: the scaled one (units 2^E), or the reference's mantissa arithmetic (products in units of
//       2^(dz.e + cur.e), the aligned operand of an addition in the units of the larger one, Reduce's re-scaling by up to
//       2^-4).  Such an event injects an absolute error below 2^-126 in ITS units; in the run's units that is below
//       2^(-126 + 35): the exponent of a product's units is that of max|w| (kept below 29, see H) plus that of 2Z + dz
//       (< 5), and dc's units are at most 2^7 (start condition).  Two roundings lie between the event and a part of the
//       new state (p = a - b, q = p + dc), each amplifies the error by at most 2^26 relative to the result it rounds, so a
//       part of q that is wrong because of it is smaller than 2^(-91 + 29.2) < 2^-61;
//   (d) the reference DROPS the smaller operand of q = p + dc when the exponents are 120 or more apart
//       (HDRFloatComplex::plus_mutable) while the scaled step adds it: a part of q can differ only where the dropped
//       operand is within 2^26 of the kept one's part, and then that part of q is below 2^(28 + exponent of the kept
//       operand's units - 120) <= 2^-57 (dc dropped: units of p, < 2^35) or far below (p dropped: dc's units <= 2^7).
//       (cur = 2Z + dz: a dropped dz is more than 2^79 below either part of a usable orbit entry -- the companion's
//       "usable" test -- and changes no bit.)
// Hence: a new state whose TWO parts are both at least 2^-56 in magnitude (the floor, F) is the reference's state, bit for
// bit, provided the state it was stepped from was (induction) and had max|w| < 2^29 (H).  Every state is tested against
// the floor (one v_min / v_min3 per state, one compare per two states); H is tested where a block starts (max|w| < 2^14:
// a step multiplies max|w| by less than 25.2 and adds at most 2^7, so the block's other three states stay below 2^29).
// The form above tests every SECOND state and therefore needs a test relative to the state's size (part ratio 2^-40 and a
// 60-binade window: six vector instructions per two states instead of three).  Exact zero parts fail the floor (pixels on
// the axes go to the exponent-tracking loop, as before).
// Two forms, selected at build time (FS_FL_EVERY).  1 (the default) = every state against the floor 2^-56, as derived above:
// each state of a run is certified.  0 = every SECOND state (a trip's second step) against the higher floor 2^-44 -- one
// v_min and one compare per two states instead of two and one: 49.0 instead of 51.5 ms on C3, the same frames on every
// test -- but its argument has a gap and it is NOT the default: the untested first state `a` of a trip can differ from the
// reference's in a part that is itself below 2^-60 (by less than 2^-86), and although that difference is 2^25 ulps below
// anything that matters in a second state b whose parts are at least 2^-44, it can still flip a rounding of b when one of
// b's intermediate sums happens to land within that distance of a rounding boundary (probability of the order of 2^-15
// per such trip).  tools/floor_check.py (FS_VERIFY_FLOOR build) counts the trips whose first state has a part below 2^-56
// while the second passes: about 1 in 10^4 wave-trips on C3's view -- rare, not absent.
#ifndef FS_FL_EVERY
#if defined(FS_VERIFY_FLOOR)
#define FS_FL_EVERY 0
#else
#define FS_FL_EVERY 1
#endif
#endif
#if defined(FS_VERIFY_FLOOR) && FS_FL_EVERY
#error "FS_VERIFY_FLOOR measures the every-second-state form"
#endif
#ifndef FS_FL_SHIFT
#define FS_FL_SHIFT 24 /* measured on C3 (every-state form): 20 / 24 / 28 -> 51.5 / 51.7 / 51.5 ms; second-state form 10 .. 28 in DESIGN.md */
#endif
constexpr int kScaleShift = FS_FL_SHIFT;
#ifndef FS_FL_FLOOR_EXP
#if FS_FL_EVERY
#define FS_FL_FLOOR_EXP 56
#else
#define FS_FL_FLOOR_EXP 44
#endif
#endif
static_assert(FS_FL_EVERY ? FS_FL_FLOOR_EXP <= 56 : FS_FL_FLOOR_EXP <= 48, "the floor's margins (see above)");
#define FS_FL_FLOOR __builtin_amdgcn_ldexpf(1.0f, -FS_FL_FLOOR_EXP)
constexpr int kFloorBits = (127 - FS_FL_FLOOR_EXP) << 23;
#define FS_FL_HIGH 0x1p14f   /* max|w| where a 4-step block starts */
#define FS_FL_HIGH_TRIP 0x1p24f /* the per-lane paths test H once per two-step trip: 25.2 * 2^24 + 2^7 < 2^29 */
#if defined(FS_VERIFY_FLOOR)
// VERIFICATION BUILD (tools/floor_check.py): the every-second-state form, plus a record of every trip whose FIRST state has a
// part below 2^-56 (the every-state floor) -- the only trips on which the two forms can differ at all.  The record is the
// sticky lane mask %[xa]; the caller counts the loop invocations that leave it non-zero.
#define FS_FL_N1(A, B) "v_min_f32_e64 v61, |" A "|, |" B "|\n\tv_cmp_gt_f32_e32 vcc, %[flr56], v61\n\ts_or_b64 %[xa], %[xa], vcc\n\t"
#define FS_FL_N2(A, B) "v_min_f32_e64 v61, |" A "|, |" B "|\n\t"
#elif FS_FL_EVERY
#define FS_FL_N1(A, B) "v_min_f32_e64 v61, |" A "|, |" B "|\n\t"
#define FS_FL_N2(A, B) "v_min3_f32 v61, |" A "|, |" B "|, v61\n\t"
#else
#define FS_FL_N1(A, B) ""
#define FS_FL_N2(A, B) "v_min_f32_e64 v61, |" A "|, |" B "|\n\t"
#endif
#define FS_FL_C "v_cmp_gt_f32_e32 vcc, %[flr], v61\n\t"   /* floor > the smallest part tested */
#define FS_FL_H "v_cmp_lt_f32_e32 vcc, 0x46800000, v60\n\t"   /* 2^14 < max|w| at a block's first state */
#define FS_STEP_FLOOR(NW_, V)                                                                                       \
    V |= __builtin_amdgcn_ballot_w64(!(fs_min_abs(NW_.x, NW_.y) >= FS_FL_FLOOR));
#if defined(FS_VERIFY_FLOOR)
#define FS_STEP_FLOOR_FIRST(NW_, V)                                                                                 \
    if (kStats && __builtin_amdgcn_ballot_w64(!(fs_min_abs(NW_.x, NW_.y) >= 0x1p-56f)) != 0ull) \
        c_blk_violation++;
#elif FS_FL_EVERY
#define FS_STEP_FLOOR_FIRST(NW_, V) FS_STEP_FLOOR(NW_, V)
#else
#define FS_STEP_FLOOR_FIRST(NW_, V)
#endif
// The untested body, floor form (round 3's form of this statement tested every second state against a ratio and a window:
// six vector instructions per two states).  Registers, rotation of the four state pairs and exits as described above; a
// trip's two states (first step, second step) are tested together while the next trip's packed arithmetic is in flight,
// the verdict arrives before that trip's second step overwrites the failed trip's start state.  On entry the pending
// "previous trip" is (v[54:55], v[48:49]): the caller passes the entering state in both.
#define FS_FAST_LOOP_FL(PF)                                                                                           \
    asm volatile(                                                                                                   \
        ".Lfl_loop_%=:\n\t" /* eight steps left?  the first block's tests: max(max|w|, max|dc|) against .w (s67), H */ \
        "v_max_i32_e32 v62, v60, %[imdc]\n\t"                                                                       \
        "s_cmp_gt_u32 %[off], %[lim8]\n\t"                                                                          \
        "v_add_u32_e32 v62, v62, %[esh]\n\t"                                                                        \
        "s_cbranch_scc1 .Lfl_out_%=\n\t"                                                                            \
        "v_cmp_lt_i32_e64 %[m], s67, v62\n\t" FS_FL_H                                                               \
        "s_or_b64 %[m], %[m], vcc\n\t"                                                                              \
        "s_cbranch_scc1 .Lfl_out_%=\n\t" /* steps 1, 2 + the pending tests (previous body's last trip) */           \
        FS_PK_F(FS_R0, "s[64:65]") "s_mov_b32 %[eb], s62\n\t"                                                       \
        "s_load_dwordx16 s[36:51], s[68:69], %[off]\n\t"                                                            \
        "s_load_dwordx16 s[52:67], s[68:69], %[off] offset:0x40\n\t"                                                \
        FS_PK_MA(FS_R0) FS_FL_N1("v54", "v55") FS_PK_MB(FS_R0) FS_FL_N2("v48", "v49") FS_PK_P FS_FL_C FS_PK_A(FS_R1) \
        "s_waitcnt lgkmcnt(0)\n\t" PF                                                                               \
        FS_PK_F(FS_R1, "s[36:37]") FS_PK_MA(FS_R1) FS_PK_MB(FS_R1) FS_PK_P                                          \
        "s_cbranch_vccnz .Lfl_fp_%=\n\t" FS_PK_A(FS_R2) /* steps 3, 4 + the tests of trip 1 (v[50:51], v[52:53]) */ \
        FS_PK_F(FS_R2, "s[40:41]") FS_FL_N1("v50", "v51") FS_PK_MA(FS_R2) FS_PK_MB(FS_R2) FS_FL_N2("v52", "v53")    \
        FS_PK_P FS_FL_C FS_PK_A(FS_R3)                                                                              \
        FS_PK_F(FS_R3, "s[44:45]") FS_PK_MA(FS_R3) FS_PK_MB(FS_R3) FS_PK_P                                          \
        "s_cbranch_vccnz .Lfl_f1_%=\n\t" FS_PK_A(FS_R0) /* steps 5, 6 + the tests of trip 2 (v[54:55], v[48:49]) */ \
        FS_PK_F(FS_R0, "s[48:49]") FS_FL_N1("v54", "v55") FS_PK_MA(FS_R0) FS_PK_MB(FS_R0) FS_FL_N2("v48", "v49")    \
        FS_PK_P FS_FL_C FS_PK_A(FS_R1) FS_T_X("v48", "v49")                                                         \
        FS_PK_F(FS_R1, "s[52:53]") FS_PK_MA(FS_R1) FS_PK_MB(FS_R1) FS_PK_P                                          \
        "s_cbranch_vccnz .Lfl_f2_%=\n\t" /* the second block's tests: max(max|w4|, max|dc|) against entry 3's .w, H */ \
        "v_max_i32_e32 v62, v60, %[imdc]\n\t" FS_PK_A(FS_R2) "v_add_u32_e32 v62, v62, %[esh]\n\t"                   \
        /* steps 7, 8 + the tests of trip 3 (v[50:51], v[52:53]) */                                                 \
        FS_PK_F(FS_R2, "s[56:57]") "v_cmp_lt_i32_e64 %[m], s51, v62\n\t" FS_FL_H FS_PK_MA(FS_R2)                    \
        "s_or_b64 %[m], %[m], vcc\n\t" FS_PK_MB(FS_R2) FS_PK_P "s_cbranch_scc1 .Lfl_blk_%=\n\t" FS_PK_A(FS_R3)      \
        FS_FL_N1("v50", "v51")                                                                                      \
        FS_PK_F(FS_R3, "s[60:61]") FS_FL_N2("v52", "v53") FS_PK_MA(FS_R3) FS_PK_MB(FS_R3) FS_FL_C FS_PK_P           \
        "s_cbranch_vccnz .Lfl_f3_%=\n\t" FS_PK_A(FS_R0)                                                             \
        "s_add_u32 %[off], %[off], 0x80\n\t" /* max|w8| for the next block test; its floor test rides in the next body */ \
        FS_T_X("v48", "v49") "s_branch .Lfl_loop_%=\n"                                                              \
        ".Lfl_out_%=:\n\t" /* the block here needs its bound tests, or fewer than 8 steps are left: the pending tests */ \
        "s_mov_b32 %[eb], s62\n\t" FS_FL_N1("v54", "v55") FS_FL_N2("v48", "v49") FS_FL_C                            \
        "s_cbranch_vccnz .Lfl_fp_%=\n\t"                                                                            \
        "s_mov_b32 %[st], 0\n\t"                                                                                    \
        "s_branch .Lfl_end_%=\n"                                                                                    \
        ".Lfl_blk_%=:\n\t" /* the same after the first block (no verdict is pending there) */                       \
        "s_mov_b32 %[st], 0\n\t"                                                                                    \
        "s_mov_b64 s[64:65], s[48:49]\n\t"                                                                          \
        "s_mov_b32 s67, s51\n\t"                                                                                    \
        "s_add_u32 %[off], %[off], 0x40\n\t"                                                                        \
        "s_branch .Lfl_end_%=\n"                                                                                    \
        ".Lfl_fp_%=:\n\t" /* the previous body's last trip: start state v[52:53], first step v[54:55] */            \
        "s_mov_b32 %[st], 2\n\t"                                                                                    \
        "s_sub_u32 %[off], %[off], 0x20\n\t"                                                                        \
        "s_branch .Lfl_end_%=\n"                                                                                    \
        ".Lfl_f1_%=:\n\t"                                                                                           \
        "s_mov_b32 %[st], 1\n\t"                                                                                    \
        "s_mov_b32 %[eb], s38\n\t"                                                                                  \
        "s_branch .Lfl_end_%=\n"                                                                                    \
        ".Lfl_f2_%=:\n\t"                                                                                           \
        "s_mov_b32 %[st], 2\n\t"                                                                                    \
        "s_mov_b32 %[eb], s46\n\t"                                                                                  \
        "s_add_u32 %[off], %[off], 0x20\n\t"                                                                        \
        "s_branch .Lfl_end_%=\n"                                                                                    \
        ".Lfl_f3_%=:\n\t"                                                                                           \
        "s_mov_b32 %[st], 1\n\t"                                                                                    \
        "s_mov_b32 %[eb], s54\n\t"                                                                                  \
        "s_add_u32 %[off], %[off], 0x40\n"                                                                          \
        ".Lfl_end_%=:\n\t"                                                                                          \
        "s_waitcnt lgkmcnt(0)" /* (a failed pending trip leaves after the loads: nothing stays in flight) */        \
        : "+{v[48:49]}"(wv), "={v[50:51]}"(r1), "={v[52:53]}"(r2), "+{v[54:55]}"(r3), "={v[56:57]}"(ts_),           \
          "={v[58:59]}"(ta_), "+{v60}"(mxS), "={v61}"(tn_), "={v62}"(tl_), [m] "=&s"(msk_), [st] "=&s"(st),         \
          [eb] "=&s"(ebo), "+{s67}"(pwi), "+{s[64:65]}"(zS), [off] "+s"(off), [pf] "=&s"(pf_), [pg] "=&s"(pg_),     \
          [ph] "=&s"(ph_), [xa] "+s"(xacc_)                                                                         \
        : [se] "v"(sE2), [dc] "v"(dcs), [esh] "v"(Esh), [imdc] "v"(imdc), [lim8] "s"(lim8), "{s[68:69]}"(zpb),      \
          [flr] "s"(kFloorBits), [flr56] "s"((127 - 56) << 23)                                                      \
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",  \
          "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s66", "vcc",  \
          "scc")

// The untested body with the floor verdict DEFERRED (every-state form only): one v_min3 per state accumulates the smallest
// part of every state the invocation passes through (v61, +inf on entry: the entering state has been certified by whoever
// made it), no compare and no branch per trip; the verdict is taken once, on the way out -- status 3 = some state fell below
// the floor: the caller discards the whole run attempt (nothing has been committed) and repeats it with FS_FAST_LOOP_FL,
// whose per-trip verdicts stop at the failing trip.  A state below the floor is rare (about 1 in 10^4 wave-trips on C3), the
// repeat costs next to nothing, and each state of an accepted invocation has been tested exactly as in the per-trip form --
// every-state rigour at the price of the every-second-state form.  H and the block bounds are tested where a block starts,
// as in FS_FAST_LOOP_FL (they guard the steps that follow, so they cannot be deferred).
#define FS_FL_ACC(A, B) "v_min3_f32 v61, |" A "|, |" B "|, v61\n\t"
// The block test's pieces are macro parameters (BMAX / BADD / HCMP / HOR): a second and third copy of the loop without the dc half
// (max|dc| 2^E within the smallest block bound of the whole orbit) and without H (E >= -26 in every lane) were written and would
// save about 1 ms on C3, but more than one copy of this statement per kernel makes the backend fail ("illegal VGPR to SGPR copy":
// the statement's scalar in/out operands meet in phis it treats as divergent) -- one copy, the general one, is instantiated.
#define FS_BT_DC_MAX "v_max_i32_e32 v62, v60, %[imdc]\n\t"
#define FS_BT_DC_ADD "v_add_u32_e32 v62, v62, %[esh]\n\t"
#define FS_BT_NODC_MAX ""
#define FS_BT_NODC_ADD "v_add_u32_e32 v62, v60, %[esh]\n\t"
#define FS_BT_H_CMP FS_FL_H
#define FS_BT_H_OR "s_or_b64 %[m], %[m], vcc\n\t"
#define FS_BT_NOH_CMP ""
#define FS_BT_NOH_OR "s_cmp_lg_u64 %[m], 0\n\t"
#define FS_FAST_LOOP_FD(PF, BMAX, BADD, HCMP, HOR)                                                                                          \
    asm volatile(                                                                                                   \
        "v_mov_b32_e32 v61, 0x7f800000\n"                                                                           \
        ".Lfd_loop_%=:\n\t" /* eight steps left?  the first block's tests: max(max|w|, max|dc|) against .w (s67), H */ \
        BMAX "s_cmp_gt_u32 %[off], %[lim8]\n\t" BADD                                                                \
        "s_cbranch_scc1 .Lfd_out_%=\n\t"                                                                            \
        "v_cmp_lt_i32_e64 %[m], s67, v62\n\t" HCMP HOR                                                              \
        "s_cbranch_scc1 .Lfd_out_%=\n\t" /* steps 1 .. 4 */                                                         \
        FS_PK_F(FS_R0, "s[64:65]")                                                                                  \
        "s_load_dwordx16 s[36:51], s[68:69], %[off]\n\t"                                                            \
        "s_load_dwordx16 s[52:67], s[68:69], %[off] offset:0x40\n\t"                                                \
        FS_PK_MA(FS_R0) FS_PK_MB(FS_R0) FS_PK_P FS_PK_A(FS_R1)                                                      \
        "s_waitcnt lgkmcnt(0)\n\t" PF                                                                               \
        FS_PK_F(FS_R1, "s[36:37]") FS_FL_ACC("v50", "v51") FS_PK_MA(FS_R1) FS_PK_MB(FS_R1) FS_PK_P FS_PK_A(FS_R2)   \
        FS_PK_F(FS_R2, "s[40:41]") FS_FL_ACC("v52", "v53") FS_PK_MA(FS_R2) FS_PK_MB(FS_R2) FS_PK_P FS_PK_A(FS_R3)   \
        FS_PK_F(FS_R3, "s[44:45]") FS_FL_ACC("v54", "v55") FS_PK_MA(FS_R3) FS_PK_MB(FS_R3) FS_PK_P FS_PK_A(FS_R0)   \
        /* step 5 + w4's floor part and max; the second block's tests in step 6, before anything of block 2 is counted */ \
        FS_PK_F(FS_R0, "s[48:49]") FS_FL_ACC("v48", "v49") FS_PK_MA(FS_R0) FS_T_X("v48", "v49") FS_PK_MB(FS_R0)     \
        FS_PK_P BMAX FS_PK_A(FS_R1) BADD                                                                            \
        FS_PK_F(FS_R1, "s[52:53]") "v_cmp_lt_i32_e64 %[m], s51, v62\n\t" HCMP FS_PK_MA(FS_R1)                       \
        HOR FS_PK_MB(FS_R1) FS_PK_P "s_cbranch_scc1 .Lfd_blk_%=\n\t" FS_PK_A(FS_R2)                                 \
        FS_PK_F(FS_R2, "s[56:57]") FS_FL_ACC("v50", "v51") FS_PK_MA(FS_R2) FS_FL_ACC("v52", "v53") FS_PK_MB(FS_R2)  \
        FS_PK_P FS_PK_A(FS_R3)                                                                                      \
        FS_PK_F(FS_R3, "s[60:61]") FS_FL_ACC("v54", "v55") FS_PK_MA(FS_R3) FS_PK_MB(FS_R3) FS_PK_P FS_PK_A(FS_R0)   \
        "s_add_u32 %[off], %[off], 0x80\n\t"                                                                        \
        FS_T_X("v48", "v49") FS_FL_ACC("v48", "v49") "s_branch .Lfd_loop_%=\n"                                      \
        ".Lfd_blk_%=:\n\t" /* the second block needs its bound tests (or H): the state is w4 in v[48:49] */         \
        "s_mov_b64 s[64:65], s[48:49]\n\t"                                                                          \
        "s_mov_b32 s67, s51\n\t"                                                                                    \
        "s_add_u32 %[off], %[off], 0x40\n"                                                                          \
        ".Lfd_out_%=:\n\t" /* the verdict over every state of this invocation */                                    \
        "s_mov_b32 %[st], 0\n\t" FS_FL_C                                                                            \
        "s_cbranch_vccz .Lfd_end_%=\n\t"                                                                            \
        "s_mov_b32 %[st], 3\n"                                                                                      \
        ".Lfd_end_%=:\n\t"                                                                                          \
        "s_waitcnt lgkmcnt(0)"                                                                                      \
        : "+{v[48:49]}"(wv), "={v[50:51]}"(r1), "={v[52:53]}"(r2), "={v[54:55]}"(r3), "={v[56:57]}"(ts_),           \
          "={v[58:59]}"(ta_), "+{v60}"(mxS), "={v61}"(tn_), "={v62}"(tl_), [m] "=&s"(msk_), [st] "=&s"(st),         \
          "+{s67}"(pwi), "+{s[64:65]}"(zS), [off] "+s"(off), [pf] "=&s"(pf_), [pg] "=&s"(pg_), [ph] "=&s"(ph_)      \
        : [se] "v"(sE2), [dc] "v"(dcs), [esh] "v"(Esh), [imdc] "v"(imdc), [lim8] "s"(lim8), "{s[68:69]}"(zpb),      \
          [flr] "s"(kFloorBits)                                                                                     \
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",  \
          "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s66", "vcc",  \
          "scc")

// The same loop with the block test on WAVE-UNIFORM thresholds (round 5): the kernel issues one vector instruction per SIMD
// every four cycles and nothing else, and the block test was five of them per four steps (max|w|, max with max|dc|, + the lane's
// scale, the compare, H's compare).  Both sides of it that are not the state are made scalar, each in the safe direction:
//   max|dc| 2^E is the pixel's true max|dc|, a constant: its largest value over the wave (`sdc`, made once per tile) is compared
//     with the block bound on the scalar unit;
//   bits(max|w|) + Esh <= bound holds in every lane when bits(max|w|) <= bound - (the LARGEST Esh of the running lanes: `eshm`,
//     made once per run by a few votes);  H is bits(max|w|) <= bits(2^14);
// so a block's test is  bits(max|w|) <= T,  T = min(bound - eshm, bits(2^14)), or -1 ("never") when sdc > bound -- which the
// "never" bound, the most negative integer, always is: five scalar instructions, then max|w| and ONE compare on the vector unit.
// (For a usable bound, >= 0, the difference can only overflow upwards, and an overflow means T = H; what the first three
// instructions make of the "never" bound is overwritten.  The first form of this macro replaced that bound by -2^30 and
// subtracted: positive again under a scale shift below -2^30, i.e. for |dz| < 2^-152 -- tools/block_bound_check.py counted 24 150
// such blocks among 3.6e9 on the deep views 11, 14 and 19, none on View 5.)  A wave whose lanes' scales are k binades
// apart tests its lower lanes against a bound 2^k tighter than theirs: such a block takes the tested path, nothing else changes.
#define FS_BT_T(BW)                                                                                                 \
    "s_sub_i32 %[t], " BW ", %[eshm]\n\t"                                                                           \
    "s_cselect_b32 %[t], 0x46800000, %[t]\n\t"                                                                      \
    "s_min_i32 %[t], %[t], 0x46800000\n\t"                                                                          \
    "s_cmp_gt_i32 %[sdc], " BW "\n\t"                                                                               \
    "s_cselect_b32 %[t], -1, %[t]\n\t"
#define FS_BT_V "v_cmp_lt_i32_e32 vcc, %[t], v60\n\t"
#define FS_FAST_LOOP_FDU(PF)                                                                                        \
    asm volatile(                                                                                                   \
        "v_mov_b32_e32 v61, 0x7f800000\n\t" FS_BT_T("s67") FS_BT_V                                                  \
        ".Lfu_loop_%=:\n\t" /* eight steps left?  the first block's verdict (taken where max|w| was made) */         \
        "s_cmp_gt_u32 %[off], %[lim8]\n\t"                                                                          \
        "s_cbranch_scc1 .Lfu_out_%=\n\t"                                                                            \
        "s_cbranch_vccnz .Lfu_out_%=\n\t" /* steps 1 .. 4 */                                                        \
        FS_PK_F(FS_R0, "s[64:65]")                                                                                  \
        "s_load_dwordx16 s[36:51], s[68:69], %[off]\n\t"                                                            \
        "s_load_dwordx16 s[52:67], s[68:69], %[off] offset:0x40\n\t"                                                \
        FS_PK_MA(FS_R0) FS_PK_MB(FS_R0) FS_PK_P FS_PK_A(FS_R1)                                                      \
        "s_waitcnt lgkmcnt(0)\n\t" PF FS_BT_T("s51")                                                                \
        FS_PK_F(FS_R1, "s[36:37]") FS_FL_ACC("v50", "v51") FS_PK_MA(FS_R1) FS_PK_MB(FS_R1) FS_PK_P FS_PK_A(FS_R2)   \
        FS_PK_F(FS_R2, "s[40:41]") FS_FL_ACC("v52", "v53") FS_PK_MA(FS_R2) FS_PK_MB(FS_R2) FS_PK_P FS_PK_A(FS_R3)   \
        FS_PK_F(FS_R3, "s[44:45]") FS_FL_ACC("v54", "v55") FS_PK_MA(FS_R3) FS_PK_MB(FS_R3) FS_PK_P FS_PK_A(FS_R0)   \
        /* step 5 + w4's floor part and max; the second block's verdict in step 6, before anything of block 2 is counted */ \
        FS_PK_F(FS_R0, "s[48:49]") FS_FL_ACC("v48", "v49") FS_PK_MA(FS_R0) FS_T_X("v48", "v49") FS_PK_MB(FS_R0)     \
        FS_PK_P FS_PK_A(FS_R1)                                                                                      \
        FS_PK_F(FS_R1, "s[52:53]") FS_BT_V FS_PK_MA(FS_R1) FS_PK_MB(FS_R1) FS_PK_P                                  \
        "s_cbranch_vccnz .Lfu_blk_%=\n\t" FS_PK_A(FS_R2)                                                            \
        FS_PK_F(FS_R2, "s[56:57]") FS_FL_ACC("v50", "v51") FS_PK_MA(FS_R2) FS_FL_ACC("v52", "v53") FS_PK_MB(FS_R2)  \
        FS_PK_P FS_PK_A(FS_R3)                                                                                      \
        FS_PK_F(FS_R3, "s[60:61]") FS_FL_ACC("v54", "v55") FS_PK_MA(FS_R3) FS_PK_MB(FS_R3) FS_PK_P FS_PK_A(FS_R0)   \
        "s_add_u32 %[off], %[off], 0x80\n\t" FS_BT_T("s67")                                                         \
        FS_T_X("v48", "v49") FS_FL_ACC("v48", "v49") FS_BT_V "s_branch .Lfu_loop_%=\n"                              \
        ".Lfu_blk_%=:\n\t" /* the second block needs its bound tests (or H): the state is w4 in v[48:49] */         \
        "s_mov_b64 s[64:65], s[48:49]\n\t"                                                                          \
        "s_mov_b32 s67, s51\n\t"                                                                                    \
        "s_add_u32 %[off], %[off], 0x40\n"                                                                          \
        ".Lfu_out_%=:\n\t" /* the verdict over every state of this invocation */                                    \
        "s_mov_b32 %[st], 0\n\t" FS_FL_C                                                                            \
        "s_cbranch_vccz .Lfu_end_%=\n\t"                                                                            \
        "s_mov_b32 %[st], 3\n"                                                                                      \
        ".Lfu_end_%=:\n\t"                                                                                          \
        "s_waitcnt lgkmcnt(0)"                                                                                      \
        : "+{v[48:49]}"(wv), "={v[50:51]}"(r1), "={v[52:53]}"(r2), "={v[54:55]}"(r3), "={v[56:57]}"(ts_),           \
          "={v[58:59]}"(ta_), "+{v60}"(mxS), "={v61}"(tn_), [t] "=&s"(bt_t_), [st] "=&s"(st),                       \
          "+{s67}"(pwi), "+{s[64:65]}"(zS), [off] "+s"(off), [pf] "=&s"(pf_), [pg] "=&s"(pg_), [ph] "=&s"(ph_)      \
        : [se] "v"(sE2), [dc] "v"(dcs), [eshm] "s"(Esh_cap), [sdc] "s"(sdc_bits), [lim8] "s"(lim8),                 \
          "{s[68:69]}"(zpb), [flr] "s"(kFloorBits)                                                                  \
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",  \
          "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s66", "vcc",  \
          "scc")

// The untested body with the deferred verdict, SIXTEEN steps per body (round 4).  A wave that is alone on its SIMD -- the
// never-escaping pixels that decide C2's frame time, the last waves of a rank of an N-GPU split -- pays one L2 round trip per
// body: scalar loads return out of order, so the entries of a body can only be waited for all together, and the loads that
// warm the scalar cache for the next body are waited for with them.  Twice the steps per round trip: the body reads its
// entries in the COMPACT form -- 2Z alone, 8 bytes per entry (zs2: two s_load_dwordx16 for sixteen entries), and ONE
// 16-byte record (zqb) with the block bounds of its entries 3, 7, 11 and 15 -- instead of sixteen bytes per entry.
// Registers: entries E0 .. E15 in s[36:67] (E15 = s[66:67] is the entry the state is at when the body ends: the next
// body's first step reads it BEFORE the loads overwrite it), the four block bounds in s[72:75] (s75 = the bound of the
// state's entry at the loop's top), bases s[68:69] (zs2) and s[70:71] (zqb), `off` = 16 bytes per step as everywhere.
// State pairs, temporaries, the floor accumulator and the statuses as in FS_FAST_LOOP_FD: 0 = stopped in front of a block
// that needs its tests / fewer than 16 steps left, 3 = a state below the floor (the caller repeats the run attempt with
// FS_FAST_LOOP_FL).  Blocks 2 .. 4 are tested in the second step of the block, before anything of the block is counted.
#define FS_FD16_PAIR(EA, EB_, BW, LBL)                                                                              \
    FS_PK_F(FS_R0, EA) FS_FL_ACC("v48", "v49") FS_PK_MA(FS_R0) FS_T_X("v48", "v49") FS_PK_MB(FS_R0)                 \
    FS_PK_P FS_BT_DC_MAX FS_PK_A(FS_R1) FS_BT_DC_ADD                                                                \
    FS_PK_F(FS_R1, EB_) "v_cmp_lt_i32_e64 %[m], " BW ", v62\n\t" FS_BT_H_CMP FS_PK_MA(FS_R1)                         \
    FS_BT_H_OR FS_PK_MB(FS_R1) FS_PK_P "s_cbranch_scc1 " LBL "\n\t" FS_PK_A(FS_R2)
#define FS_FD16_TAIL(EC, ED)                                                                                        \
    FS_PK_F(FS_R2, EC) FS_FL_ACC("v50", "v51") FS_PK_MA(FS_R2) FS_FL_ACC("v52", "v53") FS_PK_MB(FS_R2)              \
    FS_PK_P FS_PK_A(FS_R3)                                                                                          \
    FS_PK_F(FS_R3, ED) FS_FL_ACC("v54", "v55") FS_PK_MA(FS_R3) FS_PK_MB(FS_R3) FS_PK_P FS_PK_A(FS_R0)
#define FS_PF16_NONE ""
#define FS_PF16_NEXT_BODY                                                                                           \
    "s_load_dword %[pf], s[68:69], %[oc] offset:0x80\n\t"                                                           \
    "s_load_dword %[pg], s[68:69], %[oc] offset:0xc0\n\t"                                                           \
    "s_load_dword %[ph], s[68:69], %[oc] offset:0xfc\n\t"                                                           \
    "s_load_dword %[pi], s[70:71], %[off] offset:0x100\n\t"                                                         \
    "s_load_dword %[pj], s[70:71], %[off] offset:0x10c\n\t"
#define FS_FAST_LOOP_FD16(PF)                                                                                       \
    asm volatile(                                                                                                   \
        "v_mov_b32_e32 v61, 0x7f800000\n"                                                                           \
        ".Lfe_loop_%=:\n\t" /* sixteen steps left?  the first block's tests: max(max|w|, max|dc|) against s75, H */  \
        FS_BT_DC_MAX "s_cmp_gt_u32 %[off], %[lim16]\n\t" FS_BT_DC_ADD                                               \
        "s_cbranch_scc1 .Lfe_out_%=\n\t"                                                                            \
        "v_cmp_lt_i32_e64 %[m], s75, v62\n\t" FS_BT_H_CMP FS_BT_H_OR                                                \
        "s_cbranch_scc1 .Lfe_out_%=\n\t" /* steps 1 .. 4 */                                                         \
        FS_PK_F(FS_R0, "s[66:67]")                                                                                  \
        "s_lshr_b32 %[oc], %[off], 1\n\t"                                                                           \
        "s_load_dwordx16 s[36:51], s[68:69], %[oc]\n\t"                                                             \
        "s_load_dwordx16 s[52:67], s[68:69], %[oc] offset:0x40\n\t"                                                 \
        "s_load_dwordx4 s[72:75], s[70:71], %[off]\n\t"                                                             \
        FS_PK_MA(FS_R0) FS_PK_MB(FS_R0) FS_PK_P FS_PK_A(FS_R1)                                                      \
        "s_waitcnt lgkmcnt(0)\n\t" PF                                                                               \
        FS_PK_F(FS_R1, "s[36:37]") FS_FL_ACC("v50", "v51") FS_PK_MA(FS_R1) FS_PK_MB(FS_R1) FS_PK_P FS_PK_A(FS_R2)   \
        FS_PK_F(FS_R2, "s[38:39]") FS_FL_ACC("v52", "v53") FS_PK_MA(FS_R2) FS_PK_MB(FS_R2) FS_PK_P FS_PK_A(FS_R3)   \
        FS_PK_F(FS_R3, "s[40:41]") FS_FL_ACC("v54", "v55") FS_PK_MA(FS_R3) FS_PK_MB(FS_R3) FS_PK_P FS_PK_A(FS_R0)   \
        /* steps 5 .. 8: w4's floor part and max in step 5, the second block's tests in step 6 */                   \
        FS_FD16_PAIR("s[42:43]", "s[44:45]", "s72", ".Lfe_b1_%=") FS_FD16_TAIL("s[46:47]", "s[48:49]")              \
        /* steps 9 .. 12 */                                                                                         \
        FS_FD16_PAIR("s[50:51]", "s[52:53]", "s73", ".Lfe_b2_%=") FS_FD16_TAIL("s[54:55]", "s[56:57]")              \
        /* steps 13 .. 16 */                                                                                        \
        FS_FD16_PAIR("s[58:59]", "s[60:61]", "s74", ".Lfe_b3_%=") FS_FD16_TAIL("s[62:63]", "s[64:65]")              \
        "s_add_u32 %[off], %[off], 0x100\n\t"                                                                       \
        FS_T_X("v48", "v49") FS_FL_ACC("v48", "v49") "s_branch .Lfe_loop_%=\n"                                      \
        ".Lfe_b1_%=:\n\t" /* block 2 needs its bound tests (or H): the state is w4 in v[48:49], at entry 3 */       \
        "s_mov_b64 s[66:67], s[42:43]\n\t"                                                                          \
        "s_mov_b32 s75, s72\n\t"                                                                                    \
        "s_add_u32 %[off], %[off], 0x40\n\t"                                                                        \
        "s_branch .Lfe_out_%=\n"                                                                                    \
        ".Lfe_b2_%=:\n\t" /* block 3: w8, entry 7 */                                                                \
        "s_mov_b64 s[66:67], s[50:51]\n\t"                                                                          \
        "s_mov_b32 s75, s73\n\t"                                                                                    \
        "s_add_u32 %[off], %[off], 0x80\n\t"                                                                        \
        "s_branch .Lfe_out_%=\n"                                                                                    \
        ".Lfe_b3_%=:\n\t" /* block 4: w12, entry 11 */                                                              \
        "s_mov_b64 s[66:67], s[58:59]\n\t"                                                                          \
        "s_mov_b32 s75, s74\n\t"                                                                                    \
        "s_add_u32 %[off], %[off], 0xc0\n"                                                                          \
        ".Lfe_out_%=:\n\t" /* the verdict over every state of this invocation */                                    \
        "s_mov_b32 %[st], 0\n\t" FS_FL_C                                                                            \
        "s_cbranch_vccz .Lfe_end_%=\n\t"                                                                            \
        "s_mov_b32 %[st], 3\n"                                                                                      \
        ".Lfe_end_%=:\n\t"                                                                                          \
        "s_waitcnt lgkmcnt(0)"                                                                                      \
        : "+{v[48:49]}"(wv), "={v[50:51]}"(r1), "={v[52:53]}"(r2), "={v[54:55]}"(r3), "={v[56:57]}"(ts_),           \
          "={v[58:59]}"(ta_), "+{v60}"(mxS), "={v61}"(tn_), "={v62}"(tl_), [m] "=&s"(msk_), [st] "=&s"(st),         \
          "+{s75}"(pwi), "+{s[66:67]}"(zS), [off] "+s"(off), [oc] "=&s"(oc_), [pf] "=&s"(pf_), [pg] "=&s"(pg_),     \
          [ph] "=&s"(ph_), [pi] "=&s"(pi_), [pj] "=&s"(pj_)                                                         \
        : [se] "v"(sE2), [dc] "v"(dcs), [esh] "v"(Esh), [imdc] "v"(imdc), [lim16] "s"(lim16), "{s[68:69]}"(zpb2),   \
          "{s[70:71]}"(zqbp), [flr] "s"(kFloorBits)                                                                 \
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",  \
          "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",  \
          "s72", "s73", "s74", "vcc", "scc")

// The sixteen-step body as a TWO-STAGE PIPELINE (round 5).  Scalar loads return out of order behind one counter, so a wait is a
// wait for everything in flight -- but nothing says the wait has to follow the request: the body's entries live in two halves
// (E0 .. E7 in s[36:51] with the bounds of the entries 3 and 7 in s[72:73]; E8 .. E15 in s[52:67] with those of 11 and 15 in
// s[74:75]), and each half is requested while the OTHER one is being consumed -- the upper half at step 1 (right after the
// instruction that reads E15 of the body before), the next body's lower half at step 10 (right after the instructions that read
// E7 and the bound of entry 7) -- and waited for eight steps later, just before its first use, when it has long landed: the
// only thing in flight at either wait is the half requested eight steps ago.  A wave that is alone on its SIMD no longer
// stands still for an L2 round trip per body (C2's interior pixels: 4.7 M dependent steps; FS_FAST_LOOP_FD16 with its warming
// loads measured ~30 ns per step against the ~19 ns of the step's dependent arithmetic).  Same registers as FS_FAST_LOOP_FD16,
// statuses 0 and 3 as there (4: below).  The half requested past the end of a run is never used (the companion arrays carry 32
// entries of slack).
// The deferred floor verdict is taken PER BODY: the state a body starts from is kept (v[46:47], its step count in `cko`) once
// the verdict over the body before has passed, and a state below the floor sends the statement back to that checkpoint with
// status 3 -- the caller commits the certified steps and lets the per-trip loop (FS_FAST_LOOP_FL) find the failing trip in
// the sixteen steps that follow, instead of repeating the whole run with it (a 2048-step run that ends on a floor failure,
// which is how most runs of C2's never-escaping pixels end, was executed twice).
// The tests of the blocks INSIDE a body (its second to fourth) are deferred too (round 5): a wave that is alone on its SIMD
// issues in order, and `compare -> scalar or -> branch` makes it wait for the vector pipeline to drain at every block -- measured
// on the isolated loop (tools/microbench/lone_pace.hip) 20.7 ns per step with the three branches, 17.6 without, 11.0 for the
// arithmetic alone.  Each of those blocks leaves its verdict in v63 instead (positive = violated; it only grows):
//     max(max|w|, max|dc|) + Esh - bound, saturating (the "never" bound is the most negative integer),  and  max(..) - 2^14 (H;
//     max|dc| 2^-E <= 2^7 by the start condition, so taking the maximum with it changes nothing there)
// and the body's steps run on whatever comes.  ONE verdict per body, at the top of the next one (and on every way out): the floor
// accumulator v61 and v63 together; a violation of either sends the statement back to the body's checkpoint -- status 3 (floor
// alone: the caller commits the certified steps and lets the per-trip loop find the failing trip) or 4 (a block test: the
// caller takes the block in front of it through the tested form, as it does for status 0; the entry values it needs it reads
// itself).  What ran past a violated block test is discarded with the roll-back: nothing but registers was written.
// The first block of a body is treated the same way (blocks that need their tests are 0.1 % of the steps of C2's long pixels:
// a body run in vain in front of each costs nothing next to one more drain of the pipeline per body).
#define FS_FD16D_PAIR(EA, EB_, BW, WAIT, LOADS)                                                                     \
    FS_PK_F(FS_R0, EA) FS_FL_ACC("v48", "v49") FS_PK_MA(FS_R0)                                                      \
    "v_max3_f32 v62, |v48|, |v49|, %[imdc]\n\t"                                                                     \
    FS_PK_MB(FS_R0) FS_PK_P                                                                                         \
    "v_subrev_u32_e32 v45, 0x46800000, v62\n\t"                                                                     \
    FS_PK_A(FS_R1)                                                                                                  \
    "v_add_u32_e32 v62, v62, %[esh]\n\t" WAIT                                                                       \
    FS_PK_F(FS_R1, EB_)                                                                                             \
    "v_sub_i32 v62, v62, " BW " clamp\n\t"                                                                          \
    FS_PK_MA(FS_R1)                                                                                                 \
    "v_max3_i32 v63, v63, v62, v45\n\t"                                                                             \
    FS_PK_MB(FS_R1) FS_PK_P LOADS FS_PK_A(FS_R2)
#define FS_FAST_LOOP_FD16P                                                                                          \
    asm volatile(                                                                                                   \
        "v_mov_b32_e32 v61, 0x7f800000\n\t" /* the first body's lower half; every later body finds its own requested */ \
        "v_bfrev_b32_e32 v63, 1\n\t"                                                                                \
        "s_lshr_b32 %[oc], %[off], 1\n\t"                                                                           \
        "s_load_dwordx16 s[36:51], s[68:69], %[oc]\n\t"                                                             \
        "s_load_dwordx2 s[72:73], s[70:71], %[off]\n"                                                               \
        ".Lfp_loop_%=:\n\t" /* the verdict over the body before: floor (flr > the smallest part seen) or a block test */ \
        "v_sub_u32_e32 v45, %[flr], v61\n\t"                                                                        \
        FS_BT_DC_MAX "s_cmp_gt_u32 %[off], %[lim16]\n\t"                                                            \
        "v_max_i32_e32 v45, v45, v63\n\t"                                                                           \
        FS_BT_DC_ADD                                                                                                \
        "v_cmp_lt_i32_e32 vcc, 0, v45\n\t"                                                                          \
        "s_cbranch_vccnz .Lfp_redo_%=\n\t" /* sixteen steps left? */                                                \
        "s_cbranch_scc1 .Lfp_out_%=\n\t" /* the checkpoint: every state up to here is certified */                  \
        "v_mov_b32_e32 v46, v48\n\t"                                                                                \
        "v_mov_b32_e32 v47, v49\n\t"                                                                                \
        "s_mov_b32 %[cko], %[off]\n\t"                                                                              \
        "v_mov_b32_e32 v61, 0x7f800000\n\t" /* steps 1 .. 4 + the first block's verdict (s75, H) into v63 */        \
        FS_PK_F(FS_R0, "s[66:67]")                                                                                  \
        "v_sub_i32 v62, v62, s75 clamp\n\t"                                                                         \
        "v_subrev_u32_e32 v45, 0x46800000, v60\n\t"                                                                 \
        "s_waitcnt lgkmcnt(0)\n\t" /* the lower half has landed, the upper half is requested */                     \
        "s_lshr_b32 %[oc], %[off], 1\n\t"                                                                           \
        "s_load_dwordx16 s[52:67], s[68:69], %[oc] offset:0x40\n\t"                                                 \
        "s_load_dwordx2 s[74:75], s[70:71], %[off] offset:0x8\n\t"                                                  \
        FS_PK_MA(FS_R0)                                                                                             \
        "v_max_i32_e32 v63, v62, v45\n\t"                                                                           \
        FS_PK_MB(FS_R0) FS_PK_P FS_PK_A(FS_R1)                                                                      \
        FS_PK_F(FS_R1, "s[36:37]") FS_FL_ACC("v50", "v51") FS_PK_MA(FS_R1) FS_PK_MB(FS_R1) FS_PK_P FS_PK_A(FS_R2)   \
        FS_PK_F(FS_R2, "s[38:39]") FS_FL_ACC("v52", "v53") FS_PK_MA(FS_R2) FS_PK_MB(FS_R2) FS_PK_P FS_PK_A(FS_R3)   \
        FS_PK_F(FS_R3, "s[40:41]") FS_FL_ACC("v54", "v55") FS_PK_MA(FS_R3) FS_PK_MB(FS_R3) FS_PK_P FS_PK_A(FS_R0)   \
        /* steps 5 .. 8 */                                                                                          \
        FS_FD16D_PAIR("s[42:43]", "s[44:45]", "s72", "", "") FS_FD16_TAIL("s[46:47]", "s[48:49]")                   \
        /* steps 9 .. 12: the upper half has landed (step 10 reads E8); the next body's lower half is requested */   \
        FS_FD16D_PAIR("s[50:51]", "s[52:53]", "s73", "s_waitcnt lgkmcnt(0)\n\t",                                    \
                      "s_load_dwordx16 s[36:51], s[68:69], %[oc] offset:0x80\n\t"                                   \
                      "s_load_dwordx2 s[72:73], s[70:71], %[off] offset:0x100\n\t")                                 \
        FS_FD16_TAIL("s[54:55]", "s[56:57]")                                                                        \
        /* steps 13 .. 16 */                                                                                        \
        FS_FD16D_PAIR("s[58:59]", "s[60:61]", "s74", "", "") FS_FD16_TAIL("s[62:63]", "s[64:65]")                   \
        "s_add_u32 %[off], %[off], 0x100\n\t"                                                                       \
        FS_T_X("v48", "v49") FS_FL_ACC("v48", "v49") "s_branch .Lfp_loop_%=\n"                                      \
        ".Lfp_redo_%=:\n\t" /* back to the checkpoint (state, its max, step count): status 3 (floor) or 4 (a block test) */ \
        "v_mov_b32_e32 v48, v46\n\t"                                                                                \
        "v_mov_b32_e32 v49, v47\n\t"                                                                                \
        "v_cmp_lt_i32_e32 vcc, 0, v63\n\t"                                                                          \
        "s_mov_b32 %[off], %[cko]\n\t"                                                                              \
        FS_T_X("v46", "v47")                                                                                        \
        "s_mov_b32 %[st], 3\n\t"                                                                                    \
        "s_cbranch_vccz .Lfp_end_%=\n\t"                                                                            \
        "s_mov_b32 %[st], 4\n\t"                                                                                    \
        "s_branch .Lfp_end_%=\n"                                                                                    \
        ".Lfp_out_%=:\n\t" /* in front of a block that needs its tests, or of the last steps (the verdict has passed) */ \
        "s_mov_b32 %[st], 0\n"                                                                                      \
        ".Lfp_end_%=:\n\t"                                                                                          \
        "s_waitcnt lgkmcnt(0)"                                                                                      \
        : "+{v[48:49]}"(wv), "={v[50:51]}"(r1), "={v[52:53]}"(r2), "={v[54:55]}"(r3), "={v[56:57]}"(ts_),           \
          "={v[58:59]}"(ta_), "+{v60}"(mxS), "={v61}"(tn_), "={v62}"(tl_), "={v[46:47]}"(ck_), "={v45}"(th_),       \
          "={v63}"(va_), [st] "=&s"(st), "+{s75}"(pwi), "+{s[66:67]}"(zS), [off] "+s"(off), [oc] "=&s"(oc_),        \
          [cko] "=&s"(cko_)                                                                                         \
        : [se] "v"(sE2), [dc] "v"(dcs), [esh] "v"(Esh), [imdc] "v"(imdc), [lim16] "s"(lim16), "{s[68:69]}"(zpb2),   \
          "{s[70:71]}"(zqbp), [flr] "s"(kFloorBits)                                                                 \
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",  \
          "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",  \
          "s72", "s73", "s74", "vcc", "scc")

// ------------------------------------------------------------------------------------------------
// LAv2, T = HDRFloat<float>: tuned perturbation loop.  Same prologue (AT + LA stages) and the same results, bit for
// bit, as k_lav2_hdr32; the perturbation loop (>99.9 % of the executed work at View 5) is restructured around what
// the CPU arithmetic actually does per step (measured with an instrumented oracle):
//   * 2Z+dz and Z'+dz are "orbit bigger, 0 <= exponent gap < 120" in 99.9 % of lane-steps; dz*cur+dc is "dz bigger".
//     A straight-line, branch-free step is executed speculatively under exactly those assumptions
//     (no 4-way exponent-alignment branches, no operand swaps) and committed only if EVERY running lane of the wave
//     met them (one ballot); otherwise the wave redoes that step with the generic functions of hdr_math.hpp.
//   * Reduce(z) before |z|^2 is skipped on the fast path: scaling both parts by the same power of two commutes with
//     IEEE multiply/add (no operand is near the denormal range there: the orbit part has |mantissa| >= 0.5), so
//     Reduce(|z|^2) gives the same {mantissa, exponent}.  The reduced z is only materialised on a rebase.
//   * Rebases (3.6e-4 per lane-step) and escapes leave the hot loop through cold branches.
//   * (exp, mantissa) pairs of reduced non-negative values are compared as one signed 64-bit key, which is the
//     lexicographic compareToBothPositiveReduced (HDRFloat.h:1150-1167) because IEEE bit patterns of non-negative
//     floats order like integers.
//   * The orbit entry of the *next* step is the Z' of this step: one 16-byte load per step instead of two.
namespace {

__device__ __forceinline__ float pow2_bits(int biased) { return __int_as_float(biased << 23); }

__device__ __forceinline__ long long key_of(float m, int e)
{
    return ((long long)e << 32) | (long long)(unsigned)__float_as_int(m);
}

// Reduce(norm_squared(c)) as a key; c is any complex whose larger part is a normal float.
__device__ __forceinline__ long long norm_key(float re, float im, int e)
{
    const float m = re * re + im * im; // >= +0
    const int bits = __float_as_int(m);
    const int fe = ((bits >> 23) & 0xff) - 127;
    const bool z = m == 0.0f;
    const int mm = z ? 0 : ((bits & 0x007FFFFF) | 0x3F800000);
    const int ee = (e << 1) + (z ? 0 : fe);
    return ((long long)ee << 32) | (long long)(unsigned)mm;
}

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imin3(int a, int b, int c)
{
    const int m = a < b ? a : b;
    return m < c ? m : c;
}

// 0x7F000000 - (f << 23) = the bits of 2^(127 - f), as one v_mad_i32_i24 (f < 2^8)
__device__ __forceinline__ int mad24_scale(int f)
{
    int r;
    const int k = -8388608;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(k), "s"(0x7F000000));
    return r;
}

// Same for a sum of squares already known to be a positive normal float (no zero special case).
__device__ __forceinline__ long long norm_key_nz(float m, int e)
{
    const int bits = __float_as_int(m);
    const int fe = ((bits >> 23) & 0xff) - 127;
    const int mm = (bits & 0x007FFFFF) | 0x3F800000;
    return ((long long)((e << 1) + fe) << 32) | (long long)(unsigned)mm;
}

} // namespace

// Steps per scaled run (a multiple of the 8-step body): a run's scale is fixed, and its lanes must have this many
// steps left before the orbit ends and before their iteration limit.  Measured on View 5 (C3 / C2, ms): 64: 69.9 / 385,
// 128: 68.6 / 390, 256: 68.0 / 372, 512: 68.2 / 380, 1024: 68.0 / 380, 4096: 78.9 / 561 (too few lanes qualify).
#ifndef FS_SCALED_CHUNK
#define FS_SCALED_CHUNK 256
#endif
constexpr uint32_t kScaledChunk = FS_SCALED_CHUNK;
static_assert(kScaledChunk % 8 == 0 && kScaledChunk >= 64, "a run is a whole number of 8-step bodies");

// Steps of the next scaled run: kScaledChunk when every (active) lane has that many left, else 64, else 16, else none --
// without the shorter runs the last 256 steps of every pass over the orbit (1.6 % of View 5's 16 046-entry orbit) fall to
// the exponent-tracking loop.
// Back-off of the scaled-run attempts: after an attempt that ended before its first step the wave takes this many careful
// steps more (1, 2, ... up to the cap) before it tries again; an attempt that got 8 steps or more resets it.  On C3 two
// thirds of the attempts of a wave (300 of 460) ended that way -- lanes near their escape, where dz is never small against
// the orbit -- each for the price of an entry, a trip and an exit.  Measured (C3 kernel ms / emulated 8-rank maximum): cap 0
// (no back-off) 60.3 / 10.27, 1: 57.7 / 9.81, 3: 56.5 / 9.41, 7: 55.8 / 9.36, 15: 55.5 / 9.13, 31: 56.1 / 9.28, 63: 55.4 / 9.17;
// doubling instead of counting up: no better; neither is waiting for a careful step that leaves every lane's dz 1 .. 4 binades
// below the orbit value it arrived at (57.3 .. 58.7).  Which steps run scaled changes no result.
#ifndef FS_BACKOFF_CAP
#define FS_BACKOFF_CAP 15
#endif
constexpr uint32_t kScaledBackoffCap = FS_BACKOFF_CAP;
// Steps of a hot run (k_lav2_hdr32_fast, see there) before the scale is re-centred.
#ifndef FS_HOT_RUN_STEPS
#define FS_HOT_RUN_STEPS 64 /* measured on C3 (kernel ms): 8: 50.2, 16: 49.75, 32: 49.7, 64: 49.5, 128 .. 1024: 49.5 - 49.6 */
#endif
constexpr uint32_t kHotRunSteps = FS_HOT_RUN_STEPS;
#ifndef FS_HOT_AFTER_FAIL
#define FS_HOT_AFTER_FAIL 0 /* A/B, measured neutral on C3 (43.80 against 43.85 ms): 1 = the step a run failed on goes to a hot run before the careful step */
#endif

__device__ __forceinline__ uint32_t scaled_run_length(uint32_t left)
{
    if (__builtin_amdgcn_ballot_w64(left < kScaledChunk) == 0ull)
        return kScaledChunk;
    if (__builtin_amdgcn_ballot_w64(left < 64u) == 0ull)
        return 64u;
    return __builtin_amdgcn_ballot_w64(left < 16u) == 0ull ? 16u : 0u;
}

// ... and for the perturbation-only kernel (k_perturb_scalar): one longer tier in front.  A wave that is alone on its SIMD pays
// for every instruction of a run's entry and exit (and waits out their vector loads): at 256 steps per run they were 40 % of the
// time of C2's never-escaping pixels (tools/microbench/lone_pace.hip: the loop's own pace is 11.5 ns per step, the kernel's
// 21 - 31).  A run still ends where it has to: H, a floor or bound failure, a block that needs its tests at the very end.
#ifndef FS_PO_CHUNK
#define FS_PO_CHUNK 2048
#endif
constexpr uint32_t kPoChunk = FS_PO_CHUNK;
static_assert(kPoChunk % 16 == 0 && kPoChunk >= kScaledChunk && kPoChunk <= (1u << 20), "whole 16-step bodies; offsets stay 32-bit");
// Between the tiers: when every lane has as many steps left as the first one (the lanes of a never-escaping tile walk the orbit
// together), the run takes exactly those -- a pass over View 5's 16 046-entry orbit is then 8 runs instead of 17 (seven of 2048
// and the tail in one piece instead of 256 + 256 + ... + 16 + 16).  Multiples of four: the tested form behind the statement
// advances in four-step blocks.
__device__ __forceinline__ uint32_t scaled_run_length_po(uint32_t left)
{
    if (kPoChunk > kScaledChunk) {
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)left);
        const uint32_t want = (first < kPoChunk ? first : kPoChunk) & ~3u;
        if (want > kScaledChunk && __builtin_amdgcn_ballot_w64(left < want) == 0ull)
            return want;
    }
    return scaled_run_length(left);
}

// kLds (A/B variant, north_star "LDS staging of orbit segments shared across a wavefront"): in the scaled runs whose
// lanes share their orbit position, the entries reach the wave through LDS instead of the scalar cache.  Each wave owns
// two 1-KiB LDS buffers; one global_load_lds_dwordx4 (LDS-DMA: no VGPR destination, counted by vmcnt) brings the 64
// entries of the NEXT 64 steps while the current 64 are consumed with broadcast ds_read_b128 (every lane the same
// address; counted by lgkmcnt).  Two counters, in-order returns: a true software pipeline, which the scalar loads (one
// out-of-order counter) cannot be.  What it costs: the entries live in VGPRs (8 x 4 per body) and every ds_read writes
// 1 KiB of registers.  Measured against the scalar-cache path in DESIGN.md section 5.
// kGpuStage: the LA stage-validity test in the direction of the reference's GPU twin (FS_PARITY_CPU_GPUSTAGE) instead of
// the CPU function's (FS_PARITY_CPU) -- a template parameter so that the two parity modes are two kernels (they do very
// different work per frame, and a kernel trace then lists them separately).
// An upper bound, wave-uniform, of v over the ACTIVE lanes -- the largest value itself when a vote or two find it (the values of
// a wave's lanes, scales and dc, are a few binades apart at most), at most a few binades above it otherwise: each further trip
// adds a growing slack (1, 2, 4 ... 64 binades of a binary32 bit pattern), so the loop ends after a dozen trips at the latest whatever
// the lanes hold.  Votes instead of a reduction: nothing is written under a widened EXEC.  v <= 0x7f800000.
static __device__ __forceinline__ int wave_upper_bound_i32(int v)
{
    int m = __builtin_amdgcn_readfirstlane(v);
    int slack = 0;
    for (;;) {
        const uint64_t above = __builtin_amdgcn_ballot_w64(v > m);
        if (above == 0ull)
            return m;
        const long long next = (long long)__builtin_amdgcn_readlane(v, (int)__builtin_ctzll(above)) + slack;
        m = next > 0x7f800000ll ? 0x7f800000 : (int)next;
        slack = slack != 0 ? (slack < (64 << 23) ? slack * 2 : slack) : (1 << 23);
    }
}

