// kernels_hdr64.hip -- LAv2 for T = HDRFloat<double>, the production kernel (round 6): k_lav2_hdr64.
//
// CPU twin: Fractal::CalcCpuPerturbationFractalLAV2<uint32_t,double,Disable> (Fractal.cpp:2545-2678) with
// LAReference::getLA / isLAStageInvalid (LAReference.cpp:1076-1134), LAInfoDeep::Prepare / Evaluate (LAInfoDeep.h:395-420),
// ATInfo::PerformAT (ATInfo.h:155-188).  Replaces mandel_1xHDR_float_perturb_lav2 as instantiated for HDRFloat<double>
// (FractalSharkGpuLib/LAKernel.cuh:3-315, GPU_Render.cu:1152-1185).  k_lav2_lit<double> (kernels.hip) stays as the
// operation-by-operation A/B reference (FS_VARIANT_LITERAL) and serves the 64-bit counters and the waypoint-resident orbit.
//
// What is different from the literal kernel -- the VALUES of every state are the literal ones, bit for bit except the sign of a
// zero part (below); what changes is how the HDRFloatComplex operations are carried out on a 64-lane wave:
//
//  * HDRFloatComplex::plus_mutable (HDRFloatComplex.h:219-247) is a four-armed function of the exponent gap d = a.e - b.e:
//    a alone (d >= 120), a + b 2^-d, a 2^d + b, b alone (d <= -120).  Compiled literally every add is four divergent arms with their
//    EXEC bookkeeping -- 50 scalar instructions per step went there (SQ_INSTS_SALU 1.29e10 against SQ_INSTS_VALU 2.45e10, round 5).
//    Here: votes on the gap.  A wave whose lanes all take the SAME arm -- nearly every wave: its lanes hold neighbouring pixels
//    at the same phase of their orbits -- runs that arm alone: "a alone" / "b alone" return the operand as it is, and the two
//    sums are  hi + ldexp(lo, -gap)  -- the same IEEE operations as the literal arm: 2^-gap is an exact power of two inside the
//    normal range, so the product lo 2^-gap and ldexp(lo, -gap) are the same correctly rounded value (also where it is
//    subnormal), and the addition commutes.  A mixed wave selects hi / lo per lane (eight 32-bit selects) and runs the same sum
//    with the shift forced to -4000 (ldexp -> +-0, which leaves hi as it is) in the lanes whose gap is 120 or more.
//    Sign of zero (mixed waves only): "hi alone" in the literal code returns hi's bits, there hi + (+-0): a -0.0 part of hi can
//    come out as +0.0.
//    No operation of this kernel tells the two apart -- every comparison treats them as equal, fabs and the exponent field
//    ignore the sign, products and sums with a non-zero operand are unaffected, and a zero part stays a zero part -- so by
//    induction every later state differs at most in the signs of its zero parts and every test (thresholds, rebase, escape)
//    has the literal outcome; the iteration count is what the literal kernel writes.
//  * HDRFloatComplex::Reduce (HDRFloatComplex.h:473-510): multiplier(-d) is the exact power of two 2^-d whenever the larger
//    biased exponent field is in [1, 2045]; ldexp(part, -d) is then the literal product.  A wave with a lane outside that range
//    (both parts zero or subnormal, an infinity) takes the literal function.
//  * The norm tests (Reduce + compareToBothPositiveReduced on a squared norm, HDRFloat.h:1150-1167): for operands that are normal
//    numbers the lexicographic compare of (exponent, mantissa in [1, 2)) IS the comparison of the real values m 2^e, and
//    ldexp(n1, e1 - e2) < n2 decides that exactly (the shifted value is exact where it is normal; where it overflows or falls
//    below the normal range the order is already decided by the exponents, and n2 >= 2^-1000 keeps a subnormal rounding away from
//    it).  A wave with a lane whose operand is not >= 2^-1000 (zero, subnormal, NaN) takes the literal tests.
//  * Records are requested one step ahead: the orbit entry / LA record a step arrives at is the one the next step leaves from.
//
// Verified against: the CPU oracle's whole C4 frame (tests/golden/frame_crcs.json, CRC ff3b12fe), the reference's golden CRC-64s
// of Views 5 / 14 in HDRFloat<double> (tests/test_gpu_goldens.py), the literal kernel on every built-in view
// (tests/test_gpu_hdr64_fast.py), sampled oracle rows in every bench line.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <type_traits>
#include <stdint.h>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"
#include "at_math.hpp"
#include "kernels.h"
#include "kernel_common.hpp"
#include "lav2_common.hpp"
#ifndef FS_H64_LA_ASM_DEBUG
#define FS_H64_LA_ASM_DEBUG 0 /* 1 (probe build): the counting instantiation runs the hand-written statements too and tallies their exits by
                                 status (statistics words 20..23 LA, 24..27 perturbation) and their half-steps / general sums (28..35) */
#endif
#include "la_step_asm.hpp"
#include "pt_step_asm.hpp"

using namespace fs;

namespace {

using C64 = hcplx64;
using R64 = hreal64;

__device__ __forceinline__ uint32_t hi_word(double v) { return (uint32_t)(__builtin_bit_cast(uint64_t, v) >> 32); }
__device__ __forceinline__ uint32_t exp_field64(double v) { return (hi_word(v) >> 20) & 0x7FFu; }

// hi + lo where hi.e >= lo.e in every lane that takes part: arms "a alone" and "a + b 2^-d" of plus_mutable
__device__ __forceinline__ C64 add_hi_lo(C64 hi, const C64 lo)
{
    const int nd = lo.e - hi.e; // <= 0
    const int n = nd > -kExpDiffIgnored ? nd : -4000;
    hi.re = hi.re + __builtin_ldexp(lo.re, n);
    hi.im = hi.im + __builtin_ldexp(lo.im, n);
    return hi;
}

// HDRFloatComplex::plus_mutable by votes on the exponent gap (see the head of this file).  nd = b.e - a.e decides the arm of the
// literal function: (-120, 0] a + b 2^nd; [1, 119] a 2^-nd + b; <= -120 a alone; >= 120 b alone.  A wave whose lanes agree on the arm
// -- nearly every wave: the lanes hold neighbouring pixels at the same phase -- runs that arm alone: five instructions (or none)
// instead of the four-way divergent function; each arm ends in an empty asm with its own comment so that the compiler keeps the
// arms apart (it would otherwise sink their common tail into one sequence behind operand copies).  kFirst: which agreement is
// asked for first -- 0: the sum with a on top (2 Z + dz, Z + dz), 1: a alone (dz (2 Z + dz) + dc at a deep zoom, where dc is
// hundreds of binades below everything else).  Mixed waves: per-lane operand select, then the same sum with the gap clamped.
#ifndef FS_H64_PT_ASM
#define FS_H64_PT_ASM 1 /* the perturbation steps of the ordered frames by hand (pt_step_asm.hpp); 0: the compiled loop, A/B.  (A first version
                           -- named registers, seven waves, the step committed with seven moves -- was slower than the compiled loop: 34.65
                           against 31.77 ms, profiles/r06r_c4_hand_written_pt_loop_ab.jsonl) */
#endif
#ifndef FS_H64_LA_ASM
#define FS_H64_LA_ASM 1 /* the LA steps of a wave whose lanes stand at one record by hand (la_step_asm.hpp); 0: the compiled loop, A/B */
#endif
#ifndef FS_H64_STAGE_SCALAR
#define FS_H64_STAGE_SCALAR 1 /* the stage's words and its first record's threshold through the scalar cache (0: A/B) */
#endif
#ifndef FS_H64_ASM_TINY
#define FS_H64_ASM_TINY 0x1p-1000 /* what the hand-written statements take for "a norm the value compare cannot be trusted with".  (Test build: 1e300 -- EVERY
                                     step then leaves its statement with status 2, the one exit no view reaches by itself, and the frames must not change.) */
#endif
#ifndef FS_H64_ASM_COLD
#define FS_H64_ASM_COLD 1 /* the first frame of a view runs the hand-written loops too (0: A/B -- 52.6 against 50.6 ms once the statements had their "alone" arms; 54.8 against 52.3 before) */
#endif
#ifndef FS_H64_LA_SCALAR
#define FS_H64_LA_SCALAR 1 /* LA records through the scalar cache where the wave's lanes agree on the record (0: A/B) */
#endif
#ifndef FS_H64_LA_PIPE
#define FS_H64_LA_PIPE 1 /* the LA loop's step length travels one step ahead (0: A/B -- 33.74 against 32.41 ms; record j + 2's Ref and length two
                            steps ahead as well: 36.5 ms with 8 waves and spills, 33.7 with 7 -- not kept) */
#endif
#ifndef FS_H64_ADDV
#define FS_H64_ADDV 2 /* 2 = one vote on the SIGN of the gap, the sum with the shift clamped in every arm; 3 = votes on the ARM (a alone / a + b 2^nd / ...), no clamp in the agreed arms: 33.85 against 33.38 ms on one box (profiles/r06f_*), off */
#endif
#if FS_H64_ADDV == 2
template <int kFirst = 0> __device__ __forceinline__ C64 hc_add_w(const C64 a, const C64 b)
{
    const bool lt = a.e < b.e;
    const uint64_t m = __builtin_amdgcn_ballot_w64(lt);
    if (m == 0ull) {
        C64 r = add_hi_lo(a, b);
        asm volatile("; hc_add_w: every lane a.e >= b.e" : "+v"(r.re), "+v"(r.im));
        return r;
    }
    if (m == __builtin_amdgcn_ballot_w64(true)) {
        C64 r = add_hi_lo(b, a);
        asm volatile("; hc_add_w: every lane a.e < b.e" : "+v"(r.re), "+v"(r.im));
        return r;
    }
    C64 hi, lo;
    hi.re = lt ? b.re : a.re, hi.im = lt ? b.im : a.im, hi.e = lt ? b.e : a.e;
    lo.re = lt ? a.re : b.re, lo.im = lt ? a.im : b.im, lo.e = lt ? a.e : b.e;
    return add_hi_lo(hi, lo);
}
#else
template <int kFirst = 0> __device__ __forceinline__ C64 hc_add_w(const C64 a, const C64 b)
{
    const int nd = b.e - a.e;
    const uint64_t all = __builtin_amdgcn_ballot_w64(true);
    if (kFirst == 1 && __builtin_amdgcn_ballot_w64(nd <= -kExpDiffIgnored) == all)
        return a;
    if (__builtin_amdgcn_ballot_w64((uint32_t)(nd + (kExpDiffIgnored - 1)) < (uint32_t)kExpDiffIgnored) == all) {
        C64 r{a.re + __builtin_ldexp(b.re, nd), a.im + __builtin_ldexp(b.im, nd), a.e};
        asm volatile("; hc_add_w: a + b 2^nd" : "+v"(r.re), "+v"(r.im));
        return r;
    }
    if (__builtin_amdgcn_ballot_w64((uint32_t)(nd - 1) < (uint32_t)(kExpDiffIgnored - 1)) == all) {
        C64 r{b.re + __builtin_ldexp(a.re, -nd), b.im + __builtin_ldexp(a.im, -nd), b.e};
        asm volatile("; hc_add_w: a 2^-nd + b" : "+v"(r.re), "+v"(r.im));
        return r;
    }
    if (kFirst != 1 && __builtin_amdgcn_ballot_w64(nd <= -kExpDiffIgnored) == all)
        return a;
    if (__builtin_amdgcn_ballot_w64(nd >= kExpDiffIgnored) == all)
        return b;
    const bool lt = nd > 0; // a.e < b.e
    C64 hi, lo;
    hi.re = lt ? b.re : a.re, hi.im = lt ? b.im : a.im, hi.e = lt ? b.e : a.e;
    lo.re = lt ? a.re : b.re, lo.im = lt ? a.im : b.im, lo.e = lt ? a.e : b.e;
    return add_hi_lo(hi, lo);
}
#endif

// max(|a|, |b|) as ONE instruction (source modifiers; the C++ form first canonicalises each operand)
__device__ __forceinline__ double max_abs64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// HDRFloatComplex::Reduce.  The larger biased exponent field of the two parts is the field of max(|re|, |im|) whenever that
// is a normal number: one v_max_f64 with |.| modifiers, one class test (the vote: a zero / subnormal pair or an infinity in some
// lane sends the wave through the literal function), v_frexp_exp for the field, two ldexp.
__device__ __forceinline__ void hc_reduce_w(C64 &a)
{
    const double m = max_abs64(a.re, a.im);
    if (__builtin_amdgcn_ballot_w64(!__builtin_amdgcn_class(m, 0x100 /* +normal */)) != 0ull) {
        hc_reduce(a);
        return;
    }
    const int x = __builtin_amdgcn_frexp_exp(m); // m = f 2^x, f in [0.5, 1): biased field - 1022, so d = x - 1
    a.re = __builtin_ldexp(a.re, 1 - x);
    a.im = __builtin_ldexp(a.im, 1 - x);
    a.e += x - 1;
}

// The two tests of a perturbation step on squared norms n1 2^e1 (|Z + dz|^2) and n2 2^e2 (|dz|^2), both as
// hr_cmp_pos(hr_reduced(.), hr_reduced(.)): escaped = 256 < first, rebase = first < second.  ONE vote for both: every operand a
// normal number well inside the range (>= 2^-1000) -- then the compares are value compares and ldexp carries them out exactly (see
// the head of this file; ldexp takes any int32 shift and saturates to 0 / inf, which is the order the exponents already decide).
__device__ __forceinline__ void step_tests_w(double n1, int e1, double n2, int e2, bool &escaped, bool &rebase)
{
    double lo;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(n1), "v"(n2));
    if (__builtin_amdgcn_ballot_w64(!(lo >= 0x1p-1000)) != 0ull) {
        const R64 N1 = hr_reduced(R64{n1, e1}), N2 = hr_reduced(R64{n2, e2});
        escaped = hr_cmp_pos(N1, R64{1.0, 8}) > 0;
        rebase = hr_cmp_pos(N1, N2) < 0;
        return;
    }
    escaped = __builtin_ldexp(n1, e1) > 256.0;
    rebase = __builtin_ldexp(n1, e1 - e2) < n2;
}

// hr_cmp_pos(hr_reduced({m1, e1}), hr_reduced({m2, e2})) < 0 for two non-negative mantissas (the LA loop's rebase test on
// Chebyshev norms), the same way
__device__ __forceinline__ bool less_w(double m1, int e1, double m2, int e2)
{
    double lo;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(m1), "v"(m2));
    if (__builtin_amdgcn_ballot_w64(!(lo >= 0x1p-1000)) != 0ull)
        return hr_cmp_pos(hr_reduced(R64{m1, e1}), hr_reduced(R64{m2, e2})) < 0;
    return __builtin_ldexp(m1, e1 - e2) < m2;
}

__device__ __forceinline__ const fs_la_hdr64_u32 *la_at_off(const fs_la_hdr64_u32 *__restrict__ las, uint32_t byte_off)
{
    return (const fs_la_hdr64_u32 *)((const char *)las + byte_off);
}

// HDRFloatComplex::chebychevNorm's mantissa
__device__ __forceinline__ double cheb64(const C64 a) { return max_abs64(a.re, a.im); }

// records by a 32-bit BYTE offset from a wave-uniform base (one scalar base + one vector offset per load; the host only launches
// this kernel when the orbit and the table stay below 4 GB, renderer.cpp)
__device__ __forceinline__ C64 z_at_off(const FsZ64 *__restrict__ z, uint32_t byte_off)
{
    const FsZ64 *p = (const FsZ64 *)((const char *)z + byte_off);
    return C64{p->re, p->im, p->e};
}

// Lane -> pixel under a recorded order, XCD-aware.  The hardware hands workgroup b to XCD b mod 8, each XCD with an L2 of its own; in
// plain launch order eight NEIGHBOURING workgroups -- whose pixels, adjacent in the count order, walk the same LA records -- land on
// eight different L2s and every L2 sees the whole 15-MB table.  Here XCD x takes runs of FS_H64_XCD_RUN consecutive positions of the
// order (virtual workgroup v = group * 8 R + x * R + r for the r-th workgroup the XCD receives in the group), so the workgroups that
// share records share an L2, while every XCD still gets an even share of every stretch of the order.
#ifndef FS_H64_XCD_RUN
#define FS_H64_XCD_RUN 0
#endif
__device__ __forceinline__ void ordered_pixel_xcd(const FsFrame &f, const uint32_t *__restrict__ order, uint32_t &X, uint32_t &L)
{
    uint32_t b = blockIdx.y * gridDim.x + blockIdx.x;
#if FS_H64_XCD_RUN > 0
    constexpr uint32_t R = FS_H64_XCD_RUN, G = 8u * R;
    const uint32_t nb = gridDim.x * gridDim.y;
    if (b < nb / G * G) {
        const uint32_t g = b / G, w = b % G;
        b = g * G + (w % 8u) * R + w / 8u;
    }
#endif
    const uint32_t slot = b * blockDim.x + threadIdx.x;
    const uint32_t n = f.rounded_width * ((f.local_rows + 7u) & ~7u);
    if (slot < n) {
        const uint32_t id = order[slot];
        L = id / f.rounded_width;
        X = id - L * f.rounded_width;
    } else {
        X = 0xFFFFFFFFu, L = 0xFFFFFFFFu;
    }
}

// Statistics words of the counting build (fs_read_step_count / tools): [8] steps whose adds ran the mixed (select) form, [9] wave
// steps of the perturbation loop, [10] wave steps of the LA loop
// FS_H64_WAVES (A/B builds): 8 = the register allocator is held to 64 registers (8 waves per SIMD; it spills three or four dwords),
// 0 = left alone (67 registers, 7 waves)
#ifndef FS_H64_WAVES
#define FS_H64_WAVES 8
#endif
#if FS_H64_WAVES == 8
#define FS_H64_OCCUPANCY __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define FS_H64_OCCUPANCY
#endif
// kAtInKernel: PerformAT is iterated here (at_perform; frames without the AT pass of their own) -- false: its results come from
// A.at_res (fsk_at_pass64), and the instantiation carries neither the loop nor its registers
template <int Mode, bool kStats, bool kAtInKernel> __global__ void __launch_bounds__(256) FS_H64_OCCUPANCY k_lav2_hdr64(FsLav2ArgsT<double> A)
{
    using F = double;
    using LaRec = fs_la_hdr64_u32;
    static_assert(sizeof(LaRec) == 128 && offsetof(LaRec, Ref) == 0 && offsetof(LaRec, ZCoeff) == 24 && offsetof(LaRec, CCoeff) == 48 &&
                      offsetof(LaRec, LAThreshold) == 72 && offsetof(LaRec, StepLength) == 120 && offsetof(LaRec, NextStageLAIndex) == 124 &&
                      offsetof(fs_cplx_hdr64, im) == 8 && offsetof(fs_cplx_hdr64, e) == 16 && offsetof(fs_real_hdr64, e) == 8,
                  "la_step_asm.hpp reads the record by these offsets");
    uint32_t X, L;
    if (A.pixel_order)
        ordered_pixel_xcd(A.frame, A.pixel_order, X, L);
    else if (A.tile_order) // (a first frame: tiles in the order of a sampled PerformAT count, kernels_tile_sample.hip)
        ordered_tile_pixel(A.tile_order, A.tiles_x, X, L);
    else
        tile_pixel(X, L);
    uint64_t c_at = 0, c_la = 0, c_pt = 0, c_px = 0, c_at_exec = 0, c_at_own = 0;
    uint32_t px_cost = 0;
    const bool in_buffer = X < A.frame.width && L < A.frame.local_rows;
    const uint32_t Y = in_buffer ? global_row(A.frame, L) : 0xFFFFFFFFu;
    const bool live = in_buffer && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const uint32_t n_iterations = A.n_iterations;
        R64 deltaReal, deltaImaginary;
        pixel_delta<F>(A.coords, X, Y, deltaReal, deltaImaginary);
        const C64 dc = hc_from_hr(deltaReal, deltaImaginary);
        C64 dz = hc_from_native<F>(F(0), F(0)); // {0,0}: zero with exponent 0 (Fractal.cpp:2565)
        uint32_t iterations = 0;

        if constexpr (Mode != FS_MODE_PO && !kAtInKernel) {
            // PerformAT ran in its own pass (fsk_at_pass64): its result instead of the iteration
            const FsAtRes ar = A.at_res[(size_t)L * A.frame.rounded_width + X];
            if (ar.i != 0xFFFFFFFFu) {
                dz = C64{ar.re, ar.im, ar.e};
                iterations = ar.i * A.at.StepLength;
            }
        }
        if constexpr (Mode != FS_MODE_PO && kAtInKernel) {
            if (A.la_valid && A.use_at && hr_cmp_pos(hc_cheb(dc), ldr(A.at.ThresholdC)) <= 0) {
                const uint32_t at_step = A.at.StepLength;
                const uint32_t ATMaxIt = n_iterations / at_step;
                C64 c = hc_add(hc_mul(dc, ldc(A.at.CCoeff)), ldc(A.at.RefC));
                hc_reduce(c);
                C64 z;
                uint32_t i, i_exec = 0, i_own = 0;
                at_perform<F, uint32_t>(c, ldr(A.at.SqrEscapeRadius), ATMaxIt, z, i, &i_exec, &i_own);
                px_cost = i_own > 0xFFFFFu ? 0xFFFFFu : i_own;
                C64 d0 = hc_mul(z, ldc(A.at.InvZCoeff));
                hc_reduce(d0);
                dz = d0;
                iterations = i * at_step;
                if (kStats)
                    c_at = i, c_at_exec = i_exec, c_at_own = i_own;
            }
        }

        // counting build: how the waves' lanes agree (statistics words 8..15, tools/c4_arm_probe.py) -- wave steps of the perturbation
        // loop [8], of them with lanes on different arms of the add 2Z + dz [9], dz t + dc [10], Z + dz [11], with a rebasing lane [12];
        // wave steps of the LA loop [13], of them with lanes on different arms in any of its three adds [14], with a rebasing lane [15]
        uint32_t w_pt = 0, w_mixA = 0, w_mixB = 0, w_mixC = 0, w_reb = 0, w_la = 0, w_lamix = 0, w_lareb = 0;
        // ... and how their ADDRESSES agree (words 16..19): LA wave steps whose lanes all read the same record [16], the number of
        // distinct records summed over the LA wave steps [17]; the same two for the orbit entry of the perturbation steps [18], [19]
        uint32_t w_launi = 0, w_ladist = 0, w_ptuni = 0, w_ptdist = 0;
        // ... and on which ARM they agree (words 28..39, the counting build without FS_H64_LA_ASM_DEBUG): wave steps with every lane on
        // "a alone" / "a + b 2^nd" for 2 Ref + dz [28, 29], a alone / a on top / b on top / b alone for newDz ZCoeff + dc CCoeff [30..33],
        // a alone / a on top for next Ref + dz' [34, 35], for 2 Z + dz [36, 37] and for Z' + dz' [38, 39]
        uint32_t w_arm[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto all_on = [](int arm, int k) { return __builtin_amdgcn_ballot_w64(arm == k) == __builtin_amdgcn_ballot_w64(true) ? 1u : 0u; };
        auto distinct = [](uint32_t v) {
            uint64_t m = __builtin_amdgcn_ballot_w64(true);
            uint32_t n = 0;
            while (m != 0ull) {
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_ctzll(m));
                m &= ~__builtin_amdgcn_ballot_w64(v == f);
                n++;
            }
            return n;
        };
        auto arm_of = [](const C64 a, const C64 b) {
            const int nd = b.e - a.e;
            return nd <= -kExpDiffIgnored ? 0 : (nd <= 0 ? 1 : (nd < kExpDiffIgnored ? 2 : 3));
        };
        auto mixed = [](int arm) {
            const uint64_t all = __builtin_amdgcn_ballot_w64(true);
            bool same = false;
            for (int k = 0; k < 4; k++)
                same = same || __builtin_amdgcn_ballot_w64(arm == k) == all;
            return same ? 0u : 1u;
        };
        uint32_t n_la = 0; // LA steps of this pixel (the frame's own cost record, FsLav2ArgsT::pixel_cost)
        uint32_t RefIteration = 0;
        const uint32_t MaxRefIteration = A.orbit_count - 1;
        const uint32_t period = A.period;
        if (iterations != 0 && !(RefIteration < MaxRefIteration) && period != 0)
            RefIteration = RefIteration % period; // (Fractal.cpp:2590-2591)

        if (Mode != FS_MODE_PO) {
            uint32_t CurrentLAStage = A.la_valid ? A.stage_count : 0;
            const R64 dcCheb = hc_cheb(dc);
            while (CurrentLAStage > 0) {
                CurrentLAStage--;
#if FS_H64_STAGE_SCALAR
                // (the stage number is the same in every lane that is still in this loop -- they all count down from stage_count, one per
                // trip -- which the compiler cannot see behind the loop's divergent exits: said explicitly, the stage's two words and its
                // first record's LAThresholdC come through the scalar cache instead of three dependent vector loads)
                typedef const __attribute__((address_space(4))) fs_la_stage_u32 *CStage;
                typedef const __attribute__((address_space(4))) LaRec *CRec0;
                const CStage sp = (CStage)(uintptr_t)A.stages + (uint32_t)__builtin_amdgcn_readfirstlane((int)CurrentLAStage);
                const uint32_t LAIndex = sp->LAIndex;
                {
                    const CRec0 r0 = (CRec0)(uintptr_t)A.las + LAIndex;
                    const int cmp = hr_cmp_pos(dcCheb, R64{r0->LAThresholdC.m, r0->LAThresholdC.e});
                    const bool invalid = A.parity == FS_PARITY_LITERAL ? (cmp < 0) : (cmp >= 0);
                    if (invalid)
                        continue;
                }
                const uint32_t MacroItCount = sp->MacroItCount;
#else
                const uint32_t LAIndex = A.stages[CurrentLAStage].LAIndex;
                {
                    const int cmp = hr_cmp_pos(dcCheb, ldr(A.las[LAIndex].LAThresholdC));
                    const bool invalid = A.parity == FS_PARITY_LITERAL ? (cmp < 0) : (cmp >= 0);
                    if (invalid)
                        continue;
                }
                const uint32_t MacroItCount = A.stages[CurrentLAStage].MacroItCount;
#endif
                const uint32_t base_off = LAIndex * (uint32_t)sizeof(LaRec); // byte offset of the stage's first record
                uint32_t j = RefIteration;
                // The Ref of record j + 1, read for the rebase test of step j, is the Ref step j + 1 starts from: it travels in
                // RefJ; the rest of record j (coefficients, threshold, lengths) is requested while step j - 1 still computes.
                // ... and so does its step length, which decides first whether the step may be taken at all: a step that had to wait
                // for its own record's length before it could ask for the coefficients made two round trips to the cache.
                // One LA step: leaves from record j with its Ref in RJ and its step length in lJ (both read one step ahead), reads the
                // rest of record j and, into RN / lN, the Ref and length of record j + 1.  The loop calls it twice per trip with the two
                // register sets' roles exchanged (no copies).  -> true: the stage is left (RefIteration set).
                // (round 6) ... and from the SCALAR cache when the wave's lanes all stand at the same record -- 96.5 % of the LA wave
                // steps of C4's frame in the count order, 76 % in the tile mapping (tools/c4_arm_probe.py): the eleven vector loads of
                // a step cost the CU's one texture-address unit >= 4 cycles each whatever their lanes read, and with four SIMDs
                // behind it that unit, not the vector ALU, set the pace of this loop (TCP_TOTAL_CACHE_ACCESSES 1.7e10 per frame = 0.87
                // per CU cycle).  kUni: the record is read through a constant-address-space pointer at a wave-uniform offset
                // (s_load), its fields are scalar operands of the same operations, and RJ / lJ are not needed (Ref and length of
                // record j are in the record itself).
                // a + b where b is usually 120 binades and more below a in every lane of the wave (dz against an orbit value at a deep zoom)
                auto add_a_first = [](const C64 a, const C64 b) __attribute__((always_inline)) {
                    if (__builtin_amdgcn_ballot_w64(b.e - a.e <= -kExpDiffIgnored) == __builtin_amdgcn_ballot_w64(true))
                        return a;
                    return hc_add_w(a, b);
                };
                auto la_body = [&](auto LAj, auto uni, const C64 &RJ_in, const uint32_t &lJ_in, C64 &RN, uint32_t &lN)
                                   __attribute__((always_inline)) -> bool {
                    constexpr bool kUni = decltype(uni)::value;
#define FS_LDC(P, F) C64{(P)->F.re, (P)->F.im, (P)->F.e}
#define FS_LDR(P, F) R64{(P)->F.m, (P)->F.e}
#if !FS_H64_LA_PIPE
                    const uint32_t l = LAj->StepLength; // (A/B: the step waits for its own record's length first, as the literal kernel does)
                    const C64 RJ = RJ_in;
#else
                    const uint32_t l = kUni ? LAj->StepLength : lJ_in;
                    const C64 RJ = kUni ? FS_LDC(LAj, Ref) : RJ_in;
#endif
                    const uint32_t next_stage = LAj->NextStageLAIndex;
                    const C64 ZCoeff = FS_LDC(LAj, ZCoeff), CCoeff = FS_LDC(LAj, CCoeff);
                    const R64 thr = FS_LDR(LAj, LAThreshold);
                    RN = FS_LDC(LAj + 1, Ref);
                    lN = LAj[1].StepLength;
#undef FS_LDC
#undef FS_LDR
                    if (kStats) {
                        w_la++;
                        const uint32_t nd_ = distinct(base_off + j * (uint32_t)sizeof(LaRec));
                        w_ladist += nd_;
                        w_launi += nd_ == 1u ? 1u : 0u;
                    }
                    if (iterations + l > n_iterations) { // the step would pass the iteration limit: unusable
                        RefIteration = next_stage;
                        return true;
                    }
                    if (kStats) {
                        const int arm = arm_of(C64{RJ.re, RJ.im, clamp_exp(RJ.e + 1)}, dz);
                        w_lamix |= mixed(arm) << 8;
                        w_arm[0] += all_on(arm, 0), w_arm[1] += all_on(arm, 1);
                    }
                    C64 newDz = hc_mul(dz, add_a_first(C64{RJ.re, RJ.im, clamp_exp(RJ.e + 1)}, dz));
                    hc_reduce_w(newDz);
                    if (hr_cmp_pos(R64{cheb64(newDz), newDz.e}, thr) >= 0) { // LAInfoDeep::Prepare's unusable
                        RefIteration = next_stage;
                        return true;
                    }
                    iterations += l;
                    n_la++;
                    if (kStats) {
                        const int arm = arm_of(hc_mul(newDz, ZCoeff), hc_mul(dc, CCoeff));
                        w_lamix |= mixed(arm) << 8;
                        for (int k = 0; k < 4; k++)
                            w_arm[2 + k] += all_on(arm, k);
                    }
                    // (dc CCoeff 120 binades and more below newDz ZCoeff in every lane -- nine steps of ten at C4's zoom: plus_mutable returns
                    // its first operand, and the second product need not be formed)
                    if (__builtin_amdgcn_ballot_w64(clamp_exp(dc.e + CCoeff.e) - clamp_exp(newDz.e + ZCoeff.e) <= -kExpDiffIgnored) ==
                        __builtin_amdgcn_ballot_w64(true))
                        dz = hc_mul(newDz, ZCoeff);
                    else
                        dz = hc_add_w(hc_mul(newDz, ZCoeff), hc_mul(dc, CCoeff));
                    if (kStats) {
                        const int arm = arm_of(RN, dz);
                        w_arm[6] += all_on(arm, 0), w_arm[7] += all_on(arm, 1);
                        w_lamix |= mixed(arm) << 8;
                        w_lamix = (w_lamix & 0xFFu) + (w_lamix >> 8 ? 1u : 0u); // (one per wave step with any mixed add)
                    }
                    const C64 complex0 = add_a_first(RN, dz);
                    j++;
                    const bool la_rebase = less_w(cheb64(complex0), complex0.e, cheb64(dz), dz.e) || j >= MacroItCount;
                    if (kStats)
                        w_lareb += __builtin_amdgcn_ballot_w64(la_rebase) != 0ull ? 1u : 0u;
                    if (la_rebase) {
                        dz = complex0;
                        j = 0;
                        RN = ldc(la_at_off(A.las, base_off)->Ref);
                        lN = la_at_off(A.las, base_off)->StepLength;
                    }
                    return false;
                };
                auto la_step = [&](const C64 &RJ, const uint32_t &lJ, C64 &RN, uint32_t &lN) __attribute__((always_inline)) -> bool {
                    const uint32_t off = base_off + j * (uint32_t)sizeof(LaRec);
#if FS_H64_LA_SCALAR
                    const uint32_t uoff = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
                    if (__builtin_amdgcn_ballot_w64(off == uoff) == __builtin_amdgcn_ballot_w64(true)) {
                        typedef const __attribute__((address_space(4))) LaRec *CRec;
                        return la_body((CRec)((uintptr_t)A.las + uoff), std::true_type{}, RJ, lJ, RN, lN);
                    }
#endif
                    return la_body(la_at_off(A.las, off), std::false_type{}, RJ, lJ, RN, lN);
                };
#if FS_H64_LA_ASM
                // (frames in a recorded order only -- kAtInKernel is the first frame of a view, in the tile mapping, where a quarter of
                // the LA steps have lanes at different records and every one of them would leave the statement for a compiled step and
                // come back: 54.8 against 52.3 ms)
                if constexpr ((!kStats || FS_H64_LA_ASM_DEBUG) && (!kAtInKernel || FS_H64_ASM_COLD)) {
                    // the hand-written loop for the steps whose lanes stand at one record (la_step_asm.hpp); what it hands back --
                    // lanes at different records, a product that Reduce's fast form does not cover, a norm below 2^-1000 in the
                    // rebase test -- takes ONE compiled step (or its rebase test) and goes back in
                    while (iterations < n_iterations) {
                        double yr, yi, t0, t1, t2, t3, t4, t5;
                        int ye, i0, i1, i2, i3, i4, i5;
                        uint32_t st, so, sa;
                        uint64_t run, leftm, sx, m0, m1;
#if FS_H64_LA_ASM_DEBUG
                        uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#define FS_DBG_CNT_OPS , [c0] "+s"(c0), [c1] "+s"(c1), [c2] "+s"(c2), [c3] "+s"(c3)
#else
#define FS_DBG_CNT_OPS
#endif
                        asm volatile(FS_LA_UNIFORM_LOOP
                                     : [xr] "+v"(dz.re), [xi] "+v"(dz.im), [xe] "+v"(dz.e), [j] "+v"(j), [it] "+v"(iterations), [nla] "+v"(n_la),
                                       [refit] "+v"(RefIteration), [yr] "=&v"(yr), [yi] "=&v"(yi), [ye] "=&v"(ye), [t0] "=&v"(t0),
                                       [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [i0] "=&v"(i0),
                                       [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3), [i4] "=&v"(i4), [i5] "=&v"(i5), [st] "=&s"(st),
                                       [run] "=&s"(run), [left] "=&s"(leftm), [sx] "=&s"(sx), [so] "=&s"(so), [sa] "=&s"(sa), [m0] "=&s"(m0), [m1] "=&s"(m1) FS_DBG_CNT_OPS
                                     : [dcr] "v"(dc.re), [dci] "v"(dc.im), [dce] "v"(dc.e), [boff] "v"(base_off), [macro] "v"(MacroItCount), [m4k] "v"(-4000),
                                       [las] "s"(A.las), [nit] "s"(n_iterations), [cls] "s"(0x100), [tiny] "s"(FS_H64_ASM_TINY)
                                     : "vcc", "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48",
                                       "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s64", "s65",
                                       "s66", "s67", "s68", "s69", "s70", "s71");
#if FS_H64_LA_ASM_DEBUG
                        if (kStats && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) ==
                                          (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
                        {
                            atomicAdd((unsigned long long *)&A.stats[20 + st], 1ull); // exits of the statement by status
                            atomicAdd((unsigned long long *)&A.stats[28], (unsigned long long)c0); // half-steps begun
                            atomicAdd((unsigned long long *)&A.stats[29], (unsigned long long)c1); // general sums: 2 Ref + dz
                            atomicAdd((unsigned long long *)&A.stats[30], (unsigned long long)c2); // newDz ZCoeff + dc CCoeff
                            atomicAdd((unsigned long long *)&A.stats[31], (unsigned long long)c3); // next Ref + dz'
                        }
#endif
                        if (__builtin_amdgcn_inverse_ballot_w64(leftm))
                            break; // this lane has left the stage (RefIteration is the record's NextStageLAIndex)
                        if (!(iterations < n_iterations))
                            break;
                        const LaRec *__restrict__ LAj = la_at_off(A.las, base_off + j * (uint32_t)sizeof(LaRec));
                        if (st == 2u) { // dz' and j + 1 are in place: complex0, the rebase test and its consequence
                            const C64 complex0 = hc_add_w(ldc(LAj->Ref), dz);
                            if (less_w(cheb64(complex0), complex0.e, cheb64(dz), dz.e) || j >= MacroItCount) {
                                dz = complex0;
                                j = 0;
                            }
                        } else {
                            // one compiled step with vector loads (it reads its own record whole: handing Ref and length on
                            // from step to step, as the compiled loop does, costs more registers than this kernel has next to the
                            // statement's -- 37.9 against 30.0 ms with spills in the loop)
                            C64 RNv;
                            uint32_t lNv;
                            if (la_body(LAj, std::false_type{}, ldc(LAj->Ref), LAj->StepLength, RNv, lNv))
                                break;
                        }
                    }
                } else
#endif
                {
                C64 RefA = hc_zero<F>(), RefB = hc_zero<F>();
                uint32_t lA = 0, lB = 0;
                if (iterations < n_iterations) {
                    const LaRec *__restrict__ first = la_at_off(A.las, base_off + j * (uint32_t)sizeof(LaRec));
                    RefA = ldc(first->Ref);
                    lA = first->StepLength;
                }
                while (iterations < n_iterations) {
                    if (la_step(RefA, lA, RefB, lB))
                        break;
                    if (!(iterations < n_iterations))
                        break;
                    if (la_step(RefB, lB, RefA, lA))
                        break;
                }
                }
                if (iterations >= n_iterations)
                    break;
            }
        }

        const uint32_t it_la = iterations;
        if (Mode != FS_MODE_LAO) {
            const FsZ64 *__restrict__ zr = A.zref;
            uint32_t zoff = RefIteration * (uint32_t)sizeof(FsZ64); // byte offset of the entry the step leaves from
            const uint32_t max_off = MaxRefIteration * (uint32_t)sizeof(FsZ64);
            // One step: leaves from the entry in ZH, arrives at the entry it loads into ZN.  The loop below calls it twice per trip
            // with the two registers' roles exchanged, so that "the entry a step arrives at is the entry the next one leaves from"
            // costs no copy.  (As in the literal kernel: one orbit load per step, not two; Reduce(z) before |z|^2 only re-labels z.)
            // -> true: this lane's pixel has escaped.
            auto pt_step = [&](const C64 &ZH, C64 &ZN) __attribute__((always_inline)) -> bool {
                zoff += (uint32_t)sizeof(FsZ64);
                ZN = z_at_off(zr, zoff);
                if (kStats) {
                    w_pt++;
                    const uint32_t nd_ = distinct(zoff);
                    w_ptdist += nd_;
                    w_ptuni += nd_ == 1u ? 1u : 0u;
                    const int arm = arm_of(C64{ZH.re, ZH.im, ZH.e + 1}, dz);
                    w_mixA += mixed(arm);
                    w_arm[8] += all_on(arm, 0), w_arm[9] += all_on(arm, 1);
                }
                const C64 cur = hc_add_w(C64{ZH.re, ZH.im, ZH.e + 1}, dz); // (hc_mul2: x * 1.0 is x; e + 1 needs no clamp)
                if (kStats)
                    w_mixB += mixed(arm_of(hc_mul(dz, cur), dc));
                C64 q = hc_add_w<1>(hc_mul(dz, cur), dc);
                hc_reduce_w(q);
                dz = q;
                if (kStats)
                    c_pt++;
                if (kStats) {
                    const int arm = arm_of(ZN, dz);
                    w_mixC += mixed(arm);
                    w_arm[10] += all_on(arm, 0), w_arm[11] += all_on(arm, 1);
                }
                C64 complex0 = hc_add_w(ZN, dz);
                const double n1 = complex0.re * complex0.re + complex0.im * complex0.im;
                const double n2 = dz.re * dz.re + dz.im * dz.im;
                bool escaped, rebase;
                step_tests_w(n1, complex0.e << 1, n2, dz.e << 1, escaped, rebase);
                if (escaped)
                    return true;
                rebase = rebase || zoff >= max_off; // (RefIteration >= MaxRefIteration)
                if (kStats)
                    w_reb += __builtin_amdgcn_ballot_w64(rebase) != 0ull ? 1u : 0u;
                if (rebase) {
                    hc_reduce_w(complex0);
                    dz = complex0;
                    zoff = 0;
                    ZN = z_at_off(zr, 0u);
                }
                return false;
            };
            C64 ZA = hc_zero<F>(), ZB = hc_zero<F>();
            if (iterations < n_iterations)
                ZA = z_at_off(zr, zoff);
#if FS_H64_PT_ASM
            if constexpr ((!kStats || FS_H64_LA_ASM_DEBUG) && (!kAtInKernel || FS_H64_ASM_COLD)) {
                // ---- the perturbation loop by hand (pt_step_asm.hpp) for the frames in a recorded order; what the statement hands
                // back takes one compiled step (status 1, 3) or the compiled tests of the step it has computed (status 2)
                const uint32_t z0re_lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_bit_cast(uint64_t, zr[0].re));
                const uint32_t z0re_hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(__builtin_bit_cast(uint64_t, zr[0].re) >> 32));
                const uint32_t z0im_lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_bit_cast(uint64_t, zr[0].im));
                const uint32_t z0im_hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(__builtin_bit_cast(uint64_t, zr[0].im) >> 32));
                const int32_t z0e = __builtin_amdgcn_readfirstlane(zr[0].e);
                const uint64_t z0re = ((uint64_t)z0re_hi << 32) | z0re_lo, z0im = ((uint64_t)z0im_hi << 32) | z0im_lo;
                bool running = iterations < n_iterations;
                uint64_t run = __builtin_amdgcn_ballot_w64(running);
                while (run != 0ull) {
                    double yr, yi, t0, t1, t2, t3, t4;
                    int ye, i0, i1, i2, i3;
                    uint32_t st;
                    uint64_t sx, mesc, mreb, mend;
#if FS_H64_LA_ASM_DEBUG
                    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#endif
                    asm volatile(FS_PT_LOOP
                                 : [xr] "+v"(dz.re), [xi] "+v"(dz.im), [xe] "+v"(dz.e), [zoff] "+v"(zoff), [iter] "+v"(iterations),
                                   [run] "+s"(run), [yr] "=&v"(yr), [yi] "=&v"(yi), [ye] "=&v"(ye), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
                                   [t3] "=&v"(t3), [t4] "=&v"(t4), [i0] "=&v"(i0), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3),
                                   [st] "=&s"(st), [sx] "=&s"(sx), [mesc] "=&s"(mesc), [mreb] "=&s"(mreb), [mend] "=&s"(mend) FS_DBG_CNT_OPS
                                 : [dcr] "v"(dc.re), [dci] "v"(dc.im), [dce] "v"(dc.e), [m4k] "v"(-4000), [zb] "s"(zr), [niter] "s"(n_iterations),
                                   [maxoff] "s"(max_off), [cls] "s"(0x100), [tiny] "s"(FS_H64_ASM_TINY), [c256] "s"(256.0), [z0re] "s"(z0re), [z0im] "s"(z0im), [z0e] "s"(z0e)
                                 : "vcc", "scc", "memory", "v52", "v53", "v54", "v55", "v56", "v58", "v59", "v60", "v61", "v62");
#if FS_H64_LA_ASM_DEBUG
                    if (kStats && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) ==
                                      (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
                    {
                        atomicAdd((unsigned long long *)&A.stats[24 + st], 1ull); // exits of the statement by status
                        atomicAdd((unsigned long long *)&A.stats[32], (unsigned long long)c0); // half-steps begun
                        atomicAdd((unsigned long long *)&A.stats[33], (unsigned long long)c1); // general sums: 2 Z + dz
                        atomicAdd((unsigned long long *)&A.stats[34], (unsigned long long)c2); // steps with a rebasing lane
                        atomicAdd((unsigned long long *)&A.stats[35], (unsigned long long)c3); // general sums: Z' + dz'
                    }
#endif
                    if (st == 0u)
                        break;
                    running = __builtin_amdgcn_inverse_ballot_w64(run);
                    if (running) {
                        if (st == 2u) { // dz' is in place: the tests, the rebase, the count
                            zoff += (uint32_t)sizeof(FsZ64);
                            ZA = z_at_off(zr, zoff);
                            C64 complex0 = hc_add_w(ZA, dz);
                            const double n1 = complex0.re * complex0.re + complex0.im * complex0.im;
                            const double n2 = dz.re * dz.re + dz.im * dz.im;
                            bool escaped, rebase;
                            step_tests_w(n1, complex0.e << 1, n2, dz.e << 1, escaped, rebase);
                            if (escaped) {
                                running = false;
                            } else {
                                if (rebase || zoff >= max_off) {
                                    hc_reduce_w(complex0);
                                    dz = complex0;
                                    zoff = 0;
                                    ZA = z_at_off(zr, 0u);
                                }
                                iterations++;
                                running = iterations < n_iterations;
                            }
                        } else if (ZA = z_at_off(zr, zoff), pt_step(ZA, ZB)) { // one step through the compiled code for every lane still running
                            running = false;
                        } else {
                            iterations++;
                            ZA = ZB;
                            running = iterations < n_iterations;
                        }
                    }
                    run = __builtin_amdgcn_ballot_w64(running);
                }
            } else
#endif
            while (iterations < n_iterations) {
                if (pt_step(ZA, ZB))
                    break;
                iterations++;
                if (!(iterations < n_iterations))
                    break;
                if (pt_step(ZB, ZA))
                    break;
                iterations++;
            }
        }
        store_iter(A.out, A.frame, L, X, iterations);
        if (kStats) {
            c_la = n_la;
            // one tally per wave: every lane that took part in a wave step counted it, so the first active lane's counters are the
            // wave's for the steps it was in -- the longest lane's would be better; the first lane's are a sample, stated as such
            if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) ==
                (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {
                const uint32_t v[12] = {w_pt, w_mixA, w_mixB, w_mixC, w_reb, w_la, w_lamix, w_lareb, w_launi, w_ladist, w_ptuni, w_ptdist};
                for (int k = 0; k < 12; k++)
                    atomicAdd((unsigned long long *)&A.stats[8 + k], (unsigned long long)v[k]);
                if (!FS_H64_LA_ASM_DEBUG)
                    for (int k = 0; k < 12; k++)
                        atomicAdd((unsigned long long *)&A.stats[28 + k], (unsigned long long)w_arm[k]);
            }
        }
        if (A.pixel_cost) {
            // What this pixel cost THIS kernel: its LA steps and its perturbation steps (PerformAT runs in a pass of its own).  Few
            // distinct values, so the sort that follows leaves pixels of equal cost in buffer order -- neighbours stay together.
            // (with PerformAT inside the kernel -- no AT pass -- the round-5 key: AT iterations above perturbation steps)
            const uint32_t pt = iterations - it_la;
            const uint32_t steps = n_la + pt;
            A.pixel_cost[(size_t)L * A.frame.rounded_width + X] =
                !kAtInKernel ? (steps > 0xFFFFu ? 0xFFFFu : steps) : ((px_cost << 12) | (pt > 0xFFFu ? 0xFFFu : pt));
        }
    }
    if (kStats) {
        add_stats(A.stats, c_at, c_la, c_pt, c_px);
        uint64_t e = c_at_exec;
        for (int off = 32; off > 0; off >>= 1)
            e += __shfl_down(e, off);
        uint64_t o = c_at_own;
        for (int off = 32; off > 0; off >>= 1)
            o += __shfl_down(o, off);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd((unsigned long long *)&A.stats[5], (unsigned long long)e);
            atomicAdd((unsigned long long *)&A.stats[6], (unsigned long long)o);
        }
    }
}

dim3 tile_grid64(const FsFrame &f) { return dim3((f.width + 31) / 32, (f.local_rows + 7) / 8, 1); } // tile_pixel()

} // namespace

void fsk_lav2_hdr64_fast(const FsLav2ArgsT<double> &A, int mode, bool stats, hipStream_t s)
{
    const dim3 g = tile_grid64(A.frame), b(256);
#define FS_LAUNCH64F(M)                                                                                             \
    do {                                                                                                            \
        if (A.at_res && (M) != FS_MODE_PO) {                                                                        \
            if (stats)                                                                                              \
                hipLaunchKernelGGL((k_lav2_hdr64<M, true, false>), g, b, 0, s, A);                                  \
            else                                                                                                    \
                hipLaunchKernelGGL((k_lav2_hdr64<M, false, false>), g, b, 0, s, A);                                 \
        } else {                                                                                                    \
            if (stats)                                                                                              \
                hipLaunchKernelGGL((k_lav2_hdr64<M, true, true>), g, b, 0, s, A);                                   \
            else                                                                                                    \
                hipLaunchKernelGGL((k_lav2_hdr64<M, false, true>), g, b, 0, s, A);                                  \
        }                                                                                                           \
    } while (0)
    if (mode == FS_MODE_FULL)
        FS_LAUNCH64F(FS_MODE_FULL);
    else if (mode == FS_MODE_PO)
        FS_LAUNCH64F(FS_MODE_PO);
    else
        FS_LAUNCH64F(FS_MODE_LAO);
#undef FS_LAUNCH64F
}
