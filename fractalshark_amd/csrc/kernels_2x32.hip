// kernels_2x32.hip -- LAv2 for T = HDRFloat<CudaDblflt> ("2x32 + exponent"): RenderAlgorithm
// GpuHDRx2x32PerturbedLAv2[PO|LAO] (GPU_Render.cu:1152-1185).  Compiled with -ffp-contract=off (df32_math.hpp).
//
// This numeric type has no CPU RenderAlgorithm in the reference, so the semantics restated here are those of the
// CUDA kernel itself, mandel_1xHDR_float_perturb_lav2<.., HDRFloat<CudaDblflt<dblflt>>, ..>
// (FractalSharkGpuLib/LAKernel.cuh:3-315): scalar-HDR perturbation step (HDRFloat::custom_perturb3,
// HDRFloat.h:797-812), escape when the reduced |z|^2 has exponent >= 2 (compareToBothPositiveReducedTemplate<256>,
// HDRFloat.h:1169-1184), GPU direction of the LA stage-validity test (GPU_LAReference.h:238-254), delta-c built
// without intermediate Reduce (LAKernel.cuh:41-42).  The checker is oracle/gpu_ref_2x32.cpp (parity unpinned: see
// its header).
//
// Decomposition is the same as the other iteration kernels: one lane per (sub)pixel, block = 4 waves x 64 lanes over
// a 64-pixel row segment x 4 rows, orbit / LA table read straight from L2 (every lane of a wave reads the same or a
// neighbouring entry).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fs_layout.h"
#include "df32_math.hpp"
#include "kernels.h"
#include "kernel_common.hpp"

using namespace fs;

namespace {

using HR = hreal<df32>;
using HC = hcplx<df32>;

__device__ __forceinline__ HR ldr(const fs_real_2x32 &r) { return HR{df32(r.head, r.tail), r.e}; }
__device__ __forceinline__ HC ldc(const fs_cplx_2x32 &c)
{
    return HC{df32(c.re_head, c.re_tail), df32(c.im_head, c.im_tail), c.e};
}
__device__ __forceinline__ HR orbit_x(const fs_orbit_2x32 *__restrict__ o, uint32_t i)
{
    return HR{df32(o[i].x_head, o[i].x_tail), o[i].ex};
}
__device__ __forceinline__ HR orbit_y(const fs_orbit_2x32 *__restrict__ o, uint32_t i)
{
    return HR{df32(o[i].y_head, o[i].y_tail), o[i].ey};
}

// Sequential access to a SimpleCompression orbit that stays compressed in HBM (fs_set_compressed_orbit_mode 1): the
// HDRFloat<CudaDblflt> twin of SeqOrbit in kernels.hip (Perturb.cuh:160-326); same operations in the same order as
// k_decompress_hdr2x32, which expands the same waypoints once per upload.
struct Seq2x32 {
    const fs_orbit_2x32_rc *__restrict__ wp;
    uint32_t n_wp;
    HR cx, cy;
    uint32_t idx, next, next_index;
    HR zx, zy;
    __device__ __forceinline__ uint32_t index_of(uint32_t k) const { return (uint32_t)(wp[k].index_and_rebase & 0x7FFFFFFFFFFFFFFFull); }
    __device__ __forceinline__ void load(uint32_t k)
    {
        zx = HR{df32(wp[k].x_head, wp[k].x_tail), wp[k].ex};
        zy = HR{df32(wp[k].y_head, wp[k].y_tail), wp[k].ey};
    }
    __device__ __forceinline__ void step()
    {
        idx++;
        if (idx == next_index) {
            load(next);
            next++;
            next_index = next < n_wp ? index_of(next) : 0xFFFFFFFFu;
        } else {
            const HR zx_old = zx;
            zx = hr_add(hr_sub(hr_mul(zx, zx), hr_mul(zy, zy)), cx);
            hr_reduce(zx);
            zy = hr_add(hr_mul(hr_mul(hr2_from_float(2.0f), zx_old), zy), cy);
            hr_reduce(zy);
        }
    }
    __device__ __forceinline__ void seek(uint32_t i)
    {
        uint32_t lo = 0, hi = n_wp;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (index_of(mid) <= i)
                lo = mid;
            else
                hi = mid;
        }
        load(lo);
        idx = index_of(lo);
        next = lo + 1u;
        next_index = next < n_wp ? index_of(next) : 0xFFFFFFFFu;
        while (idx < i)
            step();
    }
};

// `T(X)` for an int, HDRFloat.h:293-363.
__device__ __forceinline__ HR hr2_from_int(int v) { return hr2_from_float((float)v); }

// compareToBothPositiveReducedTemplate<256>() < 0, HDRFloat.h:1169-1184
__device__ __forceinline__ bool below_bailout(HR n)
{
    if (n.e > 1)
        return false;
    if (n.e < 1)
        return true;
    return !(n.m >= df32(256.0f));
}


// The complex operations of the LA step with the real and the imaginary part side by side in packed registers (df32x2:
// the df32 operation sequences on both halves at once).  Same operations in the same order per part as hc_mul / hc_add /
// hc_mul2 of hdr_math.hpp with F = df32 (a double-float difference is the sum with the negated operand, df32_math.hpp).
__device__ __forceinline__ HC hc_mul_pk(HC a, HC b)
{
    const df32x2 P = df32x2(a.re, a.re) * df32x2(b.re, b.im); // (a.re b.re, a.re b.im)
    const df32x2 Q = df32x2(a.im, a.im) * df32x2(b.im, b.re); // (a.im b.im, a.im b.re)
    const df32x2 R = P + Q.neg_lo();                          // (a.re b.re - a.im b.im, a.re b.im + a.im b.re)
    return HC{R.lo(), R.hi(), clamp_exp(a.e + b.e)};
}
__device__ __forceinline__ HC hc_add_pk(HC a, HC b)
{
    const int32_t d = a.e - b.e;
    if (d >= kExpDiffIgnored) {
        return a;
    } else if (d >= 0) {
        const float mul = multiplier<df32>(-d).head; // {2^-d, 0}: mul_by_float is the full product for it (df32_math.hpp)
        const df32x2 R = df32x2(a.re, a.im) + mul_by_float(df32x2(b.re, b.im), (df32x2::f2){mul, mul});
        return HC{R.lo(), R.hi(), a.e};
    } else if (d > -kExpDiffIgnored) {
        const float mul = multiplier<df32>(d).head;
        const df32x2 R = mul_by_float(df32x2(a.re, a.im), (df32x2::f2){mul, mul}) + df32x2(b.re, b.im);
        return HC{R.lo(), R.hi(), b.e};
    }
    return b;
}
__device__ __forceinline__ HC hc_mul2_pk(HC a)
{
    const df32x2 R = mul_by_float(df32x2(a.re, a.im), (df32x2::f2){1.0f, 1.0f}); // a * {1, 0}
    return HC{R.lo(), R.hi(), clamp_exp(a.e + 1)};
}

// One perturbation step of the scalar-HDR loop (HDRFloat::custom_perturb3, HDRFloat.h:797-812, and the norms of
// LAKernel.cuh:170-200) with the X and the Y part side by side: its operations come in pairs of the same kind --
// (2zx + dX, 2zy + dY), (dX sumX, dX sumY), (dY sumY, dY sumX), (P - Q, P + Q), (+ cX, + cY), (z'x + nX, z'y + nY),
// (tX^2, nX^2) + (tY^2, nY^2) -- so every double-float product and sum is one packed operation for two (df32x2, hr_add2:
// the same IEEE operations per half as the scalar code the literal loop below runs, in the same order).  Returns false for
// a lane the straight line does not cover (hr_add2's `rare`, or a dz part that is an exact zero); the caller votes and runs
// the literal step for the wave then.
struct PkStep {
    HR nX, nY, tX, tY, norm, dnorm;
    bool covered;
};
__device__ __forceinline__ PkStep pt_step_pk(HR dX, HR dY, HR zx, HR zy, HR c0X, HR c0Y, HR zxn, HR zyn)
{
    // multiply_mutable clamps the sum of two exponents at kMinBigExp: that needs a factor that is an exact zero, and the only
    // factors of this step whose exponent is not bounded below by dz's own are dX and dY themselves.  (An exact zero in the
    // orbit -- entry 0, where every rebase lands -- or in delta-c only ever meets an addition, as the operand that is dropped.)
    bool rare = (dX.e < dY.e ? dX.e : dY.e) <= -(1 << 26);
    const hreal2 Z2(mul_by_float(df32x2(zx.m, zy.m), (df32x2::f2){1.0f, 1.0f}), zx.e + 1, zy.e + 1); // hr_mul2 of both (m * {1, 0})
    const hreal2 S = hr_add2<false>(Z2, hreal2(dX, dY), rare);                  // (sumX, sumY)
    const hreal2 P(df32x2(dX.m, dX.m) * S.m, dX.e + S.ex, dX.e + S.ey);           // (dX sumX, dX sumY)
    const hreal2 Q(df32x2(dY.m, dY.m) * S.m.swapped(), dY.e + S.ey, dY.e + S.ex); // (dY sumY, dY sumX)
    const hreal2 R = hr_add2<true>(P, Q, rare);                                 // (dX sumX - dY sumY, dX sumY + dY sumX)
    const hreal2 N = hr_add2<false>(R, hreal2(c0X, c0Y), rare);
    const HR nX = hr_reduced(N.x()), nY = hr_reduced(N.y());
    const hreal2 T = hr_add2<false>(hreal2(zxn, zyn), hreal2(nX, nY), rare);
    const df32x2 a(df32(T.m.head.x, T.m.tail.x), nX.m), b(df32(T.m.head.y, T.m.tail.y), nY.m);
    const hreal2 NN = hr_add2<false>(hreal2(a * a, T.ex * 2, nX.e * 2), hreal2(b * b, T.ey * 2, nY.e * 2), rare); // (|z|^2, |dz|^2)
    return PkStep{nX, nY, T.x(), T.y(), hr_reduced(NN.x()), hr_reduced(NN.y()), !rare};
}

// IterT: the reference's IterType for the counters (LAKernel.cuh:3): uint32_t, or uint64_t for iteration caps of 2^32 and
// above (iterations, the cap, the AT iteration count and the i x StepLength product in 64 bits).
// kSeq: the orbit stays compressed (A.wp): every entry comes from a Seq2x32 cursor.
template <int Mode, bool kStats, class IterT = uint32_t, bool kSeq = false>
__global__ void __launch_bounds__(256) k_lav2_2x32(FsLav2Args2x32 A)
{
    uint32_t X, L;
    if (A.pixel_order)
        ordered_pixel(A.frame, A.pixel_order, X, L);
    else if (A.tile_order) // (a first frame: tiles in the order of a sampled PerformAT count, kernels_tile_sample.hip)
        ordered_tile_pixel(A.tile_order, A.tiles_x, X, L);
    else
        tile_pixel(X, L);
    uint64_t c_at = 0, c_la = 0, c_pt = 0, c_px = 0, c_at_skipped = 0; // (c_at_skipped: AT iterations the cycle search spared this lane)
    uint32_t px_cost = 0;
    const bool in_buffer = X < A.frame.width && L < A.frame.local_rows;
    const uint32_t Y = in_buffer ? global_row(A.frame, L) : 0xFFFFFFFFu;
    const bool live = in_buffer && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        // LAKernel.cuh:39-63
        const HR DeltaSub0X = hr_sub(hr_mul(ldr(A.coords[0]), hr2_from_int((int)X)), ldr(A.coords[2]));
        const HR DeltaSub0Y = hr_sub(hr_mul(hr_neg(ldr(A.coords[1])), hr2_from_int((int)Y)), ldr(A.coords[3]));
        const HC DeltaSub0 = hc_from_hr(DeltaSub0X, DeltaSub0Y);
        HC DeltaSubN = hc_from_hr(hr2_from_int(0), hr2_from_int(0));
        IterT iter = 0;
        uint32_t RefIteration = 0;

        if (Mode != FS_MODE_PO) {
            // :66-71 + ATInfo::isValid / PerformAT, ATInfo.h:126-188
            if (A.la_valid && A.use_at && hr_cmp_pos(hc_cheb(DeltaSub0), ldr(A.at.ThresholdC)) <= 0) {
                const IterT ATMaxIt = n_iterations / A.at.StepLength;
                HC c = hc_add(hc_mul(DeltaSub0, ldc(A.at.CCoeff)), ldc(A.at.RefC));
                hc_reduce(c);
                HC z = hc_zero<df32>();
                const HR esc = ldr(A.at.SqrEscapeRadius);
                IterT i;
                for (i = 0; i < ATMaxIt; i++) {
                    // Steady state.  z starts as {0, 0, kMinBigExp}; the first z*z + c returns c itself (its exponent gap is
                    // beyond the 120 window), and from then on z carries c's exponent E whenever E <= 0 (|c| < 2): z*z has
                    // exponent 2E, the gap d = 2E - E = E selects the same arm of the complex addition in every iteration
                    // (E = 0: z*z + c * 2^0;  -120 < E < 0: (z*z) * 2^E + c, HDRFloatComplex.h plus_mutable), and the sum
                    // keeps exponent E.  The iteration is then plain double-float arithmetic on the mantissas -- the same
                    // operations in the same order as the general form below, minus its exponent bookkeeping and its
                    // three-way (divergent) addition; the constant c * 2^0 is computed once.
                    if (z.e == c.e && c.e <= 0 && c.e > -kExpDiffIgnored) {
                        const int32_t E = c.e;
                        const int32_t nsq_e = E << 1;
                        df32 re = z.re, im = z.im;
                        // Real and imaginary part side by side in packed registers (df32x2: the same operation sequences on
                        // both halves): (rr, re im) is one packed product, (ii, im re) another, (rr - ii, re im + im re) one
                        // packed sum with the first half's second operand negated, (+ c.re, + c.im) another -- 56 packed
                        // instructions and the 35 scalar ones of the norm test instead of 147 scalar ones per iteration.
                        df32x2 zz(re, im);
                        // The bailout test -- a double-float sum of the two squares, Reduce, a lexicographic compare: as many
                        // instructions as the rest of the iteration -- only decides "norm above the radius or not", and the
                        // norm is not used otherwise.  The float sum of the squares' heads is within 2^-21 (relative) of the
                        // double-float norm, so when it is below the radius by more than that (esc_low: the radius in the
                        // norm's scale, times 1 - 2^-16) the exact test would say "not above" and is skipped; otherwise
                        // (close to the radius, or not a number) the exact test runs and decides as before.
                        const int32_t esc_sh = esc.e - nsq_e;
                        const float esc_low = esc_sh > 120 ? __builtin_inff()
                                              : esc_sh < -120 ? 0.0f
                                                              : __builtin_amdgcn_ldexpf(esc.m.head, esc_sh) * 0.9999847412109375f;
                        // CYCLE SEARCH (round 5; at_math.hpp has the HDRFloat<double> form and the argument): the loop is a pure
                        // function of zz, so a state that comes back bit for bit -- looked for where i is a multiple of
                        // kAtCycleChunk, against a state kept at doubling distances -- means the lane will go round that
                        // cycle for good: it cannot escape, and its state after ATMaxIt iterations is the state
                        // (ATMaxIt - i) mod P iterations further on.  The lane's loop limit drops to exactly that many.
#ifndef FS_AT_CYCLE_CHUNK
#define FS_AT_CYCLE_CHUNK 8 /* measured on C4 as specified (ms per frame): 8: 198.8, 16: 200.0, 32: 202.4, 128: 219.2 */
#endif
                        constexpr uint32_t kAtCycleChunk = FS_AT_CYCLE_CHUNK;
                        IterT lim = ATMaxIt;      // this lane's loop limit: ATMaxIt, or where its remainder round the cycle ends
                        IterT s_it = 0, s_next = (IterT)kAtCycleChunk;
                        df32x2 s_zz(df32(__builtin_nanf(""), 0.0f), df32(0.0f, 0.0f)); // the kept state (a NaN equals nothing)
                        bool cyc = false, out = false;
#define FS_AT2_LOOP(UPDATE)                                                                                         \
    while (i < lim && !out) {                                                                                       \
        const IterT stop_ = (i | (IterT)(kAtCycleChunk - 1u)) + 1u;                                                 \
        const IterT end_ = stop_ < lim ? stop_ : lim;                                                               \
        for (; i < end_; i++) {                                                                                     \
            const df32x2 lhs = df32x2(zz.head.xx, zz.tail.xx) * zz;           /* (rr, re im) */                     \
            const df32x2 rhs = df32x2(zz.head.yy, zz.tail.yy) * zz.swapped(); /* (ii, im re) */                     \
            if (!(lhs.head.x + rhs.head.x < esc_low)) { /* (see esc_low) */                                         \
                HR nsq{lhs.lo() + rhs.lo(), nsq_e};                                                                 \
                hr_reduce(nsq);                                                                                     \
                if (hr_cmp_pos(nsq, esc) > 0) {                                                                     \
                    out = true;                                                                                     \
                    break;                                                                                          \
                }                                                                                                   \
            }                                                                                                       \
            zz = UPDATE;                                                                                            \
        }                                                                                                           \
        if (!out && !cyc && i == stop_ && i < lim) {                                                                \
            const bool same = __float_as_uint(zz.head.x) == __float_as_uint(s_zz.head.x) &&                         \
                              __float_as_uint(zz.head.y) == __float_as_uint(s_zz.head.y) &&                         \
                              __float_as_uint(zz.tail.x) == __float_as_uint(s_zz.tail.x) &&                         \
                              __float_as_uint(zz.tail.y) == __float_as_uint(s_zz.tail.y);                           \
            if (same) {                                                                                             \
                cyc = true;                                                                                         \
                lim = i + (ATMaxIt - i) % (i - s_it);                                                               \
            } else if (i >= s_next) {                                                                               \
                s_zz = zz, s_it = i, s_next = i + i;                                                                \
            }                                                                                                       \
        }                                                                                                           \
    }
                        if (E == 0) {
                            const df32 mul0 = multiplier<df32>(0);
                            const df32x2 cc(c.re * mul0, c.im * mul0);
                            // (re, re) * (re, im) and (im, im) * (im, re): the four products of z * z already paired the way the
                            // sums want them (same operand order per product as HDRFloatComplex's times), so no half of a result
                            // has to change registers;  ((rr - ii) + cre, (re im + im re) + cim)
                            FS_AT2_LOOP((lhs + rhs.neg_lo()) + cc)
                        } else {
                            const df32 mul = multiplier<df32>(E); // {2^E, 0} (-120 < E < 0)
                            const df32x2::f2 mm = {mul.head, mul.head};
                            const df32x2 cc(c.re, c.im);
                            FS_AT2_LOOP(mul_by_float(lhs + rhs.neg_lo(), mm) + cc) // (x * {2^E, 0}: df32_math.hpp)
                        }
#undef FS_AT2_LOOP
                        px_cost = (uint32_t)i; // (AT iterations this pixel ran: FsLav2Args2x32::pixel_cost)
                        if (cyc && !out) {
                            if (kStats)
                                c_at_skipped = (uint64_t)(ATMaxIt - i);
                            i = ATMaxIt; // (the state is the one ATMaxIt iterations produce: see above)
                        }
                        re = zz.lo(), im = zz.hi();
                        z = HC{re, im, E};
                        break;
                    }
                    HR nsq = hc_norm2(z);
                    hr_reduce(nsq);
                    if (hr_cmp_pos(nsq, esc) > 0)
                        break;
                    z = hc_add(hc_mul(z, z), c);
                }
                HC dz = hc_mul(z, ldc(A.at.InvZCoeff));
                hc_reduce(dz);
                DeltaSubN = dz;
                iter = i * A.at.StepLength;
                if (kStats)
                    c_at = i;
#ifdef FS_2X32_PROBE
                if (kStats) { // probe: lane slots of the AT loop = 64 x the longest lane of the wave (statistics word 13)
                    unsigned long long m = (unsigned long long)i;
                    for (int d = 32; d >= 1; d >>= 1) {
                        const unsigned long long o = __shfl_xor(m, d);
                        m = o > m ? o : m;
                    }
                    const uint64_t act = __builtin_amdgcn_ballot_w64(true);
                    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u)) == 0u)
                        atomicAdd((unsigned long long *)&A.stats[13], m * 64ull);
                }
#endif
            }
            // :73-131 (complex0's value before the stage loop is dead)
            uint32_t CurrentLAStage = A.la_valid ? A.stage_count : 0;
            const HR dcCheb = hc_cheb(DeltaSub0);
            while (CurrentLAStage > 0) {
                CurrentLAStage--;
                const uint32_t LAIndex = A.stages[CurrentLAStage].LAIndex;
                if (hr_cmp_pos(dcCheb, ldr(A.las[LAIndex].LAThresholdC)) >= 0) // GPU_LAReference.h:238-254
                    continue;
                const uint32_t MacroItCount = A.stages[CurrentLAStage].MacroItCount;
                uint32_t j = RefIteration;
                // (round 6, as k_lav2_hdr64) The Ref and the step length of record j are read ONE STEP AHEAD -- record j + 1's Ref is
                // read by step j anyway, for its rebase test, and the length decides first whether the step may be taken: a step that
                // waits for its own record's length before it asks for the coefficients makes two round trips to the cache.  Two steps
                // per trip with the two register sets' roles exchanged: nothing is copied.  -> true: the stage is left.
                auto la_step = [&](const HC &RJ, const uint32_t &lJ, HC &RN, uint32_t &lN) __attribute__((always_inline)) -> bool {
                    const fs_la_2x32_u32 *LAj = &A.las[LAIndex + j]; // getLA, GPU_LAReference.h:271-303
                    const uint32_t l = lJ;
                    const uint32_t next_stage = LAj->NextStageLAIndex;
                    const HC ZCoeff = ldc(LAj->ZCoeff), CCoeff = ldc(LAj->CCoeff);
                    const HR thr = ldr(LAj->LAThreshold);
                    RN = ldc(LAj[1].Ref);
                    lN = LAj[1].StepLength;
                    if (iter + l > n_iterations) { // the step would pass the iteration limit: unusable
                        RefIteration = next_stage;
                        return true;
                    }
                    // Prepare, GPU_LAInfoDeep.h:90-106
                    HC newDz = hc_mul_pk(DeltaSubN, hc_add_pk(hc_mul2_pk(RJ), DeltaSubN));
                    hc_reduce(newDz);
                    if (hr_cmp_pos(hc_cheb(newDz), thr) >= 0) {
                        RefIteration = next_stage;
                        return true;
                    }
                    iter += l;
                    if (kStats)
                        c_la++;
                    // Evaluate GPU_LAInfoDeep.h:120-124, getZ LAstep.h:181-185
                    // (round 6: dc CCoeff 120 binades and more below newDz ZCoeff in every lane of the wave -- nine LA steps of ten at
                    // C4's zoom, tools/c4_arm_probe.py -- plus_mutable returns its first operand: the second product is not formed)
                    if (__builtin_amdgcn_ballot_w64(clamp_exp(newDz.e + ZCoeff.e) - clamp_exp(DeltaSub0.e + CCoeff.e) >= kExpDiffIgnored) ==
                        __builtin_amdgcn_ballot_w64(true))
                        DeltaSubN = hc_mul_pk(newDz, ZCoeff);
                    else
                        DeltaSubN = hc_add_pk(hc_mul_pk(newDz, ZCoeff), hc_mul_pk(DeltaSub0, CCoeff));
                    const HC complex0 = hc_add_pk(RN, DeltaSubN);
                    j++;
                    const HR lhs = hr_reduced(hc_cheb(complex0));
                    const HR rhs = hr_reduced(hc_cheb(DeltaSubN));
                    if (hr_cmp_pos(lhs, rhs) < 0 || j >= MacroItCount) {
                        DeltaSubN = complex0;
                        j = 0;
                        RN = ldc(A.las[LAIndex].Ref);
                        lN = A.las[LAIndex].StepLength;
                    }
                    return false;
                };
                HC RefA = hc_zero<df32>(), RefB = hc_zero<df32>();
                uint32_t lA = 0, lB = 0;
                if (iter < n_iterations) {
                    RefA = ldc(A.las[LAIndex + j].Ref);
                    lA = A.las[LAIndex + j].StepLength;
                }
                while (iter < n_iterations) {
                    if (la_step(RefA, lA, RefB, lB))
                        break;
                    if (!(iter < n_iterations))
                        break;
                    if (la_step(RefB, lB, RefA, lA))
                        break;
                }
                if (iter >= n_iterations)
                    break;
            }
        }

        const IterT it_la = iter; // (the perturbation steps of this pixel = its final count - this)
        if (Mode != FS_MODE_LAO) {
            // :133-235.  perturbLoop(maxRefIteration) at :254-276 reads the block's previous results, which are zero on
            // the cleared buffer every caller passes (Fractal.cpp:2822), so only perturbLoop(n_iterations) runs.
            const fs_orbit_2x32 *__restrict__ orb = A.orbit;
            const uint32_t MaxRef = A.orbit_count - 1;
            HR dX = hc_re(DeltaSubN), dY = hc_im(DeltaSubN);
            Seq2x32 seq;
            HR zx, zy;
            if constexpr (kSeq) {
                seq.wp = A.wp;
                seq.n_wp = A.n_wp;
                seq.cx = ldr(A.cxLow), seq.cy = ldr(A.cyLow);
                seq.seek(RefIteration);
                zx = seq.zx, zy = seq.zy;
            } else {
                zx = orbit_x(orb, RefIteration), zy = orbit_y(orb, RefIteration);
            }
            for (;;) {
                HR tX, tY, normSquared, dnorm = HR{df32(0.0f), 0};
                bool fast = false;
                if constexpr (!kSeq) {
                    // the packed straight-line step (pt_step_pk above); taken when every running lane of the wave is covered
                    const HR zxn = orbit_x(orb, RefIteration + 1), zyn = orbit_y(orb, RefIteration + 1);
                    const PkStep st = pt_step_pk(dX, dY, zx, zy, DeltaSub0X, DeltaSub0Y, zxn, zyn);
                    if (__builtin_amdgcn_ballot_w64(!st.covered) == 0ull) {
                        fast = true;
                        ++RefIteration;
                        dX = st.nX, dY = st.nY;
                        zx = zxn, zy = zyn;
                        tX = st.tX, tY = st.tY, normSquared = st.norm, dnorm = st.dnorm;
                        if (kStats)
                            c_pt++;
                    }
                }
                if (!fast) {
#ifdef FS_2X32_PROBE
                    if (kStats)
                        atomicAdd((unsigned long long *)&A.stats[12], 1ull); // probe: lane-steps through the literal step
#endif
                    const HR sumY = hr_add(hr_mul2(zy), dY); // tempSum1
                    const HR sumX = hr_add(hr_mul2(zx), dX); // tempSum2
                    ++RefIteration;
                    // custom_perturb3 (its tempSum1 parameter is bound to tempSum2 and vice versa)
                    HR nX = hr_add(hr_sub(hr_mul(dX, sumX), hr_mul(dY, sumY)), DeltaSub0X);
                    hr_reduce(nX);
                    HR nY = hr_add(hr_add(hr_mul(dX, sumY), hr_mul(dY, sumX)), DeltaSub0Y);
                    hr_reduce(nY);
                    dX = nX;
                    dY = nY;
                    if (kStats)
                        c_pt++;
                    if constexpr (kSeq) {
                        seq.step();
                        zx = seq.zx, zy = seq.zy;
                    } else {
                        zx = orbit_x(orb, RefIteration);
                        zy = orbit_y(orb, RefIteration);
                    }
                    tX = hr_add(zx, dX);
                    tY = hr_add(zy, dY);
                    normSquared = hr_reduced(hr_add(hr_square(tX), hr_square(tY)));
                }
                if (below_bailout(normSquared) && iter < n_iterations) {
                    const HR DeltaNormSquared = fast ? dnorm : hr_reduced(hr_add(hr_square(dX), hr_square(dY)));
                    if (hr_cmp_pos(normSquared, DeltaNormSquared) < 0 || RefIteration >= MaxRef) {
                        dX = tX;
                        dY = tY;
                        RefIteration = 0;
                        if constexpr (kSeq) {
                            seq.seek(0); // a new SeqWorkspace at the start of the orbit
                            zx = seq.zx, zy = seq.zy;
                        } else {
                            zx = orbit_x(orb, 0);
                            zy = orbit_y(orb, 0);
                        }
                    }
                    ++iter;
                } else {
                    break;
                }
            }
        }
        store_iter(A.out, A.frame, L, X, iter);
        if (A.pixel_cost) {
            const uint64_t pt = (uint64_t)(iter - it_la); // (AT iterations first, perturbation steps second: see k_lav2_lit)
            A.pixel_cost[(size_t)L * A.frame.rounded_width + X] =
                ((px_cost > 0xFFFFFu ? 0xFFFFFu : px_cost) << 12) | (pt > 0xFFFull ? 0xFFFu : (uint32_t)pt);
        }
    }
    if (kStats) {
        add_stats(A.stats, c_at, c_la, c_pt, c_px);
        // statistics word 5: AT iterations executed (= counted - spared by the cycle search)
        uint64_t e = c_at - c_at_skipped;
        for (int off = 32; off > 0; off >>= 1)
            e += __shfl_down(e, off);
        if ((threadIdx.x & 63) == 0)
            atomicAdd((unsigned long long *)&A.stats[5], (unsigned long long)e);
    }
}

} // namespace

void fsk_lav2_2x32(const FsLav2Args2x32 &A, int mode, bool stats, hipStream_t s)
{
    const dim3 b(256);
    const dim3 g((A.frame.width + 31) / 32, (A.frame.local_rows + 7) / 8); // tile_pixel()
#define FS_LAUNCH(M)                                                                                                    \
    do {                                                                                                                \
        if (A.wp != nullptr && A.frame.wide != 0u)                                                                      \
            hipLaunchKernelGGL((k_lav2_2x32<M, false, uint64_t, true>), g, b, 0, s, A);                                 \
        else if (A.wp != nullptr)                                                                                       \
            hipLaunchKernelGGL((k_lav2_2x32<M, false, uint32_t, true>), g, b, 0, s, A);                                 \
        else if (A.frame.wide != 0u)                                                                                    \
            hipLaunchKernelGGL((k_lav2_2x32<M, false, uint64_t>), g, b, 0, s, A);                                       \
        else if (stats)                                                                                                 \
            hipLaunchKernelGGL((k_lav2_2x32<M, true>), g, b, 0, s, A);                                                  \
        else                                                                                                            \
            hipLaunchKernelGGL((k_lav2_2x32<M, false>), g, b, 0, s, A);                                                 \
    } while (0)
    if (mode == FS_MODE_PO)
        FS_LAUNCH(FS_MODE_PO);
    else if (mode == FS_MODE_LAO)
        FS_LAUNCH(FS_MODE_LAO);
    else
        FS_LAUNCH(FS_MODE_FULL);
#undef FS_LAUNCH
}
