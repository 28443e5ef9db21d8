// kernels.hip -- gfx950 device code of libfsmi355.so.  Compiled with -ffp-contract=off (see hdr_math.hpp).
//
// Parity target of every iteration kernel is a reference *CPU* RenderAlgorithm function (cited per kernel);
// the decomposition is this project's own: one lane per (sub)pixel, 64-wide wavefronts walking a row
// segment, reference orbit pre-converted once per upload into the complex form the CPU loop rebuilds on
// every access.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"
#include "at_math.hpp"
#include "kernels.h"
#include <cstdlib>
#include "kernel_common.hpp"
#include "lav2_common.hpp"

using namespace fs;

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

// ------------------------------------------------------------------------------------------------
// Orbit preparation: fs_orbit_hdr32 (reference layout) -> {re, im, exp, 0}.
__global__ void k_prepare_orbit_hdr32(const fs_orbit_hdr32 *__restrict__ in, float4 *__restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const fs_orbit_hdr32 e = in[i];
    const hcplx32 c = hc_from_hr(hreal32{e.mx, e.ex}, hreal32{e.my, e.ey});
    // .w = 2^(8 - 2*exp): with z = Z + dz aligned to Z's exponent, |z|^2 > 256  <=>  re^2 + im^2 > .w  (tuned loop).
    // Overflows to +inf for a tiny Z (never "escaped" there), cannot underflow (orbit values are < 2^9).
    out[i] = make_float4(c.re, c.im, __int_as_float(c.e), ldexpf(1.0f, 8 - 2 * (c.e < -1000 ? -1000 : c.e)));
}

// Bound of a scaled-run arrival at orbit entry v = {re, im, exp}: 2^-2 * max(|Z.re|, |Z.im|) in true scale, or -0.0 (bit
// pattern INT_MIN = "never": the loops compare bit patterns as integers) unless 2^-40 <= max part < 5.6 (|Z| < 8) and the
// smaller part is within 2^40 of the larger one.
__device__ __forceinline__ float scaled_bound(const float4 v)
{
    const int e = __float_as_int(v.z);
    const float hi = fs_max_abs(v.x, v.y);
    const float lo = fs_min_abs(v.x, v.y);
    const float zmax = __builtin_amdgcn_ldexpf(hi, e < -200 ? -200 : (e > 100 ? 100 : e));
    const bool usable = zmax >= 0x1p-40f && zmax < 5.6f && lo >= hi * 0x1p-40f;
    return usable ? zmax * 0x1p-2f : -0.0f;
}

// A scaled run may start at entry e of the second companion array: a usable entry (bound not "never") or an exact zero.
__device__ __forceinline__ bool scaled_startable(const float4 e)
{
    return __float_as_int(e.z) != (int)0x80000000 || (e.x == 0.0f && e.y == 0.0f);
}

// Companion array of the tuned LAv2 loop: {re, im, s, -} with s = ~exp + 116 (-(s - 116) = exp + 1 = the exponent of 2Z;
// the bias turns the loop's range tests into comparisons against constants) for orbit values below 8, and a large
// positive poison for larger ones, which makes the range test fail there.  zq must hold 2 n entries.
// Block bound of entry j (the .w of the second companion, see below; "never" when one of the four entries after j has no
// bound).  With G = max(max|dz|, max|dc|) at entry j (maximum norms of the parts, true scale) the block's four arrivals pass
// their own bound tests whenever G <= this value:
//   one step from entry k:  dz' = dz (2 Z_k + dz) + dc,  |a b|_inf <= 2 |a|_inf |b|_inf for complex a, b, so
//   |dz'|_inf <= 2 G (2 M_k + G) + G <= G (4 M_k + 3.8)        M_k = max part of Z_k,  G <= 1.4 (every bound is <= 0.25 * 5.6)
//   => G grows by at most g_k = 4 M_k + 3.8 per step (dc included: g_k > 1), and arrival j + m passes its test when
//      G g_j g_(j+1) .. g_(j+m-1) <= bound[j + m]:   block bound = min over m = 1 .. 4 of bound[j + m] / (g_j .. g_(j+m-1)),
//   by induction over the block's steps (arrival j + m - 1 has passed its test, so G <= 1.4 holds where step m starts; at
//   entry j itself G <= bound[j + 1] / 3.8).  Rounding: the steps' own rounding errors are 2^-23 relative, each g carries
//   a factor 1 + 2^-10.  (Round 3 used one constant for every step -- |2Z + dz| < 20, i.e. 2^-18 for four steps; the orbit's
//   own |Z| gives 2^-9 .. 2^-12 for typical entries, and fewer blocks need their bound tests.)
__device__ __forceinline__ float scaled_block_bound(const float4 *__restrict__ zref, uint64_t j, uint64_t n)
{
    if (j >= n)
        return -0.0f;
    float best = 0x1p60f, grow = 1.0f;
    bool ok = true;
    for (uint32_t k = 0; k < 4; k++) {
        // growth of the step that leaves entry j + k
        const float4 v = zref[j + k < n ? j + k : n - 1];
        const int e = __float_as_int(v.z);
        const float m = __builtin_amdgcn_ldexpf(fs_max_abs(v.x, v.y),
                                               e < -200 ? -200 : (e > 100 ? 100 : e));
        ok = ok && m < 5.6f; // (a state never sits at an entry this large: its own bound is "never")
        grow *= (4.0f * m + 3.8f) * (1.0f + 0x1p-10f);
        const float bk = j + k + 1 < n ? scaled_bound(zref[j + k + 1]) : -0.0f;
        ok = ok && __float_as_int(bk) != (int)0x80000000;
        best = __builtin_fminf(best, bk / grow);
    }
    return ok ? best * (1.0f - 0x1p-10f) : -0.0f;
}

__global__ void k_make_quiet_orbit(const float4 *__restrict__ zref, float4 *__restrict__ zq, float2 *__restrict__ zs2,
                                   float4 *__restrict__ zqb, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float4 v = zref[i];
    const int e = __float_as_int(v.z);
    zq[i] = make_float4(v.x, v.y, __int_as_float(e <= 2 ? ~e + 116 : (1 << 24)), 0.0f);
    // second companion, zq[n + i], for the scaled runs: {2Z.re, 2Z.im, scaled_bound(Z), block bound} as plain floats (true
    // scale).  (Packing the entry into 8 bytes and deriving the bound from 2Z costs one more vector
    // instruction per step and was measured slower: the loop is not bound by its loads.)
    const float b0 = scaled_bound(v);
    // .w: the BLOCK bound of the four entries after this one (scaled_block_bound above): G = max(max|dz|, max|dc|) <= .w at
    // this entry implies that each of the four arrivals passes its own bound test, and the scalar-cache path of the scaled
    // runs then skips those tests (same accepted steps).  (Eight entries per test -- one test per loop body -- passed for
    // 67 % of C3's blocks instead of 93.6 % with round 3's constant-growth bound: 63.5 instead of 55.8 ms.)
    // (A run may START at every usable entry and at an exact zero -- entry 0, where every rebase lands; 2Z + dz is then dz
    // itself in either arithmetic -- the kernels read that off .z and .xy.  Nothing arrives at a zero entry: its bound is
    // the "never" pattern.)
    zq[n + i] = make_float4(__builtin_amdgcn_ldexpf(v.x, e + 1), __builtin_amdgcn_ldexpf(v.y, e + 1), b0,
                            scaled_block_bound(zref, i, n));
    // the compact form for the 16-step body of the untested loop (FS_FAST_LOOP_FD16): 2Z alone, and the block bounds a body
    // whose first arrival is entry i needs -- those of its entries 3, 7, 11 (the states its second to fourth blocks start
    // from) and 15 (the state the NEXT body starts from)
    zs2[i] = make_float2(__builtin_amdgcn_ldexpf(v.x, e + 1), __builtin_amdgcn_ldexpf(v.y, e + 1));
    zqb[i] = make_float4(scaled_block_bound(zref, i + 3, n), scaled_block_bound(zref, i + 7, n),
                         scaled_block_bound(zref, i + 11, n), scaled_block_bound(zref, i + 15, n));
}

// ------------------------------------------------------------------------------------------------
// Sequential access to a PerturbExtras::SimpleCompression orbit that stays compressed in HBM: the device twin of
// GPUPerturbSingleResults::SeqWorkspace / GetIterSeq / BinarySearch (Perturb.cuh:160-326) and of the CPU
// RuntimeDecompressor (PerturbationResultsHelpers.h:35-199).  A cursor holds one orbit value; seek() starts from the
// last waypoint at or before the index and iterates z = z^2 + c forward in T arithmetic (c = OrbitXLow / OrbitYLow),
// step() moves one index on: the next waypoint when its index comes up, one iteration otherwise.  Same operations in
// the same order as k_decompress_orbit_* (which expand the same waypoints once per upload), so both modes see the same
// orbit bit for bit; this one trades ~60 vector instructions per step for an orbit that occupies only its waypoints.
namespace {

template <class F> struct FsWaypoints;
template <> struct FsWaypoints<float> {
    using Rec = fs_orbit_hdr32_rc;
};
template <> struct FsWaypoints<double> {
    using Rec = fs_orbit_hdr64_rc;
};

// PosT = the reference's IterType for orbit POSITIONS: uint32_t, or uint64_t in the instantiation that counts in 64 bits --
// a compressed orbit can have 2^32 and more uncompressed entries (its waypoints are what is resident), and waypoint
// indices, the cursor, RefIteration and the period are then 64-bit like the reference's (Perturb.cuh:21-23,202-203,247-271).
template <class F, class PosT = uint32_t> struct SeqOrbit {
    const typename FsWaypoints<F>::Rec *__restrict__ wp;
    uint32_t n_wp;
    hreal<F> cx, cy;
    // cursor
    PosT idx;        // orbit index of (zx, zy)
    uint32_t next;   // number of the first waypoint behind the cursor
    PosT next_index; // ... and its orbit index (all ones: none left)
    hreal<F> zx, zy;

    __device__ __forceinline__ PosT index_of(uint32_t k) const
    {
        return (PosT)(wp[k].index_and_rebase & 0x7FFFFFFFFFFFFFFFull);
    }
    __device__ __forceinline__ void load(uint32_t k)
    {
        zx = hreal<F>{wp[k].mx, wp[k].ex};
        zy = hreal<F>{wp[k].my, wp[k].ey};
    }
    __device__ __forceinline__ void step()
    {
        idx++;
        if (idx == next_index) { // GetCompressedComplexSeq, Perturb.cuh:303-326
            load(next);
            next++;
            next_index = next < n_wp ? index_of(next) : ~(PosT)0;
        } else { // runOneIter, PerturbationResultsHelpers.h:51-58
            const hreal<F> zx_old = zx;
            zx = hr_add(hr_sub(hr_mul(zx, zx), hr_mul(zy, zy)), cx);
            hr_reduce(zx);
            zy = hr_add(hr_mul(hr_mul(hreal<F>{F(1), 1}, zx_old), zy), cy);
            hr_reduce(zy);
        }
    }
    __device__ __forceinline__ void seek(PosT i)
    {
        // BinarySearch, Perturb.cuh:241-263: the last waypoint whose index is <= i (waypoint 0 sits at index 0)
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
        next_index = next < n_wp ? index_of(next) : ~(PosT)0;
        while (idx < i)
            step();
    }
    // PerturbationResults::GetComplex on the cursor's value
    __device__ __forceinline__ hcplx<F> value() const { return hc_from_hr(zx, zy); }
};

// test hook (fs_seq_cursor_probe): one lane seeks to `start` and walks n entries on; out[k] = the value at start + k
template <class F, class PosT>
__global__ void k_seq_cursor_probe(const void *wp, uint32_t n_wp, hreal<F> cx, hreal<F> cy, uint64_t start, uint32_t n,
                                   hcplx<F> *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    SeqOrbit<F, PosT> seq;
    seq.wp = (const typename FsWaypoints<F>::Rec *)wp;
    seq.n_wp = n_wp;
    seq.cx = cx, seq.cy = cy;
    seq.seek((PosT)start);
    for (uint32_t k = 0; k < n; k++) {
        out[k] = seq.value();
        seq.step();
    }
}

} // namespace

// ------------------------------------------------------------------------------------------------
// LAv2, T = HDRFloat<float>.  CPU twin: Fractal::CalcCpuPerturbationFractalLAV2<uint32_t,float,Disable>
// (Fractal.cpp:2545-2678) with LAReference::getLA / isLAStageInvalid (LAReference.cpp:1076-1134),
// LAInfoDeep::Prepare / Evaluate (LAInfoDeep.h:395-420), ATInfo::PerformAT (ATInfo.h:155-188).
// Replaces mandel_1xHDR_float_perturb_lav2 (FractalSharkGpuLib/LAKernel.cuh:3-315).
// IterT = the reference's IterType for the COUNTERS (LAKernel.cuh:3): uint32_t, or uint64_t for iteration caps of 2^32 and
// above (iterations, the cap, the AT iteration count and the skipped-iteration product are then 64-bit; table step
// lengths and indices stay 32-bit -- an orbit or table with 2^32 entries would not fit any device).
// kSeq: the orbit is a compressed one that stays compressed (A.wp): every orbit value comes from a SeqOrbit cursor.
template <class F, int Mode, bool kStats, class IterT = uint32_t, bool kSeq = false>
__global__ void __launch_bounds__(256) k_lav2_lit(FsLav2ArgsT<F> A)
{
    // The float instantiation is the operation-by-operation A/B reference of the tuned kernel and keeps the literal AT
    // loop; the double instantiation is the production HDRFloat<double> kernel and uses the steady-state AT loop.
    constexpr bool kFastAT = sizeof(F) == 8;
    // 64-bit POSITIONS go with the waypoint-resident orbit and 64-bit counters (see SeqOrbit); every other instantiation
    // keeps 32-bit ones (an expanded orbit of 2^32 entries does not fit a device)
    constexpr bool kWidePos = kSeq && sizeof(IterT) == 8;
    using PosT = std::conditional_t<kWidePos, uint64_t, uint32_t>;
    using LaRec = typename FsDev<F>::LA;
    using LaRec64 = typename FsLaU64<F>::T;
    // the table's records: the narrowed (uint32_t) ones, or the reference's uint64_t records as they are (A.la_u64) -- the
    // two layouts agree up to StepLength, so the coefficients are read through the narrow type either way
    const bool la64 = kWidePos && A.la_u64 != 0u;
    auto la_rec = [&](uint64_t idx) -> const LaRec * {
        return la64 ? (const LaRec *)((const LaRec64 *)(const void *)A.las + idx) : A.las + idx;
    };
    auto la_step_length = [&](const LaRec *p) -> IterT {
        return la64 ? (IterT)((const LaRec64 *)(const void *)p)->StepLength : (IterT)p->StepLength;
    };
    auto la_next_stage = [&](const LaRec *p) -> PosT {
        return la64 ? (PosT)((const LaRec64 *)(const void *)p)->NextStageLAIndex : (PosT)p->NextStageLAIndex;
    };
    uint32_t X, L;
    if (A.pixel_order)
        ordered_pixel(A.frame, A.pixel_order, X, L);
    else
        tile_pixel(X, L);
    uint64_t c_at = 0, c_la = 0, c_pt = 0, c_px = 0, c_at_exec = 0; // (c_at_exec: AT iterations actually run -- the cycle search of at_perform spares the rest)
    uint32_t px_cost = 0; // AT iterations this pixel ran (FsLav2ArgsT::pixel_cost)
    uint64_t c_at_own = 0; // AT iterations this pixel needs by itself (what the AT pass of its own runs for it: statistics word 6)
    const bool in_buffer = X < A.frame.width && L < A.frame.local_rows;
    const uint32_t Y = in_buffer ? global_row(A.frame, L) : 0xFFFFFFFFu;
    const bool live = in_buffer && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const IterT n_iterations = sizeof(IterT) == 8 ? (IterT)(((uint64_t)A.n_iterations_hi << 32) | A.n_iterations)
                                                      : (IterT)A.n_iterations;
        hreal<F> deltaReal, deltaImaginary;
        pixel_delta<F>(A.coords, X, Y, deltaReal, deltaImaginary);
        const hcplx<F> DeltaSub0 = hc_from_hr(deltaReal, deltaImaginary);
        hcplx<F> DeltaSubN = hc_from_native<F>(F(0), F(0)); // {0,0}: zero with exponent 0 (Fractal.cpp:2565)
        IterT iterations = 0;

        bool at_done = false;
        if constexpr (Mode != FS_MODE_PO && std::is_same<F, double>::value && sizeof(IterT) == 4 && !kSeq) {
            if (A.at_res) { // PerformAT ran in its own pass (fsk_at_pass64): its result instead of the iteration
                const FsAtRes ar = A.at_res[(size_t)L * A.frame.rounded_width + X];
                at_done = true;
                if (ar.i != 0xFFFFFFFFu) {
                    DeltaSubN = hcplx<F>{ar.re, ar.im, ar.e};
                    iterations = (IterT)ar.i * (IterT)A.at.StepLength;
                }
            }
        }
        if (Mode != FS_MODE_PO && !at_done) {
            if (A.la_valid && A.use_at && hr_cmp_pos(hc_cheb(DeltaSub0), ldr(A.at.ThresholdC)) <= 0) {
                const IterT at_step = kWidePos ? (IterT)(((uint64_t)A.at_step_hi << 32) | A.at.StepLength) : (IterT)A.at.StepLength;
                const IterT ATMaxIt = n_iterations / at_step;
                hcplx<F> c = hc_add(hc_mul(DeltaSub0, ldc(A.at.CCoeff)), ldc(A.at.RefC));
                hc_reduce(c);
                hcplx<F> z;
                IterT i, i_exec = 0, i_own = 0;
                if (kFastAT) {
                    at_perform<F, IterT>(c, ldr(A.at.SqrEscapeRadius), ATMaxIt, z, i, &i_exec, &i_own);
                    px_cost = i_own > (IterT)0xFFFFFu ? 0xFFFFFu : (uint32_t)i_own;
                } else {
                    z = hc_zero<F>();
                    const hreal<F> esc = ldr(A.at.SqrEscapeRadius);
                    for (i = 0; i < ATMaxIt; i++) {
                        hreal<F> nsq = hc_norm2(z);
                        hr_reduce(nsq);
                        if (hr_cmp_pos(nsq, esc) > 0)
                            break;
                        z = hc_add(hc_mul(z, z), c);
                    }
                }
                hcplx<F> dz = hc_mul(z, ldc(A.at.InvZCoeff));
                hc_reduce(dz);
                DeltaSubN = dz;
                iterations = i * at_step;
                if (kStats) {
                    c_at = i;
                    c_at_exec = kFastAT ? (uint64_t)i_exec : (uint64_t)i;
                    c_at_own = kFastAT ? (uint64_t)i_own : (uint64_t)i;
                }
            }
        }

        PosT RefIteration = 0;
        const PosT MaxRefIteration = (kWidePos ? (PosT)(((uint64_t)A.orbit_count_hi << 32) | A.orbit_count) : (PosT)A.orbit_count) - 1;
        const PosT period = kWidePos ? (PosT)(((uint64_t)A.period_hi << 32) | A.period) : (PosT)A.period;
        // complex0 before the LA stages is dead in the CPU function (only its norm was read, into a variable
        // that is overwritten before use), so it is not materialised; the RefIteration %= period side effect
        // (Fractal.cpp:2590-2591) is kept.
        if (iterations != 0 && !(RefIteration < MaxRefIteration) && period != 0)
            RefIteration = RefIteration % period;

        if (Mode != FS_MODE_PO) {
            uint32_t CurrentLAStage = A.la_valid ? A.stage_count : 0;
            const hreal<F> dcCheb = hc_cheb(DeltaSub0);
            while (CurrentLAStage > 0) {
                CurrentLAStage--;
                const uint32_t LAIndex = A.stages[CurrentLAStage].LAIndex;
                {
                    const int cmp = hr_cmp_pos(dcCheb, ldr(la_rec(LAIndex)->LAThresholdC));
                    const bool invalid = A.parity == FS_PARITY_LITERAL ? (cmp < 0) : (cmp >= 0);
                    if (invalid)
                        continue;
                }
                const uint32_t MacroItCount = A.stages[CurrentLAStage].MacroItCount;
                PosT j = RefIteration;
                // (round 4) the Ref of record j + 1, read for the rebase test of step j, is the Ref step j + 1 starts from:
                // one load of it per step, not two, unless the test reset j
                hcplx<F> RefJ = hc_zero<F>();
                if (iterations < n_iterations)
                    RefJ = ldc(la_rec((uint64_t)LAIndex + j)->Ref);
                while (iterations < n_iterations) {
                    const LaRec *LAj = la_rec((uint64_t)LAIndex + j);
                    const IterT l = la_step_length(LAj);
                    bool unusable = true;
                    hcplx<F> newDz = hc_zero<F>();
                    if (iterations + l <= n_iterations) {
                        newDz = hc_mul(DeltaSubN, hc_add(hc_mul2(RefJ), DeltaSubN));
                        hc_reduce(newDz);
                        unusable = hr_cmp_pos(hc_cheb(newDz), ldr(LAj->LAThreshold)) >= 0;
                    }
                    if (unusable) {
                        RefIteration = la_next_stage(LAj);
                        break;
                    }
                    iterations += l;
                    if (kStats)
                        c_la++;
                    DeltaSubN = hc_add(hc_mul(newDz, ldc(LAj->ZCoeff)), hc_mul(DeltaSub0, ldc(LAj->CCoeff)));
                    const hcplx<F> RefN = ldc(la_rec((uint64_t)LAIndex + j + 1)->Ref);
                    const hcplx<F> complex0 = hc_add(RefN, DeltaSubN);
                    j++;
                    const hreal<F> lhs = hr_reduced(hc_cheb(complex0));
                    const hreal<F> rhs = hr_reduced(hc_cheb(DeltaSubN));
                    if (hr_cmp_pos(lhs, rhs) < 0 || j >= MacroItCount) {
                        DeltaSubN = complex0;
                        j = 0;
                        RefJ = ldc(la_rec((uint64_t)LAIndex)->Ref);
                    } else {
                        RefJ = RefN;
                    }
                }
                if (iterations >= n_iterations)
                    break;
            }
        }

        const IterT it_la = iterations; // (the perturbation steps of this pixel = its final count - this)
        if (Mode != FS_MODE_LAO) {
            const hreal<F> TwoFiftySix = hreal<F>{F(1), 8};
            const typename FsDev<F>::Z *__restrict__ zr = A.zref;
            SeqOrbit<F, PosT> seq;
            if constexpr (kSeq) {
                seq.wp = (const typename FsWaypoints<F>::Rec *)A.wp;
                seq.n_wp = A.n_wp;
                seq.cx = ldr(A.cxLow), seq.cy = ldr(A.cyLow);
                if (iterations < n_iterations)
                    seq.seek(RefIteration); // SeqWorkspace(results, RefIteration)
            }
            // (round 4) Two things the reference's loop does per step are not done per step here, with the same results:
            // the orbit entry a step arrives at is the entry the next step leaves from -- one load per step, not two, unless
            // a rebase moved the index; and Reduce(z) before |z|^2 only re-labels z -- both parts scaled by the same power of
            // two, which commutes with the squares and their sum (the larger part of z is an O(1) mantissa: nothing comes
            // near the ends of the range) -- so the reduced norm is the same and z is reduced only where it is stored, on a
            // rebase.
            hcplx<F> Zhere = hc_zero<F>();
            if constexpr (!kSeq) {
                if (iterations < n_iterations)
                    Zhere = zref_at(zr, (uint32_t)RefIteration);
            }
            for (; iterations < n_iterations; iterations++) {
                hcplx<F> cur;
                if constexpr (kSeq)
                    cur = seq.value();
                else
                    cur = Zhere;
                cur = hc_mul2(cur);
                cur = hc_add(cur, DeltaSubN);
                DeltaSubN = hc_mul(DeltaSubN, cur);
                DeltaSubN = hc_add(DeltaSubN, DeltaSub0);
                hc_reduce(DeltaSubN);
                if (kStats)
                    c_pt++;
                RefIteration++;
                hcplx<F> Znext;
                if constexpr (kSeq) {
                    seq.step(); // GetIterSeq
                    Znext = seq.value();
                } else {
                    Znext = zref_at(zr, (uint32_t)RefIteration);
                    Zhere = Znext;
                }
                hcplx<F> complex0 = hc_add(Znext, DeltaSubN);
                const hreal<F> normSquared = hr_reduced(hc_norm2(complex0));
                const hreal<F> DeltaNormSquared = hr_reduced(hc_norm2(DeltaSubN));
                if (hr_cmp_pos(normSquared, TwoFiftySix) > 0)
                    break;
                if (hr_cmp_pos(normSquared, DeltaNormSquared) < 0 || RefIteration >= MaxRefIteration) {
                    hc_reduce(complex0);
                    DeltaSubN = complex0;
                    RefIteration = 0;
                    if constexpr (kSeq)
                        seq.seek(0); // a new SeqWorkspace at the start of the orbit
                    else
                        Zhere = zref_at(zr, 0u);
                }
            }
        }
        store_iter(A.out, A.frame, L, X, iterations);
        if (A.pixel_cost) {
            // AT iterations first, perturbation steps second: the two phases run one after the other and a wave pays the
            // longest lane of each, so pixels should agree in both (a single sum puts a pixel that iterates long and steps
            // little next to one that does the opposite)
            const uint64_t pt = (uint64_t)(iterations - it_la);
            A.pixel_cost[(size_t)L * A.frame.rounded_width + X] = (px_cost << 12) | (pt > 0xFFFull ? 0xFFFu : (uint32_t)pt);
        }
    }
    if (kStats) {
        add_stats(A.stats, c_at, c_la, c_pt, c_px);
        // statistics word 5 (this kernel has no careful steps to report there): AT iterations executed
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
// Test hook (fs_test_block_threshold): the wave-uniform block threshold T of FS_FAST_LOOP_FDU, evaluated by the very macro the loop
// uses (FS_BT_T), one case per wave -- so that tests/test_gpu_block_threshold.py can hold it against its definition at the corners
// ("never" bounds under scale shifts of either sign and any size, overflow, max|dc| above the bound).
__global__ void k_test_block_threshold(const int *__restrict__ bw, const int *__restrict__ eshm, const int *__restrict__ sdc,
                                       int *__restrict__ t_out, uint32_t n)
{
    const uint32_t i = blockIdx.x;
    if (i >= n)
        return;
    const int b = __builtin_amdgcn_readfirstlane(bw[i]), e = __builtin_amdgcn_readfirstlane(eshm[i]),
              d = __builtin_amdgcn_readfirstlane(sdc[i]);
    int t;
    asm volatile("s_mov_b32 s67, %[b]\n\t" FS_BT_T("s67") : [t] "=&s"(t) : [b] "s"(b), [eshm] "s"(e), [sdc] "s"(d) : "s67", "scc");
    if (threadIdx.x == 0)
        t_out[i] = t;
}

void fsk_test_block_threshold(const int *bw, const int *eshm, const int *sdc, int *t_out, uint32_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_test_block_threshold, dim3(n), dim3(64), 0, s, bw, eshm, sdc, t_out, n);
}

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

template <int Mode, bool kStats, bool kScaled, bool kLds = false, bool kGpuStage = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) k_lav2_hdr32_fast(FsLav2Args32 A)
{
    __shared__ float4 s_zs_lds[kLds ? 4 * 2 * 64 : 1];
    // cost recording (A.tile_cost): a lane parks its count at the start of the perturbation loop here, so that nothing
    // extra stays in a register across the loop
    __shared__ uint32_t s_it0[256];
    // dc of the wave's pixels (two mantissas, one exponent): constant over the perturbation loop and needed only where a run or
    // a careful step starts, it is read back from here there instead of holding three registers across the loops (with the
    // hot runs of round 4 the register allocator had none left and spilled to scratch -- 180 MB of writes per frame)
    __shared__ float4 s_dcp[256]; // (16 bytes per lane: one shift for the address, ONE 12-byte LDS read for the three words)
    // ... and the state a scaled run starts from (dz's mantissas), needed again only when a run is repeated with the per-trip
    // verdicts: parked here for the run instead of held in two registers across it
    __shared__ float s_dzp[2 * 256];
    // The tile this wave renders, as two wave-uniform numbers: named by the launch order when there is one (longest tiles
    // first, from the costs the previous frame recorded), by the block index otherwise.  The pixel is tile + lane, and
    // it is worked out twice -- here, and again for the store at the end from the scalar tile numbers and a freshly
    // computed lane number -- so that neither the pixel position nor the thread index occupies vector registers across the
    // perturbation loop (they used to be spilled to scratch around it: 64 registers at 8 waves per SIMD, 15 of them
    // named by the hand-scheduled loop).
    uint32_t tile_x, tile_y;
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (A.tile_order) {
        const uint32_t w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave_in_block;
        const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.tile_order[w]);
        tile_y = tile != 0xFFFFFFFFu ? tile / A.tiles_x : 0u;
        tile_x = tile != 0xFFFFFFFFu ? tile - tile_y * A.tiles_x : 0xFFFFFFFu; // (no tile: a column beyond every frame)
    } else {
        tile_x = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
        tile_y = blockIdx.y;
    }
    uint32_t X, L;
    uint32_t lds_lane16 = 0; // (kLds) lane * 16, made here where all 64 lanes are active: see FS_GLDS_CHUNK
    {
        uint32_t lane; // (opaque, so that no later use of the lane number is served from a register kept since here)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        X = tile_x * 8u + (lane & 7u);
        L = tile_y * 8u + (lane >> 3);
        if constexpr (kLds)
            asm volatile("v_lshlrev_b32_e32 %0, 4, %1" : "=v"(lds_lane16) : "v"(lane));
    }
    uint32_t lane_cost = 0;
#ifdef FS_TRACE_WAVES
    // measurement build (tools/wave_trace.py): every wave records when and where it ran.  100 MHz constant clock.
    const uint64_t trace_t0 = wall_clock64();
#endif
    uint64_t c_at = 0, c_la = 0, c_pt = 0, c_px = 0;
    uint64_t c_careful = 0, c_scaled = 0, c_runs = 0;
    uint32_t c_why[4] = {0, 0, 0, 0};
    uint32_t c_nz[4] = {0, 0, 0, 0}; // (counting build) careful passes by the kind of entry they arrive at, see below
    bool was_skip = false;
    uint32_t c_wentry = 0, c_wstart = 0, c_wshort = 0; // run entries tried / runs started / runs of fewer than 8 steps (per wave)
    uint32_t c_blk_violation = 0; // (verification build) blocks that passed the block test and failed a bound test: must stay 0
    uint32_t c_pass = 0, c_generic = 0; // careful passes of the wave / those that took the generic step
    uint32_t c_blk_free = 0, c_blk_tested = 0; // 4-step blocks of the scalar-cache scaled path without / with bound tests (per wave)
    uint32_t c_lane_steps = 0, c_lane_runs = 0; // (counting build) wave-steps / runs taken on the per-lane entry path of the scaled runs
#ifdef FS_PROFILE_CYCLES
    uint64_t cyc_loop = 0, cyc_run = 0, cyc_body = 0, cyc_t0 = 0, cyc_t1 = 0, cyc_t2 = 0;
    uint64_t cyc_asm = 0, cyc_tested = 0, cyc_hot = 0, cyc_t3 = 0, cyc_t4 = 0, cyc_t5 = 0, wall_loop = 0, wall_t0 = 0;
#define FS_CYC(stmt) do { if (kStats) { stmt; } } while (0)
#else
#define FS_CYC(stmt) do { } while (0)
#endif
    const uint32_t Y = global_row(A.frame, L);
    const bool live = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const uint32_t n_iterations = A.n_iterations;
        hreal32 deltaReal, deltaImaginary;
        pixel_delta<float>(A.coords, X, Y, deltaReal, deltaImaginary);
        const hcplx32 DeltaSub0 = hc_from_hr(deltaReal, deltaImaginary);
        hcplx32 DeltaSubN = hc_from_native<float>(0.0f, 0.0f);
        uint32_t iterations = 0;
        uint32_t la_cost = 0; // what ran before the perturbation loop, in units of a perturbation step (tile cost only)

        if (Mode != FS_MODE_PO) {
            if (A.la_valid && A.use_at && hr_cmp_pos(hc_cheb(DeltaSub0), ldr(A.at.ThresholdC)) <= 0) {
                const uint32_t ATMaxIt = n_iterations / A.at.StepLength;
                hcplx32 c = hc_add(hc_mul(DeltaSub0, ldc(A.at.CCoeff)), ldc(A.at.RefC));
                hc_reduce(c);
                hcplx32 z;
                uint32_t i;
                at_perform<float>(c, ldr(A.at.SqrEscapeRadius), ATMaxIt, z, i);
                hcplx32 dz = hc_mul(z, ldc(A.at.InvZCoeff));
                hc_reduce(dz);
                DeltaSubN = dz;
                iterations = i * A.at.StepLength;
                la_cost = i;
                if (kStats)
                    c_at = i;
            }
        }

        uint32_t RefIteration = 0;
        const uint32_t MaxRefIteration = A.orbit_count - 1;
        if (iterations != 0 && !(RefIteration < MaxRefIteration) && A.period != 0)
            RefIteration = RefIteration % A.period;

        if (Mode != FS_MODE_PO) {
            uint32_t CurrentLAStage = A.la_valid ? A.stage_count : 0;
            const hreal32 dcCheb = hc_cheb(DeltaSub0);
            while (CurrentLAStage > 0) {
                CurrentLAStage--;
                const uint32_t LAIndex = A.stages[CurrentLAStage].LAIndex;
                {
                    const int cmp = hr_cmp_pos(dcCheb, ldr(A.las[LAIndex].LAThresholdC));
                    const bool invalid = kGpuStage ? (cmp >= 0) : (cmp < 0);
                    if (invalid)
                        continue;
                }
                const uint32_t MacroItCount = A.stages[CurrentLAStage].MacroItCount;
                uint32_t j = RefIteration;
                // (the Ref of record j + 1, read for the rebase test of step j, is the Ref step j + 1 starts from: one load
                // of it per step unless the test reset j)
                hcplx32 RefJ = hc_zero<float>();
                if (iterations < n_iterations)
                    RefJ = ldc(A.las[LAIndex + j].Ref);
                while (iterations < n_iterations) {
                    const fs_la_hdr32_u32 *LAj = &A.las[LAIndex + j];
                    const uint32_t l = LAj->StepLength;
                    bool unusable = true;
                    hcplx32 newDz = hc_zero<float>();
                    if (iterations + l <= n_iterations) {
                        newDz = hc_mul(DeltaSubN, hc_add(hc_mul2(RefJ), DeltaSubN));
                        hc_reduce(newDz);
                        unusable = hr_cmp_pos(hc_cheb(newDz), ldr(LAj->LAThreshold)) >= 0;
                    }
                    if (unusable) {
                        RefIteration = LAj->NextStageLAIndex;
                        break;
                    }
                    iterations += l;
                    la_cost += 8u;
                    if (kStats)
                        c_la++;
                    DeltaSubN = hc_add(hc_mul(newDz, ldc(LAj->ZCoeff)), hc_mul(DeltaSub0, ldc(LAj->CCoeff)));
                    const hcplx32 RefN = ldc(LAj[1].Ref);
                    const hcplx32 complex0 = hc_add(RefN, DeltaSubN);
                    j++;
                    const hreal32 lhs = hr_reduced(hc_cheb(complex0));
                    const hreal32 rhs = hr_reduced(hc_cheb(DeltaSubN));
                    if (hr_cmp_pos(lhs, rhs) < 0 || j >= MacroItCount) {
                        DeltaSubN = complex0;
                        j = 0;
                        RefJ = ldc(A.las[LAIndex].Ref);
                    } else {
                        RefJ = RefN;
                    }
                }
                if (iterations >= n_iterations)
                    break;
            }
        }

        if (Mode != FS_MODE_LAO) {
            const float4 *__restrict__ zr = A.zref;
            hcplx32 dz = DeltaSubN;
            const hcplx32 dc = DeltaSub0; // (parked in LDS below; not used past that)
            uint32_t ref = RefIteration;
            {
                uint32_t lane_s;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
                s_it0[wave_in_block * 64u + lane_s] = iterations - la_cost;
            }
            bool running = iterations < n_iterations;
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 dzm = {dz.re, dz.im};
            int dze = dz.e;
            {
                uint32_t lane_s;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
                volatile __attribute__((address_space(3))) float *pd =
                    (volatile __attribute__((address_space(3))) float *)s_dcp + (wave_in_block * 64u + lane_s) * 4u;
                pd[0] = dc.re, pd[1] = dc.im, pd[2] = __int_as_float(dc.e);
            }
            // the largest true max|dc| of the wave's pixels as a binary32 bit pattern, never below the true value (2^-126 for
            // anything smaller, +inf beyond the range): the dc half of the block test, FS_FAST_LOOP_FDU
            int sdc_bits;
            {
                const float mdc = fs_max_abs(dc.re, dc.im);
                const int de = dc.e < -400 ? -400 : (dc.e > 400 ? 400 : dc.e);
                const int lane_bits = mdc > 0.0f ? __float_as_int(__builtin_fmaxf(__builtin_amdgcn_ldexpf(mdc, de), 0x1p-126f))
                                                 : (mdc == 0.0f ? 0 : 0x7f800000);
                sdc_bits = wave_upper_bound_i32(lane_bits);
            }
            // (each use site reads dc back: FS_LOAD_DC declares dcm / dce in its scope)
#define FS_LOAD_DC()                                                                                                \
    f2 dcm;                                                                                                         \
    int dce;                                                                                                        \
    {                                                                                                               \
        uint32_t lane_d;                                                                                            \
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_d));             \
        const uint32_t dc_addr_ = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4 *)s_dcp +          \
                                  ((wave_in_block * 64u + lane_d) << 4);                                            \
        typedef float f3l_ __attribute__((ext_vector_type(3)));                                                     \
        f3l_ dc3_;                                                                                                  \
        asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dc3_) : "v"(dc_addr_) : "memory");       \
        dcm = (f2){dc3_.x, dc3_.y};                                                                                 \
        dce = __float_as_int(dc3_.z);                                                                               \
    }
            // (the orbit value the pixel is at is read where a quiet run or a careful step starts -- zq / zr [ref] -- instead of
            // being carried in registers across the runs)
            // The careful step as straight-line code for EVERY exponent order of its three sums.  plus_mutable
            // (HDRFloatComplex.h:219-247, hc_add) keeps the operand with the larger exponent and adds the other one scaled
            // by 2^-gap -- or not at all from a gap of 120 on: with f(g) = 2^g for g > -120 and 0 below,
            //   sum = a f(a.e - e) + b f(b.e - e),  e = max(a.e, b.e)
            // is the same two IEEE operations per part in each of its four arms (one factor is 1, the product by it exact;
            // a product by 0 adds a zero), up to the sign of a zero part, which no later operation can see.
            auto pow2_or_zero = [](int g) -> float {
                return g > -kExpDiffIgnored ? __int_as_float((int)(((uint32_t)g << 23) + 0x3F800000u)) : 0.0f; // g <= 0
            };
            // Quiet-run state: sC = ~(exponent of Zc) + 116 for an orbit value below 8, a large positive poison otherwise
            // (zq[i].z, written by k_make_quiet_orbit).
            const float4 *__restrict__ zq = A.zq;
            const float4 *__restrict__ zs = A.zs;
            FS_CYC(cyc_t0 = __builtin_readcyclecounter());
            FS_CYC(wall_t0 = wall_clock64());
            uint32_t sc_skip = 0, sc_penalty = 0; // (wave-uniform) back-off of the scaled-run attempts, see below
            bool hot_next = false; // (wave-uniform) the step a scaled run has just failed on goes to a hot run first (FS_HOT_AFTER_FAIL)
            bool fl_per_trip = false; // (wave-uniform) the next run attempt uses the per-trip floor verdicts (FS_FAST_LOOP_FD)
            while (running) {
                // ---- run of "scaled" quiet steps.  HDRFloat addition and multiplication are the correctly rounded binary32
                // operations on the represented values (an exponent gap >= 120 drops an addend that is far below half an
                // ulp of the other; Reduce only re-labels a value), so as long as nothing leaves binary32's normal range the
                // reference's step  dz' = dz (2Z + dz) + dc  can be carried out on plain floats under one fixed power-of-two
                // scale per lane:  w = dz 2^-E,  s = fma(w, 2^E, 2Z),  q = w s + dc 2^-E  -- the same IEEE operations on the
                // same (scaled) operands, hence the same bits.  A step is accepted when (all lanes of the wave)
                //   max|q| 2^E <= 2^-2 max|Z'|   |dz'| <= 0.354 |Z'| in the 2-norm: neither exit test of the CPU loop can fire
                //                                (|z| >= 0.646 |Z'| > 1.8 |dz'|: a 3.3x margin in the squares the rebase test
                //                                compares; |z|^2 < 115 with max|Z'| < 5.6), and Z' passed the companion's range
                //                                test.  (2^-3 was the first choice; 2^-2 loses fewer runs: -1.5 % frame time on View 5);
                //   min|q| >= 2^-40 max|q|       no part of a product that matters is lost below 2^-126 in either
                //                                representation (a dropped term is >= 2^40 below what it is added to);
                //   2^-20 <= max|q| <= 2^40      the scale still fits.
                // The last two are tested on every second step, and a two-step trip is dropped as a whole when either of
                // its steps fails: the first step of a trip starts from a state that passed them, so its own products are
                // exact; a part of its result that is out of proportion (or a result that left the window -- it cannot come
                // back from below in one step, the factor |2Z + dz| is < 2^5) either shows in the second step's result or
                // sits >= 2^80 below everything that result is made of.
                // Anything else leaves the state of the last accepted step to the exponent-tracking loop below.
                bool sc_stopped = false; // a scaled run ended on a step it could not take: that step goes to the careful path
                if (kScaled && (sc_skip != 0u || hot_next)) {
                    hot_next = false;
                    // back-off: the last run attempts of this wave ended before their first step (a lane sits where dz is not
                    // small against the orbit -- near its escape, or between two near-zero orbit values): an attempt costs an
                    // entry, a trip and an exit, so a few careful steps are taken before the next one
                    //
                    // ---- HOT RUN (round 4).  What the wave is waiting for is a pixel on its way out: for its last half-dozen
                    // steps its dz is no longer small against the orbit, it rebases every other step, and the 63 others
                    // take careful steps with it (356 of a wave's 454 careful passes on C3).  Those steps run here on the
                    // scaled form instead, PER LANE -- each lane at its own orbit position (entries through per-lane loads)
                    // and with the CPU loop's two exit tests evaluated exactly, in true scale, on every step:
                    //   z = Z' + q 2^E as one fma (the exact sum rounded once, like the reference's aligned sum; where dz is
                    //   far below binary32's range the product vanishes inside the fma and z = Z', which is what the
                    //   reference's sum rounds to as well: both parts of a usable entry are >= 2^-80);  |z|^2 and
                    //   |q|^2 2^2E as sums of squares (the same roundings up to the scale; a square that underflows belongs
                    //   to a part 2^40 below its sibling -- absorbed in both arithmetics -- or to a z that cancelled to below
                    //   2^-62 against a dz' >= 2^-41: the rebase test fires either way);  escape |z|^2 > 256: the pixel is
                    //   done;  rebase |z|^2 < |dz'|^2 (or the orbit's end): dz = z -- the rounded sum itself, scaled back --
                    //   at orbit index 0.
                    // The step is the reference's step while the floor form's conditions hold (both parts of every state,
                    // a rebased one included, >= 2^-56; max|w| < 2^24; the arrival entry usable); a lane that misses one
                    // ends the run for the wave before anything of that step is committed, and the careful step below
                    // decides.  The run also ends when every lane has cooled down (its arrival passes the bound test
                    // again: the fast paths resume) and after kHotRunSteps steps (the scale is re-centred).
                    bool hot_progress = false, hot_cold = false;
                    FS_CYC(cyc_t5 = __builtin_readcyclecounter());
                    {
                        FS_LOAD_DC()
                        const int E = dze + kScaleShift;
                        const float sE = __builtin_amdgcn_ldexpf(1.0f, E);
                        const f2 sE2 = {sE, sE};
                        const int dsh = dce - E;
                        const f2 dcs = {__builtin_amdgcn_ldexpf(dcm.x, dsh), __builtin_amdgcn_ldexpf(dcm.y, dsh)};
                        const float4 e0 = zs[ref];
                        const float mx0 = fs_max_abs(dzm.x, dzm.y);
                        const float mn0 = fs_min_abs(dzm.x, dzm.y);
                        const bool start_ok = scaled_startable(e0) && mn0 >= FS_FL_FLOOR * __builtin_amdgcn_ldexpf(1.0f, kScaleShift) &&
                                              mx0 >= 1.0f && mx0 < 2.0f && dsh <= 30 - kScaleShift;
                        if (__builtin_amdgcn_ballot_w64(!start_ok) == 0ull) {
                            f2 w = dzm * __builtin_amdgcn_ldexpf(1.0f, -kScaleShift);
                            bool live = true;
#pragma unroll 1
                            for (uint32_t budget = kHotRunSteps; budget != 0u; budget--) {
                                // One step, by hand, in the registers the hand-scheduled loops name (v[48:62] are free between
                                // those loops): written in C++ the run took nine registers more than the kernel has at eight waves
                                // per SIMD -- its temporaries, 64-bit per-lane addresses, and what the compiler hoists out of the
                                // loop (2Z of the entry the lane is at, 2 E, the shifted exponent) -- and the allocator spilled to
                                // scratch.  Here both orbit entries -- the one the lane is at, and the one it arrives at -- come
                                // through a scalar base and a 32-bit per-lane offset (a finished lane reads entries 0 and 1):
                                //   s = fma(w, 2^E, 2Z);  q = w s + dc 2^-E;  z = fma(q, 2^E, Z');  |z|^2;  |q|^2 2^2E;  max / min |q|;
                                //   hb = bits(max|q|) + (E << 23, clamped): the bound test's left side
                                // (a packed result read by the very next instruction needs one wait state: s_nop 0)
                                f2 q_, zt;
                                float nz, nq, mxq, mnq, entz;
                                int hb;
                                {
                                    const uint32_t off_ = (live ? ref + 1u : 1u) << 4;
                                    asm volatile("global_load_dwordx2 v[58:59], %[off], %[zs] offset:-16\n\t"
                                                 "global_load_dwordx3 v[60:62], %[off], %[zs]\n\t"
                                                 "s_waitcnt vmcnt(0)\n\t"
                                                 "v_pk_fma_f32 v[56:57], %[w], %[se], v[58:59]\n\t"
                                                 "v_mul_f32_e32 v50, 0.5, v60\n\t"
                                                 "v_pk_mul_f32 v[58:59], %[w], v[56:57] op_sel_hi:[0,1]\n\t"
                                                 "v_pk_mul_f32 v[56:57], %[w], v[56:57] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
                                                 "v_mul_f32_e32 v51, 0.5, v61\n\t"
                                                 "v_pk_add_f32 v[58:59], v[58:59], v[56:57] neg_lo:[0,1] neg_hi:[0,0]\n\t"
                                                 "v_max_i32_e32 v60, 0xffffff02, %[e]\n\t" /* E clamped to -254 .. 127 */
                                                 "v_min_i32_e32 v60, 0x7f, v60\n\t"
                                                 "v_pk_add_f32 v[48:49], v[58:59], %[dc]\n\t"
                                                 "v_lshlrev_b32_e32 v60, 23, v60\n\t"
                                                 "v_pk_fma_f32 v[50:51], v[48:49], %[se], v[50:51]\n\t"
                                                 "v_pk_mul_f32 v[58:59], v[48:49], v[48:49]\n\t"
                                                 "v_max_f32_e64 v54, |v48|, |v49|\n\t"
                                                 "v_pk_mul_f32 v[56:57], v[50:51], v[50:51]\n\t"
                                                 "v_add_f32_e32 v53, v58, v59\n\t"
                                                 "v_min_f32_e64 v55, |v48|, |v49|\n\t"
                                                 "v_add_f32_e32 v52, v56, v57\n\t"
                                                 "v_lshlrev_b32_e32 v61, 1, %[e]\n\t"
                                                 "v_add_u32_e32 v56, v54, v60\n\t"
                                                 "v_ldexp_f32 v53, v53, v61"
                                                 : "=&{v[48:49]}"(q_), "=&{v[50:51]}"(zt), "=&{v52}"(nz), "=&{v53}"(nq), "=&{v54}"(mxq),
                                                   "=&{v55}"(mnq), "=&{v56}"(hb), "=&{v62}"(entz) /* (early clobber: no input may share one) */
                                                 : [w] "v"(w), [se] "v"(sE2), [dc] "v"(dcs), [e] "v"(E), [off] "v"(off_), [zs] "s"(zs)
                                                 : "v57", "v58", "v59", "v60", "v61", "memory");
                                }
                                const bool esc = nz > 256.0f;
                                const bool reb = !esc && (nz < nq || ref + 1u >= MaxRefIteration);
                                bool valid = mnq >= FS_FL_FLOOR && mxq < FS_FL_HIGH_TRIP && __float_as_int(entz) != (int)0x80000000 &&
                                             nz == nz;
                                // the rebased state dz = z in the run's scale (it would overflow where dz is tiny -- where no
                                // rebase happens), formed only on the steps on which some lane rebases
                                f2 wz = q_;
                                if (__builtin_amdgcn_ballot_w64(live && reb) != 0ull) {
                                    wz = (f2){__builtin_amdgcn_ldexpf(zt.x, -E), __builtin_amdgcn_ldexpf(zt.y, -E)};
                                    const float mxz = fs_max_abs(wz.x, wz.y);
                                    const float mnz = fs_min_abs(wz.x, wz.y);
                                    valid = valid && (!reb || (mnz >= FS_FL_FLOOR && mxz < FS_FL_HIGH_TRIP));
                                }
                                if (__builtin_amdgcn_ballot_w64(live && !valid) != 0ull)
                                    break;
                                const bool cold = !(hb > __float_as_int(entz)) && !reb;
                                if (live) {
                                    hot_progress = true;
                                    if (kStats) {
                                        c_pt++;
                                        c_scaled++;
                                    }
                                    if (esc) {
                                        live = false;
                                        running = false; // `break` happens before iterations++ in the CPU loop
                                    } else {
                                        iterations++;
                                        if (reb) {
                                            w = wz;
                                            ref = 0u; // (2 Z[0] is an exact zero: the orbit starts there)
                                        } else {
                                            w = q_;
                                            ref++;
                                        }
                                        if (iterations >= n_iterations) {
                                            live = false;
                                            running = false;
                                        }
                                    }
                                }
                                if (__builtin_amdgcn_ballot_w64(live) == 0ull)
                                    break;
                                if (__builtin_amdgcn_ballot_w64(live && !cold) == 0ull) {
                                    hot_cold = true;
                                    break;
                                }
                            }
                            {
                                // back to the reduced form (exact; every accepted state has two non-zero parts) -- also for a
                                // lane that took no step: its w is dz 2^-E still, and rebuilding dz, its exponent and the orbit
                                // value from it means that none of the three has to stay in a register across the run
                                const float mxw = fs_max_abs(w.x, w.y);
                                const int k = (int)((uint32_t)__float_as_int(mxw) >> 23) - 127;
                                dzm = (f2){__builtin_amdgcn_ldexpf(w.x, -k), __builtin_amdgcn_ldexpf(w.y, -k)};
                                dze = E + k;
                            }
                        }
                    }
                    FS_CYC(cyc_hot += __builtin_readcyclecounter() - cyc_t5);
                    if (__builtin_amdgcn_ballot_w64(hot_progress) != 0ull) { // (wave-uniform: a lane that is done has progressed)
                        if (hot_cold)
                            sc_skip = 0u, sc_penalty = 0u;
                        continue;
                    }
                    if (sc_skip != 0u)
                        sc_skip--;
                    sc_stopped = true;
                    was_skip = true;
                } else if (kScaled) {
                    typedef float f3 __attribute__((ext_vector_type(3)));
                    FS_CYC(cyc_t1 = __builtin_readcyclecounter());
                    for (;;) {
                        const float4 e0 = zs[ref];
                        // (floor form, see FS_FAST_LOOP_FL: the run's scale puts max|w| at 2^-24)
                        const int E = dze + kScaleShift;
                        const float sE = __builtin_amdgcn_ldexpf(1.0f, E); // 0 / denormal below 2^-126: dz then cannot matter
                        FS_LOAD_DC()
                        const int dsh = dce - E;
                        const f2 dcs = {__builtin_amdgcn_ldexpf(dcm.x, dsh), __builtin_amdgcn_ldexpf(dcm.y, dsh)};
                        const float mx0 = fs_max_abs(dzm.x, dzm.y);
                        const float mn0 = fs_min_abs(dzm.x, dzm.y);
                        const uint32_t left_ref = ref + 1 < MaxRefIteration ? MaxRefIteration - 1 - ref : 0u;
                        const uint32_t left_it = n_iterations - 1 - iterations;
                        const uint32_t left = left_ref < left_it ? left_ref : left_it;
                        // max|w| 2^E <= bound, on the bit patterns: for positive floats the exponent shift is an integer add,
                        // a result below the normal range turns negative (dz far too small to matter: passes), NaN is huge
                        // (max|w| < 2^29 and |dz| = |w| 2^E < 4: the sum of the two exponent fields stays inside a float's)
                        const int Esh = (E < -254 ? -254 : (E > 127 ? 127 : E)) * (1 << 23);
                        // (the state a run starts from has passed the CPU loop's tests already: only the entry it starts at
                        // must be one the companion vouches for -- scaled_startable: 2Z exact in true scale)
                        // dz 2^-E is exact and above the floor; dc 2^-E <= 2^7 (the same dc <= 2^30 dz as before)
                        const bool start_ok = scaled_startable(e0) && mn0 >= FS_FL_FLOOR * __builtin_amdgcn_ldexpf(1.0f, kScaleShift) && mx0 >= 1.0f && mx0 < 2.0f &&
                                              dsh <= 30 - kScaleShift;
                        // run length: the longest of 256 / 64 / 16 steps that every lane still has before the orbit ends
                        // and before its iteration limit (three votes per run, not a counter per step)
                        const uint32_t run_len = scaled_run_length(left);
                        if (kStats) {
                            c_wentry++;
                            // entries that fail (wave votes; tools/scaled_share_probe.py)
                            if (run_len == 0u || __builtin_amdgcn_ballot_w64(!start_ok) != 0ull)
                                c_why[0]++;
                        }
                        if (__builtin_amdgcn_ballot_w64(!start_ok) != 0ull || run_len == 0u)
                            break;
                        {
                            uint32_t lane_p;
                            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
                            volatile __attribute__((address_space(3))) float *pz =
                                (volatile __attribute__((address_space(3))) float *)s_dzp + (wave_in_block * 64u + lane_p) * 2u;
                            pz[0] = dzm.x, pz[1] = dzm.y;
                        }
                        const f2 sE2 = {sE, sE};
                        // One step from (W_, Z_) into (NW_, NZ_); V accumulates the lanes that fail a test.
                        // AFTER_ARITH is the statement that waits for the entry (tied to the step's results so that it stays
                        // behind the arithmetic); EX / EY / EB name the arrival entry's 2Z and bound.
                        // Two steps are tested together and the state ping-pongs between two register sets over two such
                        // trips, so neither the back-edge nor the roll-back of a failed trip needs a register copy: a
                        // trip that contains a failing step is dropped as a whole and its first step goes to the
                        // careful path.
                        // A trip that fails is rolled back to its start -- unless its first step is good on its own: the
                        // bound test it passed in the loop plus, now, the two tests the loop only applies to second steps.
                        // Then the first step's result is the exit state and only the second step goes to the careful path
                        // (a near-zero orbit entry otherwise costs two careful steps and two run entries when it sits second).
#define FS_TRIP_FAILED(T, NW_, EB, WSTART)                                                                         \
    {                                                                                                               \
        const float mn_s = fs_min_abs(NW_.x, NW_.y);                         \
        const uint64_t bad_s =                                                                                      \
            __builtin_amdgcn_ballot_w64(__float_as_int(mx_##T) + Esh > __float_as_int(EB)) |                        \
            __builtin_amdgcn_ballot_w64(!(mn_s >= FS_FL_FLOOR));                                                    \
        if (bad_s == 0ull) {                                                                                        \
            wO = NW_;                                                                                               \
            c += 1;                                                                                                 \
        } else {                                                                                                    \
            wO = WSTART;                                                                                            \
        }                                                                                                           \
        failed = true;                                                                                              \
    }
#define FS_TRIP_FAILED_NB(T, NW_, EB, WSTART)                                                                      \
    {                                                                                                               \
        const float mx_s = fs_max_abs(NW_.x, NW_.y);                         \
        const float mn_s = fs_min_abs(NW_.x, NW_.y);                         \
        const uint64_t bad_s =                                                                                      \
            __builtin_amdgcn_ballot_w64(__float_as_int(mx_s) + Esh > __float_as_int(EB)) |                        \
            __builtin_amdgcn_ballot_w64(!(mn_s >= FS_FL_FLOOR));                                                  \
        if (bad_s == 0ull) {                                                                                        \
            wO = NW_;                                                                                               \
            c += 1;                                                                                                 \
        } else {                                                                                                    \
            wO = WSTART;                                                                                            \
        }                                                                                                           \
        failed = true;                                                                                              \
    }
                        f2 w0 = dzm * __builtin_amdgcn_ldexpf(1.0f, -kScaleShift), z0 = {e0.x, e0.y}, w2, z2, wO;
                        uint32_t c = 0;
                        bool failed;
                        FS_CYC(cyc_t2 = __builtin_readcyclecounter());
                        bool fl_redo = false;
                        const uint32_t ref_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)ref);
                        if (__builtin_amdgcn_ballot_w64(ref != ref_u) == 0ull) {
                            // Every lane of the wave reads the same orbit entries (the usual case: neighbouring pixels
                            // rebase on the same step): the entries come through the scalar cache into scalar registers,
                            // four per body, and the vector memory path -- whose 12-byte returns cost the SIMD about as
                            // much as four vector instructions per step -- stays idle.
                            // Eight entries (two 64-byte lines) per body, one wait: a scalar-cache miss is an L2 round trip,
                            // and these loads cannot be waited for one at a time.
                            if constexpr (kLds) {
                                // ---- entries through LDS (see the kernel's header comment)
                                float4 *wbuf = s_zs_lds + (threadIdx.x >> 6) * 128u;
                                const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane(
                                    (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4 *)wbuf);
                                const uint32_t nchunks = (run_len + 63u) >> 6;
// (round 4) NOTHING may be written to a vector register while EXEC is widened: the register the compiler picks for a
// temporary holds, in the lanes that are masked off, whatever those lanes' pixels still need -- the first form of this
// statement built the per-lane offset there and, once the allocation had shifted, overwrote finished pixels' step counters
// (tests/test_gpu_variants.py caught it).  The per-lane byte offset (lane * 16) is therefore made ONCE, at the top of the
// kernel where every lane is active, kept for the kernel's lifetime and only read here; the chunk's first entry goes into
// the wave-uniform base.  Entries past the orbit's end land in the arrays that follow zs in the same allocation.
// ALL 64 lanes take part in the LDS-DMA whatever the loop's EXEC mask is (lanes whose pixel has finished are masked
// off here, and a masked lane would leave its 16-byte slot of the chunk unwritten): EXEC is widened for the one
// instruction, the lane number and the (clamped) entry offset are rebuilt inside the widened region, then EXEC and M0
// are restored.  vaddr = 32-bit byte offset from the scalar base (the orbit arrays are far below 4 GiB).
#define FS_GLDS_CHUNK(CH)                                                                                           \
    {                                                                                                               \
        const float4 *src_ = zs + (ref_u + 1u + (CH) * 64u); /* wave-uniform; the arrays behind zs are the slack */  \
        const uint32_t dst_ = lds_base + (((CH) & 1u) << 10);                                                       \
        uint32_t keep_;                                                                                             \
        uint64_t exec_;                                                                                             \
        asm volatile("s_or_saveexec_b64 %0, -1\n\t"                                                               \
                     "s_mov_b32 %1, m0\n\t"                                                                       \
                     "s_mov_b32 m0, %3\n\t"                                                                       \
                     "s_nop 0\n\t"                                                                                \
                     "global_load_lds_dwordx4 %2, %4\n\t"                                                         \
                     "s_mov_b32 m0, %1\n\t"                                                                       \
                     "s_mov_b64 exec, %0"                                                                           \
                     : "=&s"(exec_), "=&s"(keep_)                                                                   \
                     : "v"(lds_lane16), "s"(dst_), "s"(src_)                                                        \
                     : "memory", "scc");                                                                            \
    }
                                FS_GLDS_CHUNK(0u)
                                uint32_t chunk = 0;
                                bool run_over = false;
                                while (!run_over) {
                                    if (chunk + 1u < nchunks) {
                                        FS_GLDS_CHUNK(chunk + 1u)
                                        asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); // this chunk has landed, the next flies
                                    } else {
                                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                    }
                                    const float4 *cb = wbuf + ((chunk & 1u) << 6);
                                    const uint32_t steps_here = run_len - (chunk << 6) < 64u ? run_len - (chunk << 6) : 64u;
                                    for (uint32_t e8 = 0; e8 < steps_here; e8 += 8u) {
                                        const float4 ua = cb[e8], ub = cb[e8 + 1], uc = cb[e8 + 2], ud = cb[e8 + 3];
                                        const float4 ue = cb[e8 + 4], uf = cb[e8 + 5], ug = cb[e8 + 6], uh = cb[e8 + 7];
                                        f2 t1, u1;
                                        uint64_t v1 = 0;
                                        FS_SCALED_STEP(w0, z0, t1, u1, a, v1, false, (void)0, ua.x, ua.y, ua.z);
                                        FS_SCALED_STEP(t1, u1, w2, z2, b, v1, true, (void)0, ub.x, ub.y, ub.z);
                                        if (v1 != 0ull) {
                                            FS_TRIP_FAILED(a, t1, ua.z, w0)
                                            run_over = true;
                                            break;
                                        }
                                        c += 2;
                                        f2 t3, u3;
                                        uint64_t v2 = 0;
                                        FS_SCALED_STEP(w2, z2, t3, u3, c_, v2, false, (void)0, uc.x, uc.y, uc.z);
                                        FS_SCALED_STEP(t3, u3, w0, z0, d, v2, true, (void)0, ud.x, ud.y, ud.z);
                                        if (v2 != 0ull) {
                                            FS_TRIP_FAILED(c_, t3, uc.z, w2)
                                            run_over = true;
                                            break;
                                        }
                                        c += 2;
                                        f2 t5, u5;
                                        uint64_t v3 = 0;
                                        FS_SCALED_STEP(w0, z0, t5, u5, e, v3, false, (void)0, ue.x, ue.y, ue.z);
                                        FS_SCALED_STEP(t5, u5, w2, z2, f, v3, true, (void)0, uf.x, uf.y, uf.z);
                                        if (v3 != 0ull) {
                                            FS_TRIP_FAILED(e, t5, ue.z, w0)
                                            run_over = true;
                                            break;
                                        }
                                        c += 2;
                                        f2 t7, u7;
                                        uint64_t v4 = 0;
                                        FS_SCALED_STEP(w2, z2, t7, u7, g, v4, false, (void)0, ug.x, ug.y, ug.z);
                                        FS_SCALED_STEP(t7, u7, w0, z0, h, v4, true, (void)0, uh.x, uh.y, uh.z);
                                        if (v4 != 0ull) {
                                            FS_TRIP_FAILED(g, t7, ug.z, w2)
                                            run_over = true;
                                            break;
                                        }
                                        c += 2;
                                    }
                                    if (run_over)
                                        break;
                                    chunk++;
                                    if (c >= run_len) {
                                        wO = w0, failed = false;
                                        run_over = true;
                                    }
                                }
                                // a run that stopped early may have left its prefetch in flight: it lands before the buffers are reused
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef FS_GLDS_CHUNK
                            } else {
                            typedef float f4 __attribute__((ext_vector_type(4)));
                            const float4 *zpu = zs + ref_u + 1;
                            // Block test (k_make_quiet_orbit's .w): when max(max|w|, max|dc|) at a block's first entry is
                            // within that entry's block bound, its four arrivals pass their bound tests whatever else
                            // happens, and the block runs without them -- as the hand-scheduled body below (eight steps;
                            // it stops after four when the second block needs its bound tests).  Blocks that need them run
                            // the tested C++ form, four steps at a time.
#if defined(FS_FD_LANE_BOUND) || defined(FS_VERIFY_BLOCK_BOUND) || defined(FS_VERIFY_FLOOR) || !FS_FL_EVERY
                            const int imdc = __float_as_int(fs_max_abs(dcs.x, dcs.y));
#endif
                            // (FS_FAST_LOOP_FDU) the largest scale shift of the running lanes
                            const int Esh_cap = wave_upper_bound_i32(Esh);
                            float mxS = mx0 * __builtin_amdgcn_ldexpf(1.0f, -kScaleShift);
                            int pwi = __builtin_amdgcn_readfirstlane(__float_as_int(e0.w));
                            // (all lanes sit at the same entry here: 2Z of the entry the state is at lives in scalar registers)
                            f2 zS = {__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(e0.x))),
                                     __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(e0.y)))};
                            f2 wv = w0;
                            uint32_t cs = 0;
                            const uint32_t rl = (uint32_t)__builtin_amdgcn_readfirstlane((int)run_len);
                            const uint32_t lim8 = (rl << 4) - 0x80u; // run lengths are 16 / 64 / 256 steps
                            const float4 *const zpb = zpu;
                            for (;;) {
#ifdef FS_VERIFY_BLOCK_BOUND
                                // VERIFICATION BUILD (tools/block_bound_check.py): every block runs the tested form, and a block
                                // whose block test passes while one of its four arrivals fails its own bound test is counted
#ifdef FS_FD_LANE_BOUND
                                const int vg_ = __float_as_int(mxS) > imdc ? __float_as_int(mxS) : imdc;
                                const bool bt_pass = __builtin_amdgcn_ballot_w64(vg_ + Esh > pwi) == 0ull;
#else
                                // (the block test of FS_FAST_LOOP_FDU, restated)
                                const long long bt_d = (long long)pwi - (long long)Esh_cap;
                                const int bt_thr = sdc_bits > pwi ? -1 : (bt_d > 0x46800000ll ? 0x46800000 : (int)bt_d);
                                const bool bt_pass = __builtin_amdgcn_ballot_w64(__float_as_int(mxS) > bt_thr) == 0ull;
#endif
                                if (kStats && bt_pass)
                                    c_blk_free++;
#else
                                {
                                    // the untested bodies, as long as they last: status 0 = stopped in front of a block
                                    // that needs its tests or of the last four steps (or at the end of the run); 1 / 2 =
                                    // the first / second trip of a block failed (start state, first step: wv / r1,
                                    // r2 / r3; pwi = the first arrival's bound; cs counts the steps before the trip)
                                    f2 r1, r2, r3 = wv, ts_, ta_; // (r3 = wv: the pending pair on entry is the state itself)
                                    uint64_t xacc_ = 0; // (verification build: lanes whose first state of a trip was below 2^-56)
                                    float tn_, tl_;
                                    uint64_t msk_;
                                    int st, ebo, pf_, pg_, ph_, bt_t_;
                                    const uint32_t c_in = cs;
                                    uint32_t off = cs << 4;
                                    FS_CYC(cyc_t3 = __builtin_readcyclecounter());
#if FS_FL_EVERY && !defined(FS_VERIFY_FLOOR)
                                    if (!fl_per_trip) {
                                        {
                                            // (the 16-step body of k_perturb_scalar, FS_FAST_LOOP_FD16, measures 2 % slower here --
                                            // 47.8 - 48.1 against 46.6 - 47.0 ms at N = 1, 6.79 against 6.70 ms on the slowest of
                                            // eight emulated ranks: with seven waves per SIMD the round trip it halves is hidden)
#ifdef FS_FD_LANE_BOUND /* A/B: round 4's per-lane block test (five vector instructions per block) */
                                            FS_FAST_LOOP_FD(FS_PF_NONE, FS_BT_DC_MAX, FS_BT_DC_ADD, FS_BT_H_CMP, FS_BT_H_OR);
#else
                                            FS_FAST_LOOP_FDU(FS_PF_NONE);
#endif
                                        }
                                        ebo = 0;
                                    } else
#endif
                                    {
#if !(defined(FS_FD_LANE_BOUND) || defined(FS_VERIFY_BLOCK_BOUND) || defined(FS_VERIFY_FLOOR) || !FS_FL_EVERY)
                                        // (max|dc| in the run's scale: only this loop's per-lane block test reads it)
                                        const int imdc = __float_as_int(fs_max_abs(dcs.x, dcs.y));
#endif
                                        FS_FAST_LOOP_FL(FS_PF_NONE);
                                    }
#ifdef FS_VERIFY_FLOOR
                                    if (kStats && xacc_ != 0ull)
                                        c_blk_violation++;
#endif
                                    FS_CYC(cyc_asm += __builtin_readcyclecounter() - cyc_t3);
                                    st = __builtin_amdgcn_readfirstlane(st); // (asm results count as divergent)
                                    zS = (f2){__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(zS.x))),
                                              __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(zS.y)))};
                                    if (st == 3) {
                                        // a state of this invocation fell below the floor (deferred verdict): nothing of the run
                                        // has been committed -- the same run again, with the per-trip verdicts
                                        fl_redo = true;
                                        break;
                                    }
                                    cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)off) >> 4;
                                    pwi = __builtin_amdgcn_readfirstlane(pwi);
                                    if (kStats)
                                        c_blk_free += (cs - c_in) >> 2;
                                    if (st != 0) {
                                        const float ebf = __int_as_float(__builtin_amdgcn_readfirstlane(ebo));
                                        c = cs;
                                        if (st == 1) {
                                            FS_TRIP_FAILED_NB(a, r1, ebf, wv)
                                        } else {
                                            FS_TRIP_FAILED_NB(a, r3, ebf, r2)
                                        }
                                        break;
                                    }
                                }
#endif
                                if (cs + 4u > rl) {
                                    c = cs, wO = wv, failed = false;
                                    break;
                                }
                                // H where a block starts (the untested loop leaves here for it too): the run ends and the next
                                // one re-centres the scale
                                if (__builtin_amdgcn_ballot_w64(!(mxS < FS_FL_HIGH)) != 0ull) {
                                    c = cs, wO = wv, failed = false;
                                    break;
                                }
                                // a block with its bound tests: four entries as one 64-byte scalar load (s_load_dwordx16
                                // takes any dword-aligned address)
                                if (kStats)
                                    c_blk_tested++;
                                FS_CYC(cyc_t4 = __builtin_readcyclecounter());
                                typedef float f16 __attribute__((ext_vector_type(16)));
                                f16 U;
                                asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(U) : "s"(zpb + cs));
                                f2 tp_, tq_;
                                FS_STEP_ARITH(wv, zS, tp_, a)
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(U), "+v"(tp_));
                                const f4 ua = U.s0123, ub = U.s4567, uc = U.s89ab, ud = U.scdef;
                                const f2 up_ = {ua.x, ua.y}, uq_ = {uc.x, uc.y};
                                uint64_t vp_ = 0, vq_ = 0;
                                c = cs;
                                FS_STEP_BOUND(tp_, a, vp_, ua.z)
                                FS_STEP_FLOOR_FIRST(tp_, vp_)
                                FS_STEP_ARITH(tp_, up_, w2, b)
                                FS_STEP_BOUND(w2, b, vp_, ub.z)
                                FS_STEP_FLOOR(w2, vp_)
#ifdef FS_VERIFY_BLOCK_BOUND
                                if (kStats && bt_pass &&
                                    (__builtin_amdgcn_ballot_w64(__float_as_int(mx_a) + Esh > __float_as_int(ua.z)) |
                                     __builtin_amdgcn_ballot_w64(__float_as_int(mx_b) + Esh > __float_as_int(ub.z))) != 0ull)
                                    c_blk_violation++;
#endif
                                if (vp_ != 0ull) {
                                    FS_TRIP_FAILED(a, tp_, ua.z, wv)
                                    break;
                                }
                                z2 = (f2){ub.x, ub.y};
                                c += 2;
                                FS_STEP_ARITH(w2, z2, tq_, c_)
                                FS_STEP_BOUND(tq_, c_, vq_, uc.z)
                                FS_STEP_FLOOR_FIRST(tq_, vq_)
                                f2 w4;
                                FS_STEP_ARITH(tq_, uq_, w4, d)
                                FS_STEP_BOUND(w4, d, vq_, ud.z)
                                FS_STEP_FLOOR(w4, vq_)
#ifdef FS_VERIFY_BLOCK_BOUND
                                if (kStats && bt_pass &&
                                    (__builtin_amdgcn_ballot_w64(__float_as_int(mx_c_) + Esh > __float_as_int(uc.z)) |
                                     __builtin_amdgcn_ballot_w64(__float_as_int(mx_d) + Esh > __float_as_int(ud.z))) != 0ull)
                                    c_blk_violation++;
#endif
                                if (vq_ != 0ull) {
                                    FS_TRIP_FAILED(c_, tq_, uc.z, w2)
                                    break;
                                }
                                cs += 4;
                                wv = w4, mxS = mx_d, zS = (f2){ud.x, ud.y}, pwi = __float_as_int(ud.w);
                                FS_CYC(cyc_tested += __builtin_readcyclecounter() - cyc_t4);
                                if (cs >= rl) {
                                    c = cs, wO = wv, failed = false;
                                    break;
                                }
                            }
                            }
                        } else {
                            // per-lane orbit positions: one 12-byte vector load per step from a wave-uniform base plus a
                            // per-lane byte offset that is fixed for the run; the four loads of a body are requested up
                            // front and arrive in order
                            const uint32_t lane_off = (ref + 1) * 16u;
                            const float4 *zp = zs;
                            f3 ent_a, ent_b, ent_c_, ent_d;
                            for (;;) {
                                FS_SCALED_LOAD("0", a, w0)
                                FS_SCALED_LOAD("16", b, w0)
                                FS_SCALED_LOAD("32", c_, w0)
                                FS_SCALED_LOAD("48", d, w0)
                                f2 t1, u1;
                                uint64_t v1 = 0;
                                FS_SCALED_STEP(w0, z0, t1, u1, a, v1, false,
                                               asm volatile("s_waitcnt vmcnt(3)" : "+v"(ent_a), "+v"(mx_a)), ent_a.x,
                                               ent_a.y, ent_a.z);
                                FS_SCALED_STEP(t1, u1, w2, z2, b, v1, true,
                                               asm volatile("s_waitcnt vmcnt(2)" : "+v"(ent_b), "+v"(mx_b)), ent_b.x,
                                               ent_b.y, ent_b.z);
                                if (v1 != 0ull) {
                                    FS_TRIP_FAILED(a, t1, ent_a.z, w0)
                                    break;
                                }
                                c += 2;
                                f2 t3, u3;
                                uint64_t v2 = 0;
                                FS_SCALED_STEP(w2, z2, t3, u3, c_, v2, false,
                                               asm volatile("s_waitcnt vmcnt(1)" : "+v"(ent_c_), "+v"(mx_c_)), ent_c_.x,
                                               ent_c_.y, ent_c_.z);
                                FS_SCALED_STEP(t3, u3, w0, z0, d, v2, true,
                                               asm volatile("s_waitcnt vmcnt(0)" : "+v"(ent_d), "+v"(mx_d)), ent_d.x,
                                               ent_d.y, ent_d.z);
                                if (v2 != 0ull) {
                                    FS_TRIP_FAILED(c_, t3, ent_c_.z, w2)
                                    break;
                                }
                                c += 2;
                                zp += 4;
                                if (c >= run_len) {
                                    wO = w0, failed = false;
                                    break;
                                }
                            }
                            // a run that ends in its first trip leaves the loads of the second in flight: they land before anything else happens
                            asm volatile("s_waitcnt vmcnt(0) ; scaled run, loop exit" ::"v"(ent_a), "v"(ent_b), "v"(ent_c_), "v"(ent_d));
                            if (kStats) {
                                c_lane_steps += c;
                                c_lane_runs++;
                            }
                        }
#undef FS_TRIP_FAILED
#undef FS_TRIP_FAILED_NB
                        FS_CYC(cyc_body += __builtin_readcyclecounter() - cyc_t2);
                        if (fl_redo) {
                            // the same run again from its start state: dz's mantissas come back from where they were parked
                            uint32_t lane_p;
                            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
                            const volatile __attribute__((address_space(3))) float *pz =
                                (const volatile __attribute__((address_space(3))) float *)s_dzp + (wave_in_block * 64u + lane_p) * 2u;
                            dzm = (f2){pz[0], pz[1]};
                            dze = E - kScaleShift;
                            fl_per_trip = true;
                            continue;
                        }
                        fl_per_trip = false;
                        // back to the reduced form: the larger part's exponent moves into dze (exact).  Also when the run took
                        // no step: wO is dz 2^-E then, and rebuilding dz, its exponent and the orbit value from what the run
                        // ends with means that none of them has to survive the run in a register (round 4: with the hot runs
                        // the allocator had spilled them to scratch around every run)
                        {
                            const float mxw = fs_max_abs(wO.x, wO.y);
                            const int k = (int)((uint32_t)__float_as_int(mxw) >> 23) - 127;
                            dzm = (f2){__builtin_amdgcn_ldexpf(wO.x, -k), __builtin_amdgcn_ldexpf(wO.y, -k)};
                            dze = E + k;
                            ref += c;
                            iterations += c;
                            if (kStats && c != 0u) {
                                c_pt += c;
                                c_scaled += c;
                                c_runs++;
                                c_wstart++;
                                if (c < 8u)
                                    c_wshort++;
                            }
                        }
                        if (failed) {
                            sc_stopped = true;
                            hot_next = FS_HOT_AFTER_FAIL != 0;
                            if (c == 0u) {
                                sc_penalty = sc_penalty < kScaledBackoffCap ? sc_penalty + 1u : kScaledBackoffCap;
                                sc_skip = sc_penalty;
                            } else if (c >= 8u) {
                                sc_penalty = 0u;
                            }
                            break;
                        }
                    }
                    FS_CYC(cyc_run += __builtin_readcyclecounter() - cyc_t1);
                    // The step the run failed on: a hot run takes it (per lane, exit tests exact, 32 vector instructions) where it
                    // can -- 70 of a wave's 97 careful passes (130 vector instructions each, and a run entry behind every one)
                    // found that nothing happens at such a step -- and the careful step below where it cannot.
                    if (hot_next)
                        continue;
                }
                // ---- run of "quiet" steps: when dz is at least 2^4 below the orbit value and the orbit value is < 8,
                // neither exit test can fire and z itself is not needed:
                //   |Z'| in [0.5, 2.83) 2^Zne (larger part of an orbit entry is in [0.5, 2)),  |dz| < 2.83 * 2^qe
                //   qe <= Zne - 4  =>  |dz| < 0.18 * 2^Zne,  |z| = |Z' + dz| in (0.32, 3.01) * 2^Zne
                //   => |z| > 1.8 |dz|  (no rebase: Reduce(|z|^2) < Reduce(|dz|^2) is false with a 3x margin in the squares)
                //   => |z| < 12.1 for Zne <= 2 (no escape: |z|^2 > 256 is false with a 1.7x margin)
                // float rounding moves these norms by < 1e-6 relative, so the CPU function takes the same decisions.
                // The run continues while EVERY running lane of the wave is quiet.  The conditions are *sufficient*
                // ones (a lane that fails them takes the careful step below, which decides exactly):
                //   t1 = max(nd1, nd2, nd3 + 4) <= 0   (orbit bigger than dz / p bigger than dc / dz' 2^4 below Z')
                //   t2 = min(nd1, nd3 + 4) >= -115      (both alignment gaps inside the reference's 120 window)
                //   larger part of q a finite normal float; orbit value below 8 (poisoned sN fails t1 otherwise).
                // A lane must also stay clear of the orbit end and of its iteration limit (`left`); runs are cut into
                // chunks of 64 steps (this loop; kScaledChunk in the scaled runs) so that this is a per-chunk wave vote instead of a per-step, per-lane counter.
                // One quiet step from state (DZM, DZE, ZCM, SC, W) into (NDZM, NDZE, NZCM, NSC, NW) against entry K of the run.
                // Exponent bookkeeping is biased so that every range test is against a constant that needs no extra add:
                //   SC = ~exp(Zc) + 116 (zq[].z; poison 2^24 for an orbit value >= 8),  W = DZE + SC = nd1 + 116,
                //   pe' = DZE - SC = pe - 116,  nd2B = (dce - 5) - pe' = nd2 + 111,  NW = qe + NSC = nd3 + 115.
                // NW is next step's W: nd1 of a step is nd3 of the previous one minus 1, so only nd3 (and the first nd1 of
                // a run) needs a range test:  nd3 in [-114, -4]  <=>  NW in [1, 111];  nd2 <= 0  <=>  nd2B <= 111;
                // the larger part of q non-zero and normal  <=>  fmax >= 1  (fmax = 255 needs an infinite input, which
                // the bounded mantissas of this loop cannot produce: |p| < 32).
                // The orbit entry {re, im, s} is fetched with one 12-byte load in the scalar-base + per-lane-offset
                // addressing mode (issued by hand: the compiler folds the offset into a 64-bit per-lane pointer and then
                // spends a vector instruction per step on advancing it); the wait is tied to the loaded registers.
#define FS_QUIET_STEP(DZM, DZE, ZCM, SC, W, NDZM, NDZE, NZCM, NSC, NW, K, VIOL)                                     \
    f3 ent_##VIOL;                                                                                                  \
    {                                                                                                               \
        const float4 *zc_ = zq + (K);                                                                               \
        /* "+v"(DZE): nothing is written, it only pins the load ahead of the arithmetic that reads DZE */           \
        asm volatile("global_load_dwordx3 %0, %2, %3" : "=v"(ent_##VIOL), "+v"(DZE) : "v"(lane_off), "s"(zc_));     \
    }                                                                                                               \
    const int pe_##VIOL = DZE - SC; /* pe - 116; no clamp at kMinBigExp: an exponent that far down fails the W test */ \
    const f2 cur_##VIOL = ZCM + DZM * __int_as_float((W << 23) + (0x3F800000 - (116 << 23)));                       \
    const f2 pa_##VIOL = DZM.xx * cur_##VIOL;                                                                       \
    const f2 pb_##VIOL = DZM.yy * cur_##VIOL.yx;                                                                    \
    f2 p_##VIOL;                                                                                                    \
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(p_##VIOL) : "v"(pa_##VIOL), "v"(pb_##VIOL));     \
    const int nd2_##VIOL = dceB - pe_##VIOL; /* nd2 + 111 */                                                        \
    /* dc * 2^nd2 for nd2 > -120, 0 otherwise, as (dc * 2^7) * 2^(nd2 - 7): the clamped exponent field is 0 exactly at \
       the cut-off, and both factors stay normal */                                                                 \
    const float m2_##VIOL = __int_as_float((imax(imin(nd2_##VIOL, 111), -9) << 23) + (9 << 23));                    \
    const f2 q_##VIOL = p_##VIOL + dcm128 * m2_##VIOL;                                                              \
    const int fmax_##VIOL =                                                                                         \
        __float_as_int(fs_max_abs(q_##VIOL.x, q_##VIOL.y)) >> 23;            \
    NDZE = pe_##VIOL + fmax_##VIOL - 11; /* pe + fmax - 127 */                                                      \
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ent_##VIOL));                                                          \
    NSC = __float_as_int(ent_##VIOL.z);                                                                             \
    NZCM = (f2){ent_##VIOL.x, ent_##VIOL.y};                                                                        \
    NW = NDZE + NSC;                                                                                                \
    const uint64_t VIOL = __builtin_amdgcn_ballot_w64(imax(NW, nd2_##VIOL) > 111) |                                 \
                          __builtin_amdgcn_ballot_w64(imin(NW, fmax_##VIOL) < 1);                                   \
    NDZM = q_##VIOL * __int_as_float(mad24_scale(fmax_##VIOL)) /* 2^(127 - fmax) */
                if (!sc_stopped)
                {
                    typedef float f3 __attribute__((ext_vector_type(3)));
                    const float4 zq0 = zq[ref];
                    f2 Zcm = {zq0.x, zq0.y};
                    int sC = __float_as_int(zq0.z);
                    int W = dze + sC;
                    // entry of step k of this run = zq[done_k] + lane_off: a wave-uniform base advanced on the scalar unit
                    // plus a per-lane byte offset that is fixed for the whole run
                    const uint32_t lane_off = (ref + 1) * 16u;
                    FS_LOAD_DC()
                    const f2 dcm128 = dcm * 128.0f;
                    const int dceB = dce - 5;
                    uint32_t done = 0;
                    // first step of the run: nd1 in [-115, 0] is not implied by a previous quiet step
                    bool stop = __builtin_amdgcn_ballot_w64((unsigned)(W - 1) > 115u) != 0ull;
                    bool retry_scaled = false;
                    while (!stop) {
                        const uint32_t r0 = ref + done, i0 = iterations + done;
                        const uint32_t left_ref = r0 + 1 < MaxRefIteration ? MaxRefIteration - 1 - r0 : 0u;
                        const uint32_t left_it = n_iterations - 1 - i0; // running => iterations < n_iterations
                        uint32_t left = left_ref < left_it ? left_ref : left_it;
                        if (__builtin_amdgcn_ballot_w64(left < 64u) == 0ull) {
                            // every running lane has at least 64 quiet-eligible steps ahead: no per-step counter; two
                            // steps per trip so that the state ping-pongs between two register sets without copies
                            uint32_t c = 0;
                            for (; c < 64u; c += 2) {
                                f2 dzmB, ZcmB;
                                int dzeB, sB, WB;
                                FS_QUIET_STEP(dzm, dze, Zcm, sC, W, dzmB, dzeB, ZcmB, sB, WB, done + c, vA);
                                if (vA != 0ull) {
                                    stop = true;
                                    break;
                                }
                                f2 dzmA, ZcmA;
                                int dzeA, sA, WA;
                                FS_QUIET_STEP(dzmB, dzeB, ZcmB, sB, WB, dzmA, dzeA, ZcmA, sA, WA, done + c + 1, vB);
                                if (vB != 0ull) {
                                    dzm = dzmB, dze = dzeB, Zcm = ZcmB, sC = sB, W = WB;
                                    c++;
                                    stop = true;
                                    break;
                                }
                                dzm = dzmA, dze = dzeA, Zcm = ZcmA, sC = sA, W = WA;
                            }
                            done += c;
                            if (kScaled && !stop) {
                                // a clean chunk: hand the state back so that a scaled run can start from it (this loop is
                                // the second chance, and once in it a wave would otherwise stay for as long as it is quiet)
                                retry_scaled = true;
                                break;
                            }
                        } else {
                            for (;;) {
                                f2 dzmN, ZcmN;
                                int dzeN, sN, WN;
                                FS_QUIET_STEP(dzm, dze, Zcm, sC, W, dzmN, dzeN, ZcmN, sN, WN, done, vT);
                                if ((vT | __builtin_amdgcn_ballot_w64(left == 0u)) != 0ull)
                                    break;
                                dzm = dzmN, dze = dzeN, Zcm = ZcmN, sC = sN, W = WN;
                                left--;
                                done++;
                            }
                            stop = true;
                        }
                    }
                    ref += done;
                    iterations += done;
                    if (kStats)
                        c_pt += done;
                    if (retry_scaled)
                        continue; // (every lane of the chunk had >= 64 steps left: still running)
                }
#undef FS_QUIET_STEP
                // ---- one careful step: full exit tests (Fractal.cpp:2646-2661); the literal transcription takes over when a
                // value leaves the range the straight-line form is proven for
                const float4 zcur = zr[ref];
                const f2 Zcm = {zcur.x, zcur.y};
                const int Zce1 = __float_as_int(zcur.z) + 1; // the true exponent of 2 Zc (sC may be the poison value)
                const float4 zv = zr[ref + 1];
                const f2 Znm = {zv.x, zv.y};
                const int Zne = __float_as_int(zv.z);
                // cur = 2Z + dz
                const int e_cur = imax(Zce1, dze);
                const f2 cur = Zcm * pow2_or_zero(Zce1 - e_cur) + dzm * pow2_or_zero(dze - e_cur);
                // p = dz * cur       (re = dr*cr - di*ci, im = dr*ci + di*cr)
                const f2 pa = dzm.xx * cur;
                const f2 pb = dzm.yy * cur.yx;
                f2 p;
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(p) : "v"(pa), "v"(pb));
                const int pe = imax(dze + e_cur, kMinBigExp);
                // q = p + dc, then Reduce (the larger part a non-zero float: checked below)
                FS_LOAD_DC()
                const int e_q = imax(pe, dce);
                f2 q = p * pow2_or_zero(pe - e_q) + dcm * pow2_or_zero(dce - e_q);
                const int fmax = imax((int)__builtin_amdgcn_ubfe(__float_as_int(q.x), 23, 8),
                                      (int)__builtin_amdgcn_ubfe(__float_as_int(q.y), 23, 8));
                q = q * __int_as_float(0x7F000000 - (fmax << 23));
                const int qe = e_q + fmax - 127;
                // z = Z' + q; z is NOT reduced (see header comment)
                const int e_z = imax(Zne, qe);
                const int gq = qe - e_z;
                const f2 zm = Znm * pow2_or_zero(Zne - e_z) + q * pow2_or_zero(gq);
                const f2 zz = zm * zm;
                const float zn2 = zz.x + zz.y;
                const f2 qq = q * q;
                const float dn2 = qq.x + qq.y; // in [1,8): q's larger part is in [1,2)
                // With both norms positive normal floats, Reduce(|z|^2) > 256 and Reduce(|z|^2) < Reduce(|dz|^2)
                // (lexicographic on (exp, mantissa in [1,2))) are plain value comparisons:
                //   zn2 * 2^(2 e_z) > 2^8             <=>  zn2 > 2^(8 - 2 e_z)          (exact power of two, +inf / 0 beyond the range)
                //   zn2 * 2^(2 e_z) < dn2 * 2^(2 qe)  <=>  zn2 < dn2 * 2^(2 (qe - e_z))  (exact scaling; an underflow can
                //                                                                       only make the rhs <= min normal <= zn2)
                const int esc_e = 8 - 2 * (e_z < -100 ? -100 : (e_z > 100 ? 100 : e_z));
                bool escaped = zn2 > __builtin_amdgcn_ldexpf(1.0f, esc_e);
                bool rebase = zn2 < __builtin_amdgcn_ldexpf(dn2, gq + gq);
                // larger part of q: non-zero, finite, normal (a NaN or an infinity anywhere above ends up in q or zn2)
                const bool ok = (unsigned)(fmax - 1) < 254u && __builtin_amdgcn_classf(zn2, 0x100 /* +normal */);
                hcplx32 z;
                bool reduced_z = false;
                if (kStats)
                    c_pass++;
                if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {
                    if (kStats)
                        c_generic++;
                    // ---- generic step, literal order of Fractal.cpp:2646-2661
                    const hcplx32 Zc_g{Zcm.x, Zcm.y, Zce1 - 1};
                    const hcplx32 dz_g{dzm.x, dzm.y, dze};
                    hcplx32 curg = hc_mul2(Zc_g);
                    curg = hc_add(curg, dz_g);
                    hcplx32 ndz = hc_mul(dz_g, curg);
                    ndz = hc_add(ndz, hcplx32{dcm.x, dcm.y, dce});
                    hc_reduce(ndz);
                    z = hc_add(hcplx32{Znm.x, Znm.y, Zne}, ndz);
                    hc_reduce(z);
                    const hreal32 n = hr_reduced(hc_norm2(z));
                    const hreal32 dn = hr_reduced(hc_norm2(ndz));
                    escaped = hr_cmp_pos(n, hreal32{1.0f, 8}) > 0;
                    rebase = hr_cmp_pos(n, dn) < 0;
                    q = (f2){ndz.re, ndz.im};
                    dze = ndz.e;
                    reduced_z = true;
                } else {
                    z = hcplx32{zm.x, zm.y, e_z};
                    dze = qe;
                }
                if (kStats) {
                    c_pt++;
                    c_careful++;
                    // what a careful pass of the wave finds (tools/scaled_share_probe.py): a rebase / an escape in some lane,
                    // a rebase in every running lane
                    const uint64_t act = __builtin_amdgcn_ballot_w64(true);
                    const uint64_t rb = __builtin_amdgcn_ballot_w64(!escaped && (rebase || ref + 1 >= MaxRefIteration));
                    if (rb != 0ull)
                        c_why[1]++;
                    if (rb == act)
                        c_why[2]++;
                    if (__builtin_amdgcn_ballot_w64(escaped) != 0ull)
                        c_why[3]++;
                    // the entry this pass arrives at is one no scaled step may arrive at ("never" bound: near zero, or out of
                    // the companion's range) for every lane / and nothing happens there / nothing happens at another kind of
                    // entry / the pass is a back-off wait
                    const bool nz = __builtin_amdgcn_ballot_w64(__float_as_int(zs[ref + 1].z) != (int)0x80000000) == 0ull;
                    const bool quiet_pass = rb == 0ull && __builtin_amdgcn_ballot_w64(escaped) == 0ull;
                    if (nz)
                        c_nz[0]++;
                    if (nz && quiet_pass)
                        c_nz[1]++;
                    if (!nz && quiet_pass)
                        c_nz[2]++;
                    if (was_skip)
                        c_nz[3]++;
                }
                was_skip = false;
                ref++;
                dzm = q;
                if (escaped) {
                    running = false; // `break` happens before iterations++ in the CPU loop
                } else {
                    if (rebase || ref >= MaxRefIteration) {
                        if (!reduced_z)
                            hc_reduce(z);
                        dzm = (f2){z.re, z.im};
                        dze = z.e;
                        ref = 0;
                    }
                    iterations++;
                    running = iterations < n_iterations;
                }
            }
            FS_CYC(cyc_loop += __builtin_readcyclecounter() - cyc_t0);
            FS_CYC(wall_loop += wall_clock64() - wall_t0);
            {
                uint32_t lane_e;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
                lane_cost = iterations - s_it0[wave_in_block * 64u + lane_e];
            }
        } else {
            lane_cost = la_cost;
        }
        {
            // (the pixel again, see the top of the kernel)
            uint32_t lane_e;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
            store_iter(A.out, A.frame, tile_y * 8u + (lane_e >> 3), tile_x * 8u + (lane_e & 7u), iterations);
        }
    }
    if (A.tile_cost && tile_x < A.tiles_x) {
        // the tile's cost = its longest lane (the wave runs until that one is done)
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_down(lane_cost, off);
            lane_cost = o > lane_cost ? o : lane_cost;
        }
        uint32_t lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        if (lane_e == 0u)
            A.tile_cost[tile_y * A.tiles_x + tile_x] = lane_cost;
    }
#ifdef FS_TRACE_WAVES
    if (kStats && A.stats) {
        uint64_t steps = c_pt;
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t o = __shfl_down(steps, off);
            steps = o > steps ? o : steps;
        }
        if ((threadIdx.x & 63) == 0) {
            uint32_t hw_id, xcc_id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
            const uint64_t wave = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
            uint64_t *t = A.stats + 16 + 4 * wave;
            t[0] = trace_t0;
            t[1] = wall_clock64();
            t[2] = ((uint64_t)xcc_id << 32) | hw_id;
            t[3] = steps; // longest lane of the wave, perturbation steps
        }
    }
#endif
#ifdef FS_PROFILE_CYCLES
    if (kStats) {
        if ((threadIdx.x & 63) == 0) {
            atomicAdd((unsigned long long *)&A.stats[0], (unsigned long long)cyc_loop);
            atomicAdd((unsigned long long *)&A.stats[1], (unsigned long long)cyc_run);
            atomicAdd((unsigned long long *)&A.stats[3], (unsigned long long)cyc_body);
            atomicAdd((unsigned long long *)&A.stats[24], (unsigned long long)cyc_asm);
            atomicAdd((unsigned long long *)&A.stats[25], (unsigned long long)cyc_tested);
            atomicAdd((unsigned long long *)&A.stats[26], (unsigned long long)cyc_hot);
            atomicAdd((unsigned long long *)&A.stats[27], (unsigned long long)wall_loop); // 100 MHz ticks
        }
        c_at = c_la = c_px = 0;
    }
#endif
    if (kStats) {
        add_stats(A.stats, c_at, c_la, c_pt, c_px);
        // stats[5]: lane-steps taken through the careful path (the rest of [2] ran in quiet runs)
        for (int off = 32; off > 0; off >>= 1) {
            c_careful += __shfl_down(c_careful, off);
            c_scaled += __shfl_down(c_scaled, off);
            c_runs += __shfl_down(c_runs, off);
            const uint32_t bf = __shfl_down(c_blk_free, off), bt = __shfl_down(c_blk_tested, off);
            c_blk_free = bf > c_blk_free ? bf : c_blk_free; // wave-uniform while a lane is in the loop: the longest lane's
            c_blk_tested = bt > c_blk_tested ? bt : c_blk_tested;
            const uint32_t cp = __shfl_down(c_pass, off), cg = __shfl_down(c_generic, off);
            c_pass = cp > c_pass ? cp : c_pass;
            c_generic = cg > c_generic ? cg : c_generic;
            const uint32_t we = __shfl_down(c_wentry, off), ws = __shfl_down(c_wstart, off), wh = __shfl_down(c_wshort, off);
            c_wentry = we > c_wentry ? we : c_wentry;
            c_wstart = ws > c_wstart ? ws : c_wstart;
            c_wshort = wh > c_wshort ? wh : c_wshort;
            const uint32_t ls = __shfl_down(c_lane_steps, off), lr = __shfl_down(c_lane_runs, off);
            c_lane_steps = ls > c_lane_steps ? ls : c_lane_steps;
            c_lane_runs = lr > c_lane_runs ? lr : c_lane_runs;
            const uint32_t bv = __shfl_down(c_blk_violation, off);
            c_blk_violation = bv > c_blk_violation ? bv : c_blk_violation;
            for (int i = 0; i < 4; i++) {
                const uint32_t y = __shfl_down(c_why[i], off);
                c_why[i] = y > c_why[i] ? y : c_why[i];
                const uint32_t y2 = __shfl_down(c_nz[i], off);
                c_nz[i] = y2 > c_nz[i] ? y2 : c_nz[i];
            }
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd((unsigned long long *)&A.stats[5], (unsigned long long)c_careful);
            atomicAdd((unsigned long long *)&A.stats[6], (unsigned long long)c_scaled);
            atomicAdd((unsigned long long *)&A.stats[7], (unsigned long long)c_runs);
            atomicAdd((unsigned long long *)&A.stats[8], (unsigned long long)c_blk_free);
            atomicAdd((unsigned long long *)&A.stats[9], (unsigned long long)c_blk_tested);
            atomicAdd((unsigned long long *)&A.stats[10], (unsigned long long)c_pass);
            atomicAdd((unsigned long long *)&A.stats[11], (unsigned long long)c_generic);
            atomicAdd((unsigned long long *)&A.stats[12], (unsigned long long)c_wentry);
            atomicAdd((unsigned long long *)&A.stats[13], (unsigned long long)c_wstart);
            atomicAdd((unsigned long long *)&A.stats[14], (unsigned long long)c_wshort);
            atomicAdd((unsigned long long *)&A.stats[15], (unsigned long long)c_blk_violation);
            for (int i = 0; i < 4; i++) {
                atomicAdd((unsigned long long *)&A.stats[16 + i], (unsigned long long)c_why[i]);
                atomicAdd((unsigned long long *)&A.stats[20 + i], (unsigned long long)c_nz[i]);
            }
            atomicAdd((unsigned long long *)&A.stats[28], (unsigned long long)c_lane_steps);
            atomicAdd((unsigned long long *)&A.stats[29], (unsigned long long)c_lane_runs);
        }
    }
}
#undef FS_CYC

// ------------------------------------------------------------------------------------------------
// Scalar-HDRFloat perturbation with optional BLA skipping, T = HDRFloat<float>.
// CPU twin: Fractal::CalcCpuPerturbationFractalBLA<uint32_t,HDRFloat<float>,float> (Fractal.cpp:2266-2470),
// BLAS::LookupBackwards (BLAS.cpp:256-310), BLA::getValue (BLA.cuh:21-38).  With kBla == false the lookup is
// compiled out: that is the perturbation-only single-step branch (:2342-2466), the parity target of the
// LAv2Mode::PO entry point.  Replaces mandel_1xHDR_float_perturb_bla
// (FractalSharkGpuLib/BLAKernels.cuh:193-434) and the PO instantiation of the LAv2 kernel.
namespace {

// BLAS::LookupBackwards (BLAS.cpp:256-310).  `levels` is the workgroup's LDS copy of the level pointer table (a
// ds_read instead of a global load in front of every probe).  The (level, index) pairs a lookup visits depend only on m,
// so the r2 values of the first four levels are requested together -- one memory round trip instead of up to four
// dependent ones -- and then tested in the reference's order (highest level first).
template <class F>
__device__ __forceinline__ const typename FsDev<F>::BLA *bla_lookup(const typename FsDev<F>::BLA *const *levels, int32_t lm2,
                                                                     uint32_t m, hreal<F> z2)
{
    using B = typename FsDev<F>::BLA;
    if (m == 0)
        return nullptr;
    const int32_t k = (int32_t)m - 1;
    if ((k & 1) == 1)
        return nullptr;
    int32_t zeros;
    uint32_t ix;
    if (k == 0) {
        if (hr_cmp_pos(z2, ldr(levels[2][0].r2)) >= 0)
            return nullptr;
        zeros = 32;
        ix = 0;
    } else {
        zeros = __ffs(k) - 1; // exponent of (float)(k & -k), BLAS.cpp:283-286
        ix = (uint32_t)k >> zeros;
    }
    const int32_t startLevel = zeros <= lm2 ? zeros : lm2;
    if (startLevel < 2)
        return nullptr;
    const int32_t np = startLevel - 1 < 4 ? startLevel - 1 : 4; // levels startLevel .. startLevel - np + 1 (>= 2)
    const B *t0 = levels[startLevel] + ix, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;
    hreal<F> r0 = ldr(t0->r2), r1 = r0, r2 = r0, r3 = r0;
    if (np > 1) {
        t1 = levels[startLevel - 1] + (ix << 1);
        r1 = ldr(t1->r2);
    }
    if (np > 2) {
        t2 = levels[startLevel - 2] + (ix << 2);
        r2 = ldr(t2->r2);
    }
    if (np > 3) {
        t3 = levels[startLevel - 3] + (ix << 3);
        r3 = ldr(t3->r2);
    }
    if (hr_cmp_pos(z2, r0) < 0)
        return t0;
    if (np > 1 && hr_cmp_pos(z2, r1) < 0)
        return t1;
    if (np > 2 && hr_cmp_pos(z2, r2) < 0)
        return t2;
    if (np > 3 && hr_cmp_pos(z2, r3) < 0)
        return t3;
    ix <<= 4;
    for (int32_t level = startLevel - 4; level >= 2; --level) {
        const B *t = &levels[level][ix];
        if (hr_cmp_pos(z2, ldr(t->r2)) < 0)
            return t;
        ix <<= 1;
    }
    return nullptr;
}

// The same lookup on the device-native table (FsBlaRec / ladder, kernels.h): returns the POSITION of the record that applies,
// or ~0u.  One round = the four probes BLAS::LookupBackwards would make next, fetched as two 16-byte loads from one ladder
// entry and decided with four signed 64-bit compares (key = exponent << 32 | mantissa bits == the reference's
// lexicographic compare for reduced non-negative values); the first probe that holds, in the reference's order (highest
// level first), wins -- no assumption about the r2 being monotone along the ladder.  Levels below 2 carry keys that never
// hold.  No branch per probe, no 64-bit pointer per level: level offsets come from LDS (`off`), positions are 32-bit.
// 15 of 16 lookups start at level <= 5 and finish in their first round; deeper ones loop (another four levels per round).
// A table entry can only apply at orbit indices m = 1 (mod 4) (level >= 2 needs k = m - 1 divisible by 4): when no lane
// of the wave sits at one, the lookup is one vote.
// Round 4: before the walk, one key per orbit index -- kmax[(m - 1) / 4] = the largest key the walk at m can meet
// (k_bla_make_kmax) -- decides the lookups that find nothing, which is how every outer trip of the kernel ends (741 of a
// wave's 1061 lookup passes on C5): one 8-byte load and one compare instead of two to three rounds of the ladder.
__device__ __forceinline__ uint32_t bla_lookup_native(const int4 *__restrict__ lad, const long long *__restrict__ kmax,
                                                      const uint32_t *off, int32_t lm2, uint32_t m, long long zkey,
                                                      long long key20)
{
    if (__builtin_amdgcn_ballot_w64((m & 3u) == 1u) == 0ull)
        return 0xFFFFFFFFu;
    const int32_t k = (int32_t)m - 1;
    const bool first = k == 0;
    const int32_t zeros = first ? 32 : (int32_t)__ffs(k) - 1; // exponent of (float)(k & -k), BLAS.cpp:283-286
    uint32_t ix = first ? 0u : (uint32_t)k >> (zeros & 31);
    int32_t L = zeros <= lm2 ? zeros : lm2;
    // m == 0: no table entry; odd k: level 0; k == 0: only when the first element of level 2 applies (BLAS.cpp:270-281)
    bool live = m != 0u && (k & 1) == 0 && L >= 2 && (!first || zkey < key20);
    if (live)
        live = zkey < kmax[(uint32_t)k >> 2];
    uint32_t hit = 0xFFFFFFFFu;
    while (__builtin_amdgcn_ballot_w64(live) != 0ull) {
        if (live) {
            const uint32_t p = off[L] + ix;
            const int4 a = lad[2u * (size_t)p], b = lad[2u * (size_t)p + 1u];
            const long long k0 = (long long)(((unsigned long long)(unsigned)a.y << 32) | (unsigned)a.x);
            const long long k1 = (long long)(((unsigned long long)(unsigned)a.w << 32) | (unsigned)a.z);
            const long long k2 = (long long)(((unsigned long long)(unsigned)b.y << 32) | (unsigned)b.x);
            const long long k3 = (long long)(((unsigned long long)(unsigned)b.w << 32) | (unsigned)b.z);
            int32_t nf = zkey < k3 ? 3 : 4;
            nf = zkey < k2 ? 2 : nf;
            nf = zkey < k1 ? 1 : nf;
            nf = zkey < k0 ? 0 : nf;
            if (nf < 4) {
                hit = off[L - nf] + (ix << nf);
                live = false;
            } else {
                L -= 4;
                ix <<= 4;
                live = L >= 2;
            }
        }
    }
    return hit;
}

} // namespace

// With a table (kBla) the kernel is PERSISTENT and lanes are re-packed: pixels of one wave finish at very different
// times (BLA jumps and rebases make iteration counts of neighbours differ by orders of magnitude; a third of the lane
// slots of a one-tile-per-wave launch idle behind the longest pixel of their tile), and nothing in this loop needs the
// lanes of a wave to be neighbours -- with a table they sit at different orbit positions after the first jump anyway.
// So a wave keeps its 64 lanes fed from a frame-wide pixel queue: every kRefillEvery outer iterations the idle lanes
// are found with one ballot, the wave takes popcount(idle) consecutive pixel numbers with ONE atomic (lane prefix =
// mbcnt over the ballot) and the idle lanes start those pixels.  Pixel numbers run in 8 x 8-tile order, so a wave
// starts on one tile like the non-persistent launch.  Without a table (perturbation only) the scaled runs want all
// lanes of a wave at the same orbit position (scalar-cache entries), so that launch stays one tile per wave.
constexpr uint32_t kRefillEvery = 24;

// kNat (HDRFloat<float>, kBla, one tile per wave): the table is read in its device-native form (FsBlaRec + ladder).
// IterT: the reference's IterType for the counters (BLAKernels.cuh:193 is templated on it the same way): uint32_t, or
// uint64_t for iteration caps of 2^32 and above (one tile per wave, reference-layout lookup; the runs of the perturbation-only
// float path, whose step budgets are 32-bit, are compiled out -- the single steps and the jumps count in IterT).
template <class F, bool kBla, bool kStats, bool kRefill, bool kNat = false, class IterT = uint32_t>
__global__ void __launch_bounds__(256) k_perturb_scalar(FsBlaArgsT<F> A)
{
    constexpr bool kRuns = sizeof(IterT) == 4;
#ifdef FS_TRACE_WAVES
    const uint64_t ps_trace_t0 = wall_clock64();
    const uint64_t ps_trace_c0 = __builtin_readcyclecounter(); // shader clock: with the constant 100 MHz clock, the wave's MHz
#endif
    __shared__ const typename FsDev<F>::BLA *s_levels[kBla && !kNat ? 64 : 1];
    __shared__ uint32_t s_off[kNat ? 64 : 1];
    if constexpr (kBla && !kNat) {
        if (threadIdx.x < 64u)
            s_levels[threadIdx.x] = (int32_t)threadIdx.x < A.lm2 + 2 ? A.levels[threadIdx.x] : nullptr;
        __syncthreads();
    }
    long long nat_key20 = 0;
    if constexpr (kNat) {
        if (threadIdx.x < 64u)
            s_off[threadIdx.x] = threadIdx.x < (uint32_t)kBlaMaxLevels ? A.level_off[threadIdx.x] : 0u;
        __syncthreads();
        // key of the first element of level 2 (the k == 0 pre-test): wave-uniform, one scalar load
        const int4 e20 = A.nlad[2u * (size_t)A.level_off[2]];
        nat_key20 = (long long)(((unsigned long long)(unsigned)e20.y << 32) | (unsigned)e20.x);
    }
    uint32_t X = 0, L = 0;
    uint64_t c_la = 0, c_pt = 0, c_px = 0;
    uint64_t c_single = 0, c_runs = 0; // probes of the perturbation-only float path (tools/c2_probe.py)
    uint64_t c_blk_violation = 0; // (verification build) must stay 0
    uint64_t c_free_steps = 0, c_tested_blocks = 0; // lane-steps inside the untested loop / tested four-step blocks (per lane)
    uint32_t c_end[5] = {0, 0, 0, 0, 0}; // (counting build, per wave) why scaled runs of the perturbation-only path end: their length / H / a tested block failed / floor (status 3) -- and [4] roll-backs of a block test (status 4: not an end)
    // (kBla probes, statistics words 8..12: lane-passes through the quiet step / the step with z / the literal step, the
    // quiet jump / the jump with z -- which share of the actions the hand-written kernel's fast forms must cover)
    uint64_t c_q_step = 0, c_z_step = 0, c_lit_step = 0, c_q_jump = 0, c_z_jump = 0;
#ifdef FS_PROFILE_CYCLES
    // measurement build (tools/c5_phase_probe.py): shader-clock cycles and wave-passes per phase of the BLA loop, per wave.
    // The clock is read on the scalar unit, i.e. once per pass of the WAVE through the code, whatever the lane mask is.
    uint64_t ph_lookup = 0, ph_jump = 0, ph_step = 0, ph_literal = 0, ph_t = 0;
    uint64_t ph_n_lookup = 0, ph_n_jump = 0, ph_n_step = 0, ph_n_literal = 0, ph_n_outer = 0;
    uint64_t ph_lanes_jump = 0, ph_lanes_step = 0, ph_n_scaled = 0; // (ph_n_scaled: step passes taken by a cheap form)
#define FS_PH(stmt) do { if (kStats && kBla) { stmt; } } while (0)
    // ... and of the perturbation-only float path (tools/c2_phase_probe.py): the whole pixel loop, the scaled-run block, the
    // hand-scheduled statement inside it, the exponent-tracking (second chance) block; the single steps are the rest
    uint64_t po_total = 0, po_run = 0, po_asm = 0, po_quiet = 0, po_t0 = 0, po_t1 = 0, po_t2 = 0, po_n_run = 0, po_n_asm = 0;
#define FS_PO(stmt) do { if (kStats && !kBla) { stmt; } } while (0)
#else
#define FS_PH(stmt) do { } while (0)
#define FS_PO(stmt) do { } while (0)
#endif
    FS_PO(po_t0 = __builtin_readcyclecounter());
    const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
    const uint32_t count = A.orbit_count;
    const typename FsDev<F>::Z *__restrict__ zr = A.zref;
    const hreal<F> TwoFiftySix = hreal<F>{F(1), 8};
    // per-pixel state (lives across refill rounds)
    bool have = false;
    IterT iter = 0;
    uint32_t RefIteration = 0;
    hreal<F> DeltaSub0X = hr_zero<F>(), DeltaSub0Y = hr_zero<F>();
    hreal<F> DeltaSubNX = hr_zero<F>();
    hreal<F> DeltaSubNY = hr_zero<F>();
    hreal<F> DeltaNormSquared = hr_zero<F>();
    hcplx<F> Zcached = hc_zero<F>();
    uint32_t Zcached_at = 0xFFFFFFFFu;
    bool force_step = false; // (action loop) the BLA loop of the reference was left by its escape test: step next, no lookup
    // Exact cycle detection for pixels that never escape (perturbation only).  After a rebase the whole future of a pixel
    // is a pure function of its dz (RefIteration is 0, dc is fixed, the arithmetic is deterministic; the iteration counter
    // only decides where the loop stops).  So if the dz of a rebase equals, bit for bit, the dz of an earlier rebase, the
    // pixel's states repeat forever, none of them escaped, and the reference's loop would run on to the iteration cap and
    // return exactly n_iterations -- which is returned here at once.  Brent's scheme: the dz of rebase number 1, 2, 4, 8 ...
    // is kept, every later rebase compares against it (four integer compares on a path taken once per ~60 steps).
    // Interior pixels of C2 run 4.7 M steps each in the reference; their dz locks into an exact cycle long before that.
    hreal<F> cycX = hreal<F>{F(0), INT32_MIN}, cycY = cycX;
    uint32_t cyc_n = 0, cyc_next = 1;
#define FS_CYCLE_CHECK()                                                                                            \
    if constexpr (!kBla && std::is_same<F, float>::value) {                                                         \
        if (__float_as_int(DeltaSubNX.m) == __float_as_int(cycX.m) && DeltaSubNX.e == cycX.e &&                     \
            __float_as_int(DeltaSubNY.m) == __float_as_int(cycY.m) && DeltaSubNY.e == cycY.e) {                     \
            iter = n_iterations - 1u; /* the ++iter that follows makes it the cap */                                \
            if (kStats)                                                                                             \
                atomicAdd((unsigned long long *)&A.stats[5], 1ull); /* probe: pixels ended by the cycle test */      \
        } else if (++cyc_n == cyc_next) {                                                                           \
            cycX = DeltaSubNX, cycY = DeltaSubNY;                                                                   \
            cyc_next <<= 1;                                                                                         \
        }                                                                                                           \
    }
    // frame-wide pixel queue (kRefill)
    const uint32_t tiles_x = (A.frame.width + 7u) >> 3;
    const uint32_t total = tiles_x * ((A.frame.local_rows + 7u) >> 3) * 64u;
    bool queue_empty = false;
    auto start_pixel = [&](uint32_t x, uint32_t l) {
        X = x, L = l;
        if (A.probe_out) { // probe launch: the centre pixel of tile (x, l)
            x = x < 0x10000000u ? (x << 3) + 4u : 0xFFFFFFFFu;
            l = (l << 3) + 4u;
        }
        const uint32_t Y = global_row(A.frame, l);
        have = x < A.frame.width && l < A.frame.local_rows && Y < A.frame.height;
        if (have) {
            c_px++;
            iter = 0;
            RefIteration = 0;
            pixel_delta<F>(A.coords, x, Y, DeltaSub0X, DeltaSub0Y);
            DeltaSubNX = hr_zero<F>();
            DeltaSubNY = hr_zero<F>();
            DeltaNormSquared = hr_zero<F>();
            Zcached = hc_zero<F>();
            Zcached_at = 0xFFFFFFFFu;
            force_step = false;
            cycX = hreal<F>{F(0), INT32_MIN}, cycY = cycX;
            cyc_n = 0, cyc_next = 1;
        }
    };
    if constexpr (!kRefill) {
        uint32_t x, l;
        if (A.tile_order) {
            const uint32_t w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
            const uint32_t t = A.tile_order[w], lane = threadIdx.x & 63u;
            x = t != 0xFFFFFFFFu ? (t % tiles_x) * 8u + (lane & 7u) : 0xFFFFFFFFu;
            l = (t / tiles_x) * 8u + (lane >> 3);
            // the waves of the long tiles decide when the frame ends: they ask the instruction arbiter for priority over
            // the waves they share their SIMD with while the bulk of the frame is still being rendered
            const uint32_t nl = A.tile_order[gridDim.x * gridDim.y * (blockDim.x >> 6)];
            if ((nl >> 31) != 0u ? ((w & 3u) == 0u && (w >> 2) < (nl & 0x7FFFFFFFu)) : w < nl)
                __builtin_amdgcn_s_setprio(3);
        } else {
            tile_pixel(x, l);
        }
        start_pixel(x, l);
    }
    for (;;) {
        if constexpr (kRefill) {
            const uint64_t idle = __builtin_amdgcn_ballot_w64(!have);
            if (idle != 0ull && !queue_empty) {
                const uint32_t want = (uint32_t)__popcll(idle);
                uint32_t base = 0;
                if ((threadIdx.x & 63u) == 0u)
                    base = atomicAdd(A.queue, want);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                queue_empty = base + want >= total;
                // rank of this lane among the idle lanes = number of idle lanes below it
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t idx = base + rank;
                if (!have && idx < total) {
                    const uint32_t t = idx >> 6, ln = idx & 63u;
                    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
                    start_pixel(tx * 8u + (ln & 7u), ty * 8u + (ln >> 3));
                }
            }
            if (__builtin_amdgcn_ballot_w64(have) == 0ull) {
                if (queue_empty)
                    break;
                continue;
            }
        }
    if (have) {
        bool finished = true;
        uint32_t budget = kRefillEvery;
        while (iter < n_iterations) {
            if constexpr (kRefill) {
                // every lane counts the same trips (a SIMT loop runs its trips jointly), so the wave leaves together
                if (budget == 0u) {
                    finished = false;
                    break;
                }
                budget--;
            }
            // ---- kBla, float: the ACTION loop.  The reference's "while (a table entry applies) jump; then one
            // perturbation step" is, per pixel, a sequence of actions -- JUMP (BLA::getValue) or STEP -- that both end
            // in the same tail: z = Z[next] + dz', the two norms, the escape test, the rebase test.  Lanes of a wave are
            // rarely due for the same action, so each trip of this loop lets EVERY lane take ONE action: a short
            // divergent part that only forms the new dz (four aligned products for a jump, dz (2Z + dz) + dc for a
            // step, both in the alignment-free form described at the tuned single step below), then the shared
            // tail at full width.  (The first version ran a per-lane `while` of jumps followed by a wave-voted step:
            // lanes waited for each other's jump chains, 38 % of the vector lane-cycles did work.)  The order of
            // actions of every pixel is the reference's: a jump whose escape test fires leaves the reference's inner
            // loop, so that pixel's next action is a STEP without a lookup (force_step).
            bool act_literal_step = false;
            // Measured on C5: with one 8 x 8 tile per wave the lanes' actions are correlated and the
            // reference-shaped loop below (a per-lane chain of jumps, then a wave-voted step) is the faster one
            // (263 vs 302 ms); with lanes re-packed from the pixel queue (kRefill) actions are uncorrelated and this
            // loop is (333 vs 505 ms).  So the action loop is the persistent launch's loop.
            if constexpr (kBla && kRefill && std::is_same<F, float>::value) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                auto p2 = [](int n) { return n > -kExpDiffIgnored ? __int_as_float((n << 23) + 0x3F800000) : 0.0f; };
                const typename FsDev<F>::BLA *b =
                    force_step ? nullptr : bla_lookup<F>(s_levels, A.lm2, RefIteration, DeltaNormSquared);
                force_step = false;
                uint32_t l = 0;
                if (b != nullptr) {
                    l = (uint32_t)b->l;
                    if (RefIteration + l >= count || iter + l >= n_iterations)
                        b = nullptr; // the reference leaves its jump loop here and takes a step
                }
                const bool jump = b != nullptr;
                const uint32_t nref = RefIteration + (jump ? l : 1u);
                // Both actions are  dz' = A dz + B dc  with complex A, B given as four extended-exponent reals:
                //   JUMP: (Ax, Ay, Bx, By) = the table record (BLA::getValue, BLA.cuh:21-38);
                //   STEP: A = 2Z + dz (formed here under one exponent, as at the tuned single step), B = 1 -- the
                //         reference's (B1 - B2) + dcX and (C1 + C2) + dcY are the first three of the four terms below, in
                //         the same order, and the fourth is an exact zero.
                // So only the few instructions that produce (A, B) diverge; the four aligned products, the sums and the
                // tail run at full width for every lane.
                hreal<F> Ax, Ay, Bx, By;
                bool ok = true;
                if (jump) {
                    Ax = ldr(b->Ax), Ay = ldr(b->Ay), Bx = ldr(b->Bx), By = ldr(b->By);
                } else {
                    const hcplx<F> Z = (Zcached_at == RefIteration) ? Zcached : zref_at(zr, RefIteration);
                    const int Ze1 = Z.e + 1;
                    const int eT = imax(imax(Ze1, DeltaSubNX.e), DeltaSubNY.e);
                    const float zsT = p2(Ze1 - eT);
                    const f2 T = (f2){Z.re, Z.im} * (f2){zsT, zsT} +
                                 (f2){DeltaSubNX.m, DeltaSubNY.m} * (f2){p2(DeltaSubNX.e - eT), p2(DeltaSubNY.e - eT)};
                    Ax = hreal<F>{T.x, eT};
                    Ay = hreal<F>{T.y, eT};
                    Bx = hreal<F>{1.0f, 0};
                    By = hreal<F>{0.0f, -(1 << 25)}; // an exact zero that never sets the common exponent
                    const float tmx_ = fmaxf(fabsf(T.x), fabsf(T.y)), tmn_ = fminf(fabsf(T.x), fabsf(T.y));
                    ok = tmn_ >= 0x1p-60f && tmx_ <= 0x1p60f && nref < count;
                }
                f2 cm;        // new dz, mantissas
                int cex, cey; // ... and exponents (per part)
                {
                    const f2 D = {DeltaSubNX.m, DeltaSubNY.m}, D0 = {DeltaSub0X.m, DeltaSub0Y.m};
                    // nx = ((Ax DX - Ay DY) + Bx D0X) - By D0Y;  ny = ((Ax DY + Ay DX) + Bx D0Y) + By D0X
                    const f2 pA = (f2){Ax.m, Ax.m} * D, pB = (f2){Ay.m, Ay.m} * D.yx;
                    const f2 pC = (f2){Bx.m, Bx.m} * D0, pD = (f2){By.m, By.m} * D0.yx;
                    const int eAx = Ax.e + DeltaSubNX.e, eAy = Ax.e + DeltaSubNY.e;
                    const int eBx = Ay.e + DeltaSubNY.e, eBy = Ay.e + DeltaSubNX.e;
                    const int eCx = Bx.e + DeltaSub0X.e, eCy = Bx.e + DeltaSub0Y.e;
                    const int eDx = By.e + DeltaSub0Y.e, eDy = By.e + DeltaSub0X.e;
                    cex = imax(imax(eAx, eBx), imax(eCx, eDx)), cey = imax(imax(eAy, eBy), imax(eCy, eDy));
                    const f2 tA = pA * (f2){p2(eAx - cex), p2(eAy - cey)}, tB = pB * (f2){p2(eBx - cex), p2(eBy - cey)};
                    const f2 tC = pC * (f2){p2(eCx - cex), p2(eCy - cey)}, tD = pD * (f2){p2(eDx - cex), p2(eDy - cey)};
                    f2 s1;
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(s1) : "v"(tA), "v"(tB));
                    const f2 s2 = s1 + tC;
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(cm) : "v"(s2), "v"(tD));
                    const float zmn = fminf(fminf(fabsf(s1.x), fabsf(s1.y)), fminf(fabsf(s2.x), fabsf(s2.y)));
                    const int emin = imin(imin(imin(DeltaSubNX.e, DeltaSubNY.e), imin(DeltaSub0X.e, DeltaSub0Y.e)),
                                          imin(imin(Ax.e, Ay.e), imin(Bx.e, By.e)));
                    ok = ok && zmn > 0.0f && emin > -(1 << 26);
                    if (!jump) { // a step ends with Reduce of both parts (Fractal.cpp:2353,2360); a jump does not
                        const int qxb = __float_as_int(cm.x), qyb = __float_as_int(cm.y);
                        ok = ok && fminf(fabsf(cm.x), fabsf(cm.y)) >= 0x1p-60f && fmaxf(fabsf(cm.x), fabsf(cm.y)) <= 0x1p60f;
                        cm = (f2){__int_as_float((qxb & 0x807FFFFF) | 0x3F800000), __int_as_float((qyb & 0x807FFFFF) | 0x3F800000)};
                        cex += (int)__builtin_amdgcn_ubfe(qxb, 23, 8) - 127;
                        cey += (int)__builtin_amdgcn_ubfe(qyb, 23, 8) - 127;
                    }
                }
                // ---- shared tail: z = Z[nref] + dz' under ez, |z|^2, |dz'|^2
                const auto zn4 = zr[nref]; // in bounds: nref <= count (two spare entries)
                const int Zne = __float_as_int(zn4.z);
                const int ez = imax(imax(Zne, cex), cey);
                const float zsZ = p2(Zne - ez);
                const f2 Zt = (f2){zn4.x, zn4.y} * (f2){zsZ, zsZ} + cm * (f2){p2(cex - ez), p2(cey - ez)};
                const f2 ZZ = Zt * Zt;
                const float nm = ZZ.x + ZZ.y; // exponent 2 ez
                const f2 SQ = cm * cm;
                const int dd = (cex - cey) << 1;
                const bool sxbig = dd >= 0;
                const float md = p2(sxbig ? -dd : dd);
                const float dnm = SQ.x * (sxbig ? 1.0f : md) + SQ.y * (sxbig ? md : 1.0f);
                const int dne = (sxbig ? cex : cey) << 1;
                const float tmx = fmaxf(fmaxf(fabsf(cm.x), fabsf(cm.y)), fmaxf(fabsf(Zt.x), fabsf(Zt.y)));
                const float tmn = fminf(fminf(fabsf(cm.x), fabsf(cm.y)), fminf(fabsf(Zt.x), fabsf(Zt.y)));
                ok = ok && tmn >= 0x1p-60f && tmx <= 0x1p60f && Zne > -(1 << 26);
                if (ok) {
                    if (jump) {
                        iter += l;
                        if (kStats) {
                            c_la++;
                            if (l >= 1024u)
                                atomicAdd((unsigned long long *)&A.stats[6], 1ull);
                            if (l >= 256u)
                                atomicAdd((unsigned long long *)&A.stats[7], 1ull);
                        }
                    } else if (kStats) {
                        c_pt++;
                    }
                    RefIteration = nref;
                    Zcached = hcplx<F>{zn4.x, zn4.y, Zne};
                    Zcached_at = nref;
                    DeltaSubNX = hreal<F>{cm.x, cex};
                    DeltaSubNY = hreal<F>{cm.y, cey};
                    {
                        const int db = __float_as_int(dnm);
                        DeltaNormSquared = hreal<F>{__int_as_float((db & 0x007FFFFF) | 0x3F800000),
                                                    dne + (int)__builtin_amdgcn_ubfe(db, 23, 8) - 127};
                    }
                    if (__builtin_amdgcn_ldexpf(nm, imax((ez << 1) - 8, -400)) > 1.0f) {
                        if (jump) {
                            force_step = true; // the reference leaves its jump loop; the pixel's next action is a step
                            continue;
                        }
                        break; // a step escaped: the pixel is done (no ++iter, like the reference)
                    }
                    if (nm < __builtin_amdgcn_ldexpf(dnm, imax(dne - (ez << 1), -400)) || RefIteration >= count - 1) {
                        const int ex = imax(Zne, cex), ey = imax(Zne, cey);
                        DeltaSubNX = hreal<F>{__builtin_amdgcn_ldexpf(Zt.x, ez - ex), ex};
                        DeltaSubNY = hreal<F>{__builtin_amdgcn_ldexpf(Zt.y, ez - ey), ey};
                        const int nb = __float_as_int(nm);
                        DeltaNormSquared = hreal<F>{__int_as_float((nb & 0x007FFFFF) | 0x3F800000),
                                                    (ez << 1) + (int)__builtin_amdgcn_ubfe(nb, 23, 8) - 127};
                        RefIteration = 0;
                    }
                    if (!jump)
                        ++iter;
                    continue;
                }
                // ---- a sum left [2^-60, 2^60] or hit an exact zero: this lane's action in the literal order
                if (jump) {
                    iter += l;
                    if (kStats)
                        c_la++;
                    const hcplx<F> Z = zref_at(zr, nref);
                    {
                        const hreal<F> nx = hr_sub(
                            hr_add(hr_sub(hr_mul(Ax, DeltaSubNX), hr_mul(Ay, DeltaSubNY)), hr_mul(Bx, DeltaSub0X)),
                            hr_mul(By, DeltaSub0Y));
                        const hreal<F> ny = hr_add(
                            hr_add(hr_add(hr_mul(Ax, DeltaSubNY), hr_mul(Ay, DeltaSubNX)), hr_mul(Bx, DeltaSub0Y)),
                            hr_mul(By, DeltaSub0X));
                        DeltaSubNX = nx;
                        DeltaSubNY = ny;
                    }
                    RefIteration = nref;
                    const hreal<F> tempZX = hr_add(hc_re(Z), DeltaSubNX);
                    const hreal<F> tempZY = hr_add(hc_im(Z), DeltaSubNY);
                    const hreal<F> normSquared = hr_reduced(hr_add(hr_mul(tempZX, tempZX), hr_mul(tempZY, tempZY)));
                    DeltaNormSquared = hr_reduced(hr_add(hr_mul(DeltaSubNX, DeltaSubNX), hr_mul(DeltaSubNY, DeltaSubNY)));
                    if (hr_cmp_pos(normSquared, TwoFiftySix) > 0) {
                        force_step = true;
                        continue;
                    }
                    if (hr_cmp_pos(normSquared, DeltaNormSquared) < 0 || RefIteration >= count - 1) {
                        DeltaSubNX = tempZX;
                        DeltaSubNY = tempZY;
                        DeltaNormSquared = normSquared;
                        RefIteration = 0;
                    }
                    continue;
                }
                act_literal_step = true; // falls through to the literal step at the bottom of the loop
            }
            if (kBla && !(kRefill && std::is_same<F, float>::value)) {
                const typename FsDev<F>::BLA *b;
                FS_PH(ph_n_outer++);
                for (;;) {
                    FS_PH(ph_t = __builtin_readcyclecounter());
                    uint32_t l;
                    hreal<F> Ax, Ay, Bx, By;
                    hcplx<F> Znat = hc_zero<F>(); // (kNat) the orbit entry the jump arrives at, from the record
                    if constexpr (kNat) {
                        const long long zkey = (long long)(((unsigned long long)(unsigned)DeltaNormSquared.e << 32) |
                                                           (unsigned)__float_as_int(DeltaNormSquared.m));
                        const uint32_t pos = bla_lookup_native(A.nlad, A.nkmax, s_off, A.lm2, RefIteration, zkey, nat_key20);
                        FS_PH(ph_lookup += __builtin_readcyclecounter() - ph_t; ph_n_lookup++);
                        if (pos == 0xFFFFFFFFu)
                            break;
                        const FsBlaRec *nb = A.nrec + pos;
                        const float4 tail = *reinterpret_cast<const float4 *>(&nb->Zre); // {Z.re, Z.im, Z.exp, l}
                        l = (uint32_t)__float_as_int(tail.w);
                        if (RefIteration + l >= count)
                            break;
                        if (iter + l >= n_iterations)
                            break;
                        const float4 mant = *reinterpret_cast<const float4 *>(&nb->Axm);
                        const int4 exps = *reinterpret_cast<const int4 *>(&nb->Axe);
                        Ax = hreal<F>{mant.x, exps.x}, Ay = hreal<F>{mant.y, exps.y};
                        Bx = hreal<F>{mant.z, exps.z}, By = hreal<F>{mant.w, exps.w};
                        Znat = hcplx<F>{tail.x, tail.y, __float_as_int(tail.z)};
                        b = nullptr;
                    } else {
                        b = bla_lookup<F>(s_levels, A.lm2, RefIteration, DeltaNormSquared);
                        FS_PH(ph_lookup += __builtin_readcyclecounter() - ph_t; ph_n_lookup++);
                        if (b == nullptr)
                            break;
                        l = (uint32_t)b->l;
                        if (RefIteration + l >= count)
                            break;
                        if (iter + l >= n_iterations)
                            break;
                        Ax = ldr(b->Ax), Ay = ldr(b->Ay), Bx = ldr(b->Bx), By = ldr(b->By);
                    }
                    FS_PH(ph_t = __builtin_readcyclecounter(); ph_n_jump++;
                          ph_lanes_jump += (uint64_t)__popcll(__builtin_amdgcn_ballot_w64(true)));
                    iter += l;
                    if (kStats) {
                        c_la++;
                        // histogram probe (tools): jumps of >= 1024 / >= 256 orbit steps
                        if (l >= 1024u)
                            atomicAdd((unsigned long long *)&A.stats[6], 1ull);
                        if (l >= 256u)
                            atomicAdd((unsigned long long *)&A.stats[7], 1ull);
                    }
                    const hcplx<F> Z = kNat ? Znat : zref_at(zr, RefIteration + l);
                    bool applied = false;
                    if constexpr (std::is_same<F, float>::value) {
                        // ---- BLA::getValue + the two norms in ONE straight-line evaluation for every exponent alignment
                        // (see the tuned single step below for why this is the same arithmetic): the four products of each
                        // part are summed, in the reference's order, under the maximum of their four exponents; z = Z + dz
                        // under the maximum of the three exponents involved.  A lane whose sums leave [2^-60, 2^60] or
                        // hit an exact zero on the way takes the literal code below (per lane: a jump is per lane anyway).
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        // exact 2^n for -150 < n <= 0, else 0.  The reference ignores an operand from a gap of 120 on
                        // (kExpDiffIgnored); between 120 and 149 this factor is still a tiny power of two instead of 0 --
                        // the same thing once every sum has passed the window test below: an operand that far under
                        // the sum's leading term is absorbed by the float addition either way (one v_ldexp_f32 instead of
                        // shift-add, compare and select; there are twelve of these per jump)
                        auto p2 = [](int n) { return __builtin_amdgcn_ldexpf(1.0f, n); };
                        const f2 D = {DeltaSubNX.m, DeltaSubNY.m}, D0 = {DeltaSub0X.m, DeltaSub0Y.m};
                        // nx = ((Ax DX - Ay DY) + Bx D0X) - By D0Y;  ny = ((Ax DY + Ay DX) + Bx D0Y) + By D0X
                        const f2 pA = (f2){Ax.m, Ax.m} * D;         // (Ax DX, Ax DY)   exps Ax.e + (DX.e, DY.e)
                        const f2 pB = (f2){Ay.m, Ay.m} * D.yx;      // (Ay DY, Ay DX)   exps Ay.e + (DY.e, DX.e)
                        const f2 pC = (f2){Bx.m, Bx.m} * D0;        // (Bx D0X, Bx D0Y) exps Bx.e + (D0X.e, D0Y.e)
                        const f2 pD = (f2){By.m, By.m} * D0.yx;     // (By D0Y, By D0X) exps By.e + (D0Y.e, D0X.e)
                        const int eAx = Ax.e + DeltaSubNX.e, eAy = Ax.e + DeltaSubNY.e;
                        const int eBx = Ay.e + DeltaSubNY.e, eBy = Ay.e + DeltaSubNX.e;
                        const int eCx = Bx.e + DeltaSub0X.e, eCy = Bx.e + DeltaSub0Y.e;
                        const int eDx = By.e + DeltaSub0Y.e, eDy = By.e + DeltaSub0X.e;
                        const int Ex = imax(imax(eAx, eBx), imax(eCx, eDx)), Ey = imax(imax(eAy, eBy), imax(eCy, eDy));
                        const f2 tA = pA * (f2){p2(eAx - Ex), p2(eAy - Ey)};
                        const f2 tB = pB * (f2){p2(eBx - Ex), p2(eBy - Ey)};
                        const f2 tC = pC * (f2){p2(eCx - Ex), p2(eCy - Ey)};
                        const f2 tD = pD * (f2){p2(eDx - Ex), p2(eDy - Ey)};
                        f2 s1, s3;
                        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(s1) : "v"(tA), "v"(tB));
                        const f2 s2 = s1 + tC;
                        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(s3) : "v"(s2), "v"(tD));
                        // (window 2^+-30 on every sum and 2^30 on the record's mantissas: an operand the reference would
                        // ignore -- 120 binades under the leading term -- is then at least 2^30 below half an ulp of any sum
                        // it could be added to, so the power-of-two factor above may stay non-zero there)
                        const float zmn = fminf(fminf(fabsf(s1.x), fabsf(s1.y)), fminf(fabsf(s2.x), fabsf(s2.y)));
                        const int emin = imin(imin(imin(DeltaSubNX.e, DeltaSubNY.e), imin(DeltaSub0X.e, DeltaSub0Y.e)),
                                              imin(imin(Ax.e, Ay.e), imin(imin(Bx.e, By.e), Z.e)));
                        const float amx = fmaxf(fmaxf(fabsf(Ax.m), fabsf(Ay.m)), fmaxf(fabsf(Bx.m), fabsf(By.m)));
                        const float dmx = fmaxf(fabsf(s3.x), fabsf(s3.y)), dmn = fminf(fabsf(s3.x), fabsf(s3.y));
                        const bool sums_ok = dmn >= 0x1p-30f && dmx <= 0x1p30f && zmn > 0.0f && emin > -(1 << 26) && amx <= 0x1p30f;
                        // (round 4) QUIET jump, the jump's form of the quiet step: both parts of the new dz at least four binades
                        // below the orbit value the jump arrives at, and that value below 4 -- then |z| is within [0.646, 1.354] |Z|:
                        // neither the escape nor the rebase test can fire, and z and the norms are not formed.  (|dz|^2 is formed
                        // only when the index the jump lands on is one a table entry can apply at.)
                        const int tex = Ex + (int)__builtin_amdgcn_ubfe(__float_as_int(s3.x), 23, 8) - 127;
                        const int tey = Ey + (int)__builtin_amdgcn_ubfe(__float_as_int(s3.y), 23, 8) - 127;
                        const bool quiet_j = sums_ok && imax(tex, tey) <= Z.e - 4 && Z.e <= 1 && Z.e >= -40 &&
                                             RefIteration + l + 1u < count;
                        if (quiet_j) {
                            applied = true;
                            if (kStats)
                                c_q_jump++;
                            RefIteration += l;
                            DeltaSubNX = hreal<F>{s3.x, Ex};
                            DeltaSubNY = hreal<F>{s3.y, Ey};
                            if ((RefIteration & 3u) == 1u) {
                                const f2 SQ = s3 * s3;
                                const int dd = (Ex - Ey) << 1;
                                const bool sxbig = dd >= 0;
                                const float md = p2(sxbig ? -dd : dd);
                                const float dnm = SQ.x * (sxbig ? 1.0f : md) + SQ.y * (sxbig ? md : 1.0f);
                                const int dne = (sxbig ? Ex : Ey) << 1;
                                const int db = __float_as_int(dnm);
                                DeltaNormSquared = hreal<F>{__int_as_float((db & 0x007FFFFF) | 0x3F800000),
                                                            dne + (int)__builtin_amdgcn_ubfe(db, 23, 8) - 127};
                            }
                        } else {
                        // z = Z + dz under ez; the norms
                        const int ez = imax(imax(Z.e, Ex), Ey);
                        const float zsZ = p2(Z.e - ez);
                        const f2 Zt = (f2){Z.re, Z.im} * (f2){zsZ, zsZ} + s3 * (f2){p2(Ex - ez), p2(Ey - ez)};
                        const f2 ZZ = Zt * Zt;
                        const float nm = ZZ.x + ZZ.y; // exponent 2 ez
                        const f2 SQ = s3 * s3;
                        const int dd = (Ex - Ey) << 1;
                        const bool sxbig = dd >= 0;
                        const float md = p2(sxbig ? -dd : dd);
                        const float dnm = SQ.x * (sxbig ? 1.0f : md) + SQ.y * (sxbig ? md : 1.0f);
                        const int dne = (sxbig ? Ex : Ey) << 1;
                        const float smx = fmaxf(dmx, fmaxf(fabsf(Zt.x), fabsf(Zt.y)));
                        const float smn = fminf(dmn, fminf(fabsf(Zt.x), fabsf(Zt.y)));
                        if (smn >= 0x1p-30f && smx <= 0x1p30f && zmn > 0.0f && emin > -(1 << 26) && amx <= 0x1p30f) {
                            applied = true;
                            if (kStats)
                                c_z_jump++;
                            RefIteration += l;
                            DeltaSubNX = hreal<F>{s3.x, Ex};
                            DeltaSubNY = hreal<F>{s3.y, Ey};
                            const int db = __float_as_int(dnm);
                            DeltaNormSquared = hreal<F>{__int_as_float((db & 0x007FFFFF) | 0x3F800000),
                                                        dne + (int)__builtin_amdgcn_ubfe(db, 23, 8) - 127};
                            if (__builtin_amdgcn_ldexpf(nm, imax((ez << 1) - 8, -400)) > 1.0f)
                                break;
                            if (nm < __builtin_amdgcn_ldexpf(dnm, imax(dne - (ez << 1), -400)) || RefIteration >= count - 1) {
                                const int ex = imax(Z.e, Ex), ey = imax(Z.e, Ey);
                                DeltaSubNX = hreal<F>{__builtin_amdgcn_ldexpf(Zt.x, ez - ex), ex};
                                DeltaSubNY = hreal<F>{__builtin_amdgcn_ldexpf(Zt.y, ez - ey), ey};
                                const int nb = __float_as_int(nm);
                                DeltaNormSquared = hreal<F>{__int_as_float((nb & 0x007FFFFF) | 0x3F800000),
                                                            (ez << 1) + (int)__builtin_amdgcn_ubfe(nb, 23, 8) - 127};
                                RefIteration = 0;
                            }
                        }
                        }
                    }
                    FS_PH(ph_jump += __builtin_readcyclecounter() - ph_t);
                    if (applied)
                        continue;
                    FS_PH(ph_t = __builtin_readcyclecounter(); ph_n_literal++);
                    {
                        const hreal<F> nx = hr_sub(
                            hr_add(hr_sub(hr_mul(Ax, DeltaSubNX), hr_mul(Ay, DeltaSubNY)), hr_mul(Bx, DeltaSub0X)),
                            hr_mul(By, DeltaSub0Y));
                        const hreal<F> ny = hr_add(
                            hr_add(hr_add(hr_mul(Ax, DeltaSubNY), hr_mul(Ay, DeltaSubNX)), hr_mul(Bx, DeltaSub0Y)),
                            hr_mul(By, DeltaSub0X));
                        DeltaSubNX = nx;
                        DeltaSubNY = ny;
                    }
                    RefIteration += l;
                    const hreal<F> tempZX = hr_add(hc_re(Z), DeltaSubNX);
                    const hreal<F> tempZY = hr_add(hc_im(Z), DeltaSubNY);
                    const hreal<F> normSquared = hr_reduced(hr_add(hr_mul(tempZX, tempZX), hr_mul(tempZY, tempZY)));
                    DeltaNormSquared = hr_reduced(hr_add(hr_mul(DeltaSubNX, DeltaSubNX), hr_mul(DeltaSubNY, DeltaSubNY)));
                    if (hr_cmp_pos(normSquared, TwoFiftySix) > 0)
                        break;
                    if (hr_cmp_pos(normSquared, DeltaNormSquared) < 0 || RefIteration >= count - 1) {
                        DeltaSubNX = tempZX;
                        DeltaSubNY = tempZY;
                        DeltaNormSquared = normSquared;
                        RefIteration = 0;
                    }
                    FS_PH(ph_literal += __builtin_readcyclecounter() - ph_t);
                }
                if (iter >= n_iterations)
                    break;
            }

            // ---- perturbation-only mode, float: runs of "quiet" steps (same idea as the tuned LAv2 loop, in the scalar
            // HDRFloat arithmetic of Fractal.cpp:2342-2361).  A step is quiet when both parts of the new dz are at least
            // 2^4 below the next orbit value and that value is below 8: then neither the escape test nor the rebase test
            // can fire (|Z'| in [0.5, 1.42) 2^Zne, |dz| < 2.83 * 2^(Zne-4): |z| > 1.8 |dz| and |z|^2 < 41), z and the two
            // norms are not needed, and the step reduces to the dz update under the alignment cases listed at the tuned
            // single step below -- evaluated with the same IEEE operations in the same order.  Everything is wave-voted;
            // a lane that fails a condition sends the wave to the single step, which decides exactly.
            //   orbit companion zq[i] = {re, im, s = ~exp + 116 | poison};  aX = OXe + sC = n4 + 116 (n4 = OXe - exp(2Z));
            //   E' = max(OXe, OYe) - sC = E - 116;  ncB = (dce - 5) - E' = nc + 111.
            //   The "gap >= 120: smaller operand ignored" rule of the reference's add is a clamped exponent field that is 0
            //   exactly at the cut-off; the 2^-7 this costs is pre-paid by carrying dz's mantissas times 128 (O128), and it
            //   is consumed once on the way to N: T = Z + O128 * 2^(n-7), P = O128 * T = 128 * (O * T), N = P * 2^(e-7).
            //   valid: aX', aY' <= 111 (new dz 2^4 below Z'), ncB <= 111 (N bigger than dc), both parts of Q normal
            //   non-zero, no exact zero in T or N (the literal adds reset the exponent there).
            // ---- first chance: runs of *scaled* quiet steps (see k_lav2_hdr32_fast: HDRFloat operations are the correctly
            // rounded binary32 operations on the represented values, so while nothing leaves binary32's normal range the
            // step can run on plain floats under one power-of-two scale per lane).  The scalar-HDRFloat step of
            // Fractal.cpp:2342-2361 -- X' = X (2Zx + X) - Y (2Zy + Y) + cX,  Y' = X (2Zy + Y) + Y (2Zx + X) + cY, each part
            // with its own exponent -- is the same sequence of roundings as the complex one: s = fma(w, 2^E, 2Z),
            // q = (w.x s.x - w.y s.y, w.x s.y + w.y s.x) + c 2^-E.  Same acceptance tests, same companion array.
            bool sc_stopped = false;
            FS_PO(po_t1 = __builtin_readcyclecounter(); po_n_run++);
            if constexpr (kRuns && !kBla && std::is_same<F, float>::value) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                typedef float f3 __attribute__((ext_vector_type(3)));
                typedef float f4 __attribute__((ext_vector_type(4)));
                const float4 *__restrict__ zs = A.zs;
                const uint32_t MaxRefS = count - 1;
                bool fl_per_trip = false; // (wave-uniform) the next run attempt uses the per-trip floor verdicts
                for (;;) {
                    const float4 e0 = zs[RefIteration];
                    // Floor form of the acceptance tests (round 4; derivation above FS_FL_EVERY).  It is the simpler case here:
                    // the reference's arithmetic is scalar HDRFloat -- every operand reduced, a mantissa product in [1, 4),
                    // an aligned sum (sums have no underflow error) -- so the reference itself never loses bits to
                    // underflow, and its "gap >= 120: addend ignored" rule acts per part, where an addend 2^120 below the
                    // other is absorbed by the IEEE sum as well.  What is left is (u) on the scaled side: a product below
                    // 2^-126 in the run's units.  Such a product is either absorbed by the term it is added to (>= 2^-100: the
                    // same sum in both arithmetics) or leaves a part of the new state below 2^-72 -- under the floor.
                    // Scale: E = larger exponent + 24 (max|w| starts at 2^-24); dzs = the state with max part in [1, 2).
                    const int E0 = imax(DeltaSubNX.e, DeltaSubNY.e);
                    const int E = E0 + kScaleShift;
                    const f2 dzs = {__builtin_amdgcn_ldexpf(DeltaSubNX.m, imax(DeltaSubNX.e - E0, -200)),
                                    __builtin_amdgcn_ldexpf(DeltaSubNY.m, imax(DeltaSubNY.e - E0, -200))};
                    const float sE = __builtin_amdgcn_ldexpf(1.0f, E);
                    const int dshx = DeltaSub0X.e - E, dshy = DeltaSub0Y.e - E;
                    const f2 dcs = {__builtin_amdgcn_ldexpf(DeltaSub0X.m, imax(imin(dshx, 100), -200)),
                                    __builtin_amdgcn_ldexpf(DeltaSub0Y.m, imax(imin(dshy, 100), -200))};
                    const float mx0 = fs_max_abs(dzs.x, dzs.y);
                    const float mn0 = fs_min_abs(dzs.x, dzs.y);
                    const uint32_t left_ref = RefIteration + 1 < MaxRefS ? MaxRefS - 1 - RefIteration : 0u;
                    const uint32_t left_it = n_iterations - 1 - iter; // iter < n_iterations here
                    const uint32_t left = left_ref < left_it ? left_ref : left_it;
                    const int Esh = (E < -254 ? -254 : (E > 127 ? 127 : E)) * (1 << 23);
                    // dz 2^-E exact and above the floor; |dc| 2^-E < 2^7
                    const bool start_ok = scaled_startable(e0) && mn0 >= FS_FL_FLOOR * __builtin_amdgcn_ldexpf(1.0f, kScaleShift) &&
                                          mx0 >= 1.0f && mx0 < 2.0f && imax(dshx, dshy) <= 30 - kScaleShift;
                    const uint32_t run_len = scaled_run_length_po(left);
                    if (__builtin_amdgcn_ballot_w64(!start_ok) != 0ull || run_len == 0u)
                        break;
                    const f2 sE2 = {sE, sE};
                    // (keeping the first step of a failed trip, as k_lav2_hdr32_fast does, loses here: the exit conversion drops
                    // the cached orbit value the careful step would reuse; measured 437 -> 453 ms on C2)
                    f2 w0 = dzs * __builtin_amdgcn_ldexpf(1.0f, -kScaleShift), z0 = {e0.x, e0.y}, w2, z2, wO;
                    uint32_t c = 0;
                    bool failed;
                    bool fl_redo = false, fl_next = false;
                    const uint32_t ref_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)RefIteration);
                    if (__builtin_amdgcn_ballot_w64(RefIteration != ref_u) == 0ull) {
                        // Entries through the scalar cache (all lanes read the same ones): the hand-scheduled untested loop
                        // of k_lav2_hdr32_fast (FS_FAST_LOOP_FD / _FL; here with the next body's cache lines requested a body ahead:
                        // a wave that is alone on its SIMD -- the interior pixels' 4.7 M-step chains that decide C2's frame
                        // time -- pays per instruction issued and for every L2 round trip it waits out), and four-step
                        // blocks with their bound tests where the block test fails.  A failed trip ends the run at its start
                        // state (keeping its first step loses here, see above).
                        const float4 *zpu = zs + ref_u + 1;
                        const int imdc = __float_as_int(fs_max_abs(dcs.x, dcs.y));
                        float mxS = mx0 * __builtin_amdgcn_ldexpf(1.0f, -kScaleShift);
                        int pwi = __builtin_amdgcn_readfirstlane(__float_as_int(e0.w));
                        f2 zS = {__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(e0.x))),
                                 __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(e0.y)))};
                        f2 wv = w0;
                        uint32_t cs = 0;
                        const uint32_t rl = (uint32_t)__builtin_amdgcn_readfirstlane((int)run_len);
                        const uint32_t lim8 = (rl << 4) - 0x80u; // run lengths are 16 / 64 / 256 steps
                        const uint32_t lim16 = (rl << 4) - 0x100u;
                        const float4 *const zpb = zpu;
                        const float2 *const zpb2 = A.zs2 + ref_u + 1; // the same entries in the 16-step body's compact form
                        const float4 *const zqbp = A.zqb + ref_u + 1;
                        for (;;) {
#ifdef FS_VERIFY_BLOCK_BOUND
                            // VERIFICATION BUILD (tools/block_bound_check.py), as in k_lav2_hdr32_fast
                            const int vg_ = __float_as_int(mxS) > imdc ? __float_as_int(mxS) : imdc;
                            const bool bt_pass = __builtin_amdgcn_ballot_w64(vg_ + Esh > pwi) == 0ull;
                            if (kStats && bt_pass)
                                c_free_steps += 4;
#else
                            {
                                f2 r1, r2, r3 = wv, ts_, ta_; // (r3 = wv: the pending pair on entry is the state itself)
                                uint64_t xacc_ = 0;           // (verification build only)
                                float tn_, tl_;
                                uint64_t msk_;
                                int st, ebo, pf_, pg_, ph_, pi_, pj_;
                                uint32_t oc_, cko_;
                                f2 ck_;
                                float th_;
                                int va_;
                                uint32_t off = cs << 4;
                                const uint32_t c_in = cs;
#if FS_FL_EVERY && !defined(FS_VERIFY_FLOOR)
                                if (!fl_per_trip) {
                                    {
                                        FS_PO(po_t2 = __builtin_readcyclecounter(); po_n_asm++);
#ifdef FS_FD16_SERIAL /* A/B: round 4's body -- one wait right behind the request, the next body's lines warmed */
                                        FS_FAST_LOOP_FD16(FS_PF16_NEXT_BODY);
#else
                                        FS_FAST_LOOP_FD16P;
#endif
                                        FS_PO(po_asm += __builtin_readcyclecounter() - po_t2);
                                    }
                                    ebo = 0;
                                } else
#endif
                                {
                                    FS_FAST_LOOP_FL(FS_PF_NEXT_BODY);
                                }
#ifdef FS_VERIFY_FLOOR
                                if (kStats && xacc_ != 0ull)
                                    c_blk_violation++;
#endif
                                st = __builtin_amdgcn_readfirstlane(st); // (asm results count as divergent)
                                zS = (f2){__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(zS.x))),
                                          __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(zS.y)))};
                                if (st == 3) {
#ifdef FS_FD16_SERIAL
                                    // (deferred verdict) a state of this invocation fell below the floor: nothing of the run
                                    // has been committed -- the same run again with the per-trip verdicts
                                    fl_redo = true;
                                    break;
#else
                                    // (deferred verdict, per body) a state of the last body fell below the floor: the statement
                                    // is back at its checkpoint -- the steps up to there are certified and committed, the next
                                    // run starts there with the per-trip verdicts
                                    if (fl_per_trip) { // (the per-trip loop has no status 3)
                                        fl_redo = true;
                                        break;
                                    }
                                    cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)off) >> 4;
                                    if (kStats)
                                        c_free_steps += cs - c_in;
                                    c = cs, wO = wv, failed = false, fl_next = true;
                                    if (kStats)
                                        c_end[3]++;
                                    break;
#endif
                                }
#ifndef FS_FD16_SERIAL
                                if (st == 4) {
                                    // a block test inside the last body failed: the statement is back at the body's checkpoint.  Its
                                    // first block passed its test, so it runs once more -- through the tested form below, which is
                                    // what every block in front of a failed test gets; 2Z and the block bound of the entry the state
                                    // is at come from the companion array (the statement's copies are those of a later entry)
                                    cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)off) >> 4;
                                    if (kStats)
                                        c_free_steps += cs - c_in;
                                    const float4 ez = zs[ref_u + cs];
                                    zS = (f2){__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ez.x))),
                                              __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ez.y)))};
                                    pwi = __builtin_amdgcn_readfirstlane(__float_as_int(ez.w));
                                    st = 0;
                                    if (kStats)
                                        c_end[4]++;
                                } else
#endif
                                {
                                    cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)off) >> 4;
                                    if (kStats)
                                        c_free_steps += cs - c_in;
                                    pwi = __builtin_amdgcn_readfirstlane(pwi);
                                }
                                if (st != 0) {
                                    c = cs, wO = st == 1 ? wv : r2, failed = true;
                                    break;
                                }
                            }
#endif
                            if (cs + 4u > rl) {
                                c = cs, wO = wv, failed = false;
                                if (kStats)
                                    c_end[0]++;
                                break;
                            }
                            // H where a block starts: the run ends and the next one re-centres the scale
                            if (__builtin_amdgcn_ballot_w64(!(mxS < FS_FL_HIGH)) != 0ull) {
                                c = cs, wO = wv, failed = false;
                                if (kStats)
                                    c_end[1]++;
                                break;
                            }
                            if (kStats)
                                c_tested_blocks++;
                            typedef float f16 __attribute__((ext_vector_type(16)));
                            f16 U;
                            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(U) : "s"(zpb + cs));
                            f2 tp_, tq_, w4;
                            FS_STEP_ARITH(wv, zS, tp_, a)
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(U), "+v"(tp_));
                            const f4 ua = U.s0123, ub = U.s4567, uc = U.s89ab, ud = U.scdef;
                            const f2 up_ = {ua.x, ua.y}, uq_ = {uc.x, uc.y};
                            uint64_t vp_ = 0, vq_ = 0;
                            FS_STEP_BOUND(tp_, a, vp_, ua.z)
                            FS_STEP_FLOOR_FIRST(tp_, vp_)
                            FS_STEP_ARITH(tp_, up_, w2, b)
                            FS_STEP_BOUND(w2, b, vp_, ub.z)
                            FS_STEP_FLOOR(w2, vp_)
#ifdef FS_VERIFY_BLOCK_BOUND
                            if (kStats && bt_pass &&
                                (__builtin_amdgcn_ballot_w64(__float_as_int(mx_a) + Esh > __float_as_int(ua.z)) |
                                 __builtin_amdgcn_ballot_w64(__float_as_int(mx_b) + Esh > __float_as_int(ub.z))) != 0ull)
                                c_blk_violation++;
#endif
                            if (vp_ != 0ull) {
                                c = cs, wO = wv, failed = true;
                                if (kStats)
                                    c_end[2]++;
                                break;
                            }
                            z2 = (f2){ub.x, ub.y};
                            FS_STEP_ARITH(w2, z2, tq_, c_)
                            FS_STEP_BOUND(tq_, c_, vq_, uc.z)
                            FS_STEP_FLOOR_FIRST(tq_, vq_)
                            FS_STEP_ARITH(tq_, uq_, w4, d)
                            FS_STEP_BOUND(w4, d, vq_, ud.z)
                            FS_STEP_FLOOR(w4, vq_)
#ifdef FS_VERIFY_BLOCK_BOUND
                            if (kStats && bt_pass &&
                                (__builtin_amdgcn_ballot_w64(__float_as_int(mx_c_) + Esh > __float_as_int(uc.z)) |
                                 __builtin_amdgcn_ballot_w64(__float_as_int(mx_d) + Esh > __float_as_int(ud.z))) != 0ull)
                                c_blk_violation++;
#endif
                            if (vq_ != 0ull) {
                                c = cs + 2, wO = w2, failed = true;
                                if (kStats)
                                    c_end[2]++;
                                break;
                            }
                            cs += 4;
                            wv = w4, mxS = mx_d, zS = (f2){ud.x, ud.y}, pwi = __float_as_int(ud.w);
                            if (cs >= rl) {
                                c = cs, wO = wv, failed = false;
                                if (kStats)
                                    c_end[0]++;
                                break;
                            }
                        }
                    } else {
                        const uint32_t lane_off = (RefIteration + 1) * 16u;
                        const float4 *zp = zs;
                        f3 ent_a, ent_b, ent_c_, ent_d;
                        for (;;) {
                            FS_SCALED_LOAD("0", a, w0)
                            FS_SCALED_LOAD("16", b, w0)
                            FS_SCALED_LOAD("32", c_, w0)
                            FS_SCALED_LOAD("48", d, w0)
                            f2 t1, u1;
                            uint64_t v1 = 0;
                            FS_SCALED_STEP(w0, z0, t1, u1, a, v1, false,
                                           asm volatile("s_waitcnt vmcnt(3)" : "+v"(ent_a), "+v"(mx_a)), ent_a.x, ent_a.y,
                                           ent_a.z);
                            FS_SCALED_STEP(t1, u1, w2, z2, b, v1, true,
                                           asm volatile("s_waitcnt vmcnt(2)" : "+v"(ent_b), "+v"(mx_b)), ent_b.x, ent_b.y,
                                           ent_b.z);
                            if (v1 != 0ull) {
                                asm volatile("s_waitcnt vmcnt(0)" ::"v"(ent_c_), "v"(ent_d)); // nothing stays in flight
                                wO = w0, failed = true;
                                break;
                            }
                            c += 2;
                            f2 t3, u3;
                            uint64_t v2 = 0;
                            FS_SCALED_STEP(w2, z2, t3, u3, c_, v2, false,
                                           asm volatile("s_waitcnt vmcnt(1)" : "+v"(ent_c_), "+v"(mx_c_)), ent_c_.x,
                                           ent_c_.y, ent_c_.z);
                            FS_SCALED_STEP(t3, u3, w0, z0, d, v2, true,
                                           asm volatile("s_waitcnt vmcnt(0)" : "+v"(ent_d), "+v"(mx_d)), ent_d.x, ent_d.y,
                                           ent_d.z);
                            if (v2 != 0ull) {
                                wO = w2, failed = true;
                                break;
                            }
                            c += 2;
                            zp += 4;
                            if (c >= run_len) {
                                wO = w0, failed = false;
                                break;
                            }
                        }
                        // a run that ends in its first trip leaves the loads of the second in flight: they land before anything else happens
                        asm volatile("s_waitcnt vmcnt(0) ; scaled run, loop exit" ::"v"(ent_a), "v"(ent_b), "v"(ent_c_), "v"(ent_d));
                    }
                    if (fl_redo) {
                        fl_per_trip = true;
                        continue;
                    }
                    fl_per_trip = fl_next;
                    if (c != 0u) {
                        // back to two reduced HDRFloats: each part's own exponent moves out of the float (exact; an accepted
                        // state has no zero part)
                        const int kx = (int)(((uint32_t)__float_as_int(wO.x) >> 23) & 0xFFu) - 127;
                        const int ky = (int)(((uint32_t)__float_as_int(wO.y) >> 23) & 0xFFu) - 127;
                        DeltaSubNX = hreal<F>{__builtin_amdgcn_ldexpf(wO.x, -kx), E + kx};
                        DeltaSubNY = hreal<F>{__builtin_amdgcn_ldexpf(wO.y, -ky), E + ky};
                        RefIteration += c;
                        iter += c;
                        if (kStats) {
                            c_pt += c;
                            c_la += c; // (no BLA on this path: the slot carries the scaled steps)
                            c_runs++;
                        }
                        Zcached_at = 0xFFFFFFFFu;
                    }
                    if (failed) {
                        sc_stopped = true;
                        break;
                    }
                }
            }
            FS_PO(po_t2 = __builtin_readcyclecounter(); po_run += po_t2 - po_t1);
            if constexpr (kRuns && !kBla && std::is_same<F, float>::value) {
              if (!sc_stopped) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                typedef float f3 __attribute__((ext_vector_type(3)));
                const float4 *__restrict__ zq = A.zq;
                f2 O128 = (f2){DeltaSubNX.m, DeltaSubNY.m} * 128.0f;
                int OXe = DeltaSubNX.e, OYe = DeltaSubNY.e;
                const float4 zc0 = zq[RefIteration];
                f2 Zc = {zc0.x, zc0.y};
                int sC = __float_as_int(zc0.z);
                int aX = OXe + sC, aY = OYe + sC;
                // entry: both parts reduced (mantissa in [1,2)) and at least 2^4 below the orbit value
                const bool entry_ok = (__float_as_int(DeltaSubNX.m) & 0x7F800000) == 0x3F800000 &&
                                      (__float_as_int(DeltaSubNY.m) & 0x7F800000) == 0x3F800000 && aX <= 111 && aY <= 111;
                bool stop = __builtin_amdgcn_ballot_w64(!entry_ok) != 0ull;
                bool retry_scaled = false; // a clean chunk goes back to the scaled runs (see k_lav2_hdr32_fast)
                const uint32_t lane_off = (RefIteration + 1) * 16u;
                const f2 dcm128 = (f2){DeltaSub0X.m, DeltaSub0Y.m} * 128.0f;
                const int dcXB = DeltaSub0X.e - 5, dcYB = DeltaSub0Y.e - 5;
                const uint32_t MaxRef = count - 1;
                uint32_t done = 0;
#define FS_SQ_WAIT_ZERO(E) asm volatile("s_waitcnt vmcnt(0)" : "+v"(E))
#define FS_SQ_WAIT_NONE(E)
#define FS_SQ_LOAD(ENT, K)                                                                                          \
    {                                                                                                               \
        const float4 *zc_ = zq + (K);                                                                               \
        /* "+v"(OXe): nothing is written, it only pins the load ahead of the arithmetic that reads OXe */           \
        asm volatile("global_load_dwordx3 %0, %2, %3" : "=v"(ENT), "+v"(OXe) : "v"(lane_off), "s"(zc_));            \
    }
                // WAIT is ZERO (entry loaded by hand in this step) or NONE (entry came through an ordinary load)
#define FS_SQ_STEP(VIOL, ent_, WAIT)                                                                                \
    const f2 tsc_ = {__int_as_float((imax(imin(aX, 116), -4) << 23) + (4 << 23)),                                   \
                     __int_as_float((imax(imin(aY, 116), -4) << 23) + (4 << 23))};                                  \
    const f2 T_ = Zc + O128 * tsc_;               /* (T4.m, T3.m), exponent of 2Z */                               \
    const int dxy_ = aX - aY, dyx_ = aY - aX;                                                                       \
    const f2 P1_ = O128.xx * T_;                  /* 128 * (B1.m, C1.m), exponent OXe + exp(2Z) */                 \
    const f2 P2_ = O128.yy * T_.yx;               /* 128 * (B2.m, C2.m), exponent OYe + exp(2Z) */                 \
    const f2 P1s_ = P1_ * __int_as_float((imax(imin(dxy_, 0), -kExpDiffIgnored) << 23) + (kExpDiffIgnored << 23));  \
    const f2 P2s_ = P2_ * __int_as_float((imax(imin(dyx_, 0), -kExpDiffIgnored) << 23) + (kExpDiffIgnored << 23));  \
    f2 N_;                                        /* (B1' - B2', C1' + C2'), exponent E */                          \
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(N_) : "v"(P1s_), "v"(P2s_));                     \
    const int Ep_ = imax(OXe, OYe) - sC;          /* E - 116 */                                                    \
    const int ncx_ = dcXB - Ep_, ncy_ = dcYB - Ep_;                                                                 \
    const int cx_ = imax(imin(ncx_, 112), -9), cy_ = imax(imin(ncy_, 112), -9);                                     \
    const f2 dsc_ = {__int_as_float((cx_ << 23) + (9 << 23)), __int_as_float((cy_ << 23) + (9 << 23))};             \
    const f2 Q_ = N_ + dcm128 * dsc_;                                                                               \
    const int qxb_ = __float_as_int(Q_.x), qyb_ = __float_as_int(Q_.y);                                             \
    const int fx_ = (int)__builtin_amdgcn_ubfe(qxb_, 23, 8), fy_ = (int)__builtin_amdgcn_ubfe(qyb_, 23, 8);         \
    const int nxe_ = Ep_ + fx_ - 11, nye_ = Ep_ + fy_ - 11;                                                         \
    FS_SQ_WAIT_##WAIT(ent_);                                                                                        \
    const int sN_ = __float_as_int(ent_.z);                                                                         \
    const int aXn_ = nxe_ + sN_, aYn_ = nye_ + sN_;                                                                 \
    const int hi_ = imax(imax(imax(aXn_, aYn_), cx_), cy_);                                                         \
    const float tiny_ = __builtin_fminf(fs_min_abs(T_.x, T_.y),              \
                                        fs_min_abs(N_.x, N_.y));             \
    const uint64_t VIOL = __builtin_amdgcn_ballot_w64(imin(fx_, fy_) < 1) | __builtin_amdgcn_ballot_w64(hi_ > 111) | \
                          __builtin_amdgcn_ballot_w64(!(tiny_ > 0.0f))
#define FS_SQ_COMMIT(ent_)                                                                                          \
    O128 = (f2){__int_as_float((qxb_ & 0x807FFFFF) | 0x43000000), __int_as_float((qyb_ & 0x807FFFFF) | 0x43000000)}; \
    OXe = nxe_, OYe = nye_, aX = aXn_, aY = aYn_, sC = sN_;                                                         \
    Zc = (f2){ent_.x, ent_.y}
                while (!stop) {
                    const uint32_t r0 = RefIteration + done, i0 = iter + done;
                    const uint32_t left_ref = r0 + 1 < MaxRef ? MaxRef - 1 - r0 : 0u;
                    const uint32_t left_it = n_iterations - 1 - i0; // iter < n_iterations here
                    uint32_t left = left_ref < left_it ? left_ref : left_it;
                    if (__builtin_amdgcn_ballot_w64(left < 66u) == 0ull) {
                        // every running lane has more than 64 quiet-eligible steps ahead: no per-step counter.  The orbit
                        // entry of step k+1 is requested while step k computes (a wave that is alone on its SIMD -- the
                        // long interior chains -- would otherwise sit out the full load latency every step).  These are
                        // ordinary loads: a hand-issued load may not stay in flight across the loop's back edge (the
                        // register allocator is free to copy its destination before the data has arrived).
                        uint32_t c = 0;
                        const float4 *zl = zq + (RefIteration + 1 + done);
                        float4 nxt = zl[0];
                        for (; c < 64u; c++) {
                            const float4 cur = nxt;
                            nxt = zl[c + 1];
                            f3 entC = {cur.x, cur.y, cur.z};
                            FS_SQ_STEP(vA, entC, NONE);
                            if (vA != 0ull) {
                                stop = true;
                                break;
                            }
                            FS_SQ_COMMIT(entC);
                        }
                        done += c;
                        if (!stop) {
                            retry_scaled = true;
                            break;
                        }
                    } else {
                        for (;;) {
                            f3 entT;
                            FS_SQ_LOAD(entT, done);
                            FS_SQ_STEP(vT, entT, ZERO);
                            if ((vT | __builtin_amdgcn_ballot_w64(left == 0u)) != 0ull)
                                break;
                            FS_SQ_COMMIT(entT);
                            left--;
                            done++;
                        }
                        stop = true;
                    }
                }
#undef FS_SQ_LOAD
#undef FS_SQ_WAIT_ZERO
#undef FS_SQ_WAIT_NONE
#undef FS_SQ_STEP
#undef FS_SQ_COMMIT
                if (done != 0) {
                    RefIteration += done;
                    iter += done;
                    if (kStats)
                        c_pt += done;
                    DeltaSubNX = hreal<F>{O128.x * 0.0078125f, OXe};
                    DeltaSubNY = hreal<F>{O128.y * 0.0078125f, OYe};
                    Zcached_at = 0xFFFFFFFFu;
                }
                if (retry_scaled) {
                    FS_PO(po_quiet += __builtin_readcyclecounter() - po_t2);
                    continue;
                }
              }
            }

            FS_PO(po_quiet += __builtin_readcyclecounter() - po_t2);
            FS_PH(ph_t = __builtin_readcyclecounter(); ph_n_step++;
                  ph_lanes_step += (uint64_t)__popcll(__builtin_amdgcn_ballot_w64(true)));
            const hreal<F> OX = DeltaSubNX, OY = DeltaSubNY;
            // The orbit entry read for the escape test of the previous step is the Z of this step unless a rebase or a
            // BLA jump moved RefIteration: one dependent 16-byte load per step instead of two.
            const hcplx<F> Z = (Zcached_at == RefIteration) ? Zcached : zref_at(zr, RefIteration);

            // ---- tuned single step (float only): ONE straight-line evaluation for every exponent alignment.
            // HDRFloat addition and multiplication are the correctly rounded binary32 operations on the represented
            // values (a product is the float product of the mantissas, an aligned sum the float sum after an exact
            // power-of-two scaling; the "gap >= 120: smaller operand ignored" rule only drops what a float sum absorbs
            // anyway), so a sum may be formed under ANY common exponent that keeps both addends inside binary32's
            // normal range -- not only the one the literal code picks by comparing the operands' exponents.  Each of
            // the four sums of a step (2Z + dz, B1 - B2 / C1 + C2, + dc, Z' + dz') is therefore aligned to the MAXIMUM
            // of the exponents involved (v_max3), every operand gets the exact factor 2^(its exponent - that maximum)
            // (0 from a gap of 120 on, like the reference), and no branch asks which operand was the larger.  This
            // covers, with the same instructions, the three cases the first version of this step (orbit value bigger
            // than dz everywhere) had to hand to the literal code: the step from orbit entry 0 (Z = 0 exactly, every
            // rebase lands there), the step that rebases (Z' + dz' cancels) and dc bigger than dz (pixel start).
            //   What is NOT covered -- and is voted out to the literal step below, which decides exactly: a sum that
            //   is exactly zero (the literal add then resets the exponent), and a sum whose float leaves
            //   [2^-60, 2^60] (a component more than 2^60 below its sibling, or cancellation that deep: the products
            //   built from it could leave the normal range).  Every sum is tested, so products of two sums stay
            //   inside 2^+-120.
            // Committed only when every running lane of the wave passed (one ballot): the fall-back is the literal CPU
            // order.  With lanes re-packed from the pixel queue a wave nearly always holds a lane that is rebasing,
            // which is why the step must not care.
            bool done_fast = false;
            (void)act_literal_step;
            if constexpr (std::is_same<F, float>::value && !(kBla && kRefill)) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                const auto zn4 = zr[RefIteration + 1]; // in bounds: the prepared orbit has two spare entries
                const int Ze1 = Z.e + 1;
                const f2 O = {OX.m, OY.m};
                // exact 2^n for -120 < n <= 0, else 0 (n <= 0 by construction: n = exponent - maximum)
                auto p2 = [](int n) { return __builtin_amdgcn_ldexpf(1.0f, n); }; // (see the jump above)
                // T = 2Z + O under eT
                const int eT = imax(imax(Ze1, OX.e), OY.e);
                const float zsT = p2(Ze1 - eT);
                const f2 tsc = {p2(OX.e - eT), p2(OY.e - eT)};
                const f2 T = (f2){Z.re, Z.im} * (f2){zsT, zsT} + O * tsc; // (T4.m, T3.m), exponent eT
                const f2 P1 = O.xx * T;                                  // (B1.m, C1.m) exponent OX.e + eT
                const f2 P2 = O.yy * T.yx;                               // (B2.m, C2.m) exponent OY.e + eT
                const int dxy = OX.e - OY.e;
                const bool xbig = dxy >= 0;
                const int nad = xbig ? -dxy : dxy;
                const float ms = p2(nad);
                const f2 P1s = P1 * (xbig ? 1.0f : ms);
                const f2 P2s = P2 * (xbig ? ms : 1.0f);
                f2 N; // (B1' - B2', C1' + C2'), exponent E
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(N) : "v"(P1s), "v"(P2s));
                const int E = (xbig ? OX.e : OY.e) + eT;
                // Q = N + dc under EQ
                const int EQ = imax(imax(E, DeltaSub0X.e), DeltaSub0Y.e);
                const float nsQ = p2(E - EQ);
                const f2 dsc = {p2(DeltaSub0X.e - EQ), p2(DeltaSub0Y.e - EQ)};
                const f2 Q = N * (f2){nsQ, nsQ} + (f2){DeltaSub0X.m, DeltaSub0Y.m} * dsc;
                // scalar Reduce of each part
                const int qxb = __float_as_int(Q.x), qyb = __float_as_int(Q.y);
                const int fx = (int)__builtin_amdgcn_ubfe(qxb, 23, 8), fy = (int)__builtin_amdgcn_ubfe(qyb, 23, 8);
                const float nxm = __int_as_float((qxb & 0x807FFFFF) | 0x3F800000);
                const float nym = __int_as_float((qyb & 0x807FFFFF) | 0x3F800000);
                const int nxe = EQ + fx - 127, nye = EQ + fy - 127;
                // dn = nx^2 + ny^2 (formed where it is needed: see the quiet step below)
                const int Zne = __float_as_int(zn4.z);
#define FS_STEP_DN()                                                                                                \
    const f2 SQ = (f2){nxm, nym} * (f2){nxm, nym};                                                                  \
    const int dd = (nxe - nye) << 1;                                                                                \
    const bool sxbig = dd >= 0;                                                                                     \
    const int nadd = sxbig ? -dd : dd;                                                                              \
    const float md = p2(nadd);                                                                                      \
    const float dnm = SQ.x * (sxbig ? 1.0f : md) + SQ.y * (sxbig ? md : 1.0f);                                      \
    const int dne = (sxbig ? nxe : nye) << 1;
                // the sums of the dz update inside [2^-30, 2^30] (also excludes zeros, denormals, infinities and NaNs)
                const float dmx = fmaxf(fmaxf(fmaxf(fabsf(T.x), fabsf(T.y)), fmaxf(fabsf(N.x), fabsf(N.y))),
                                        fmaxf(fabsf(Q.x), fabsf(Q.y)));
                const float dmn = fminf(fminf(fminf(fabsf(T.x), fabsf(T.y)), fminf(fabsf(N.x), fabsf(N.y))),
                                        fminf(fabsf(Q.x), fabsf(Q.y)));
                const bool ok_dz = dmn >= 0x1p-30f && dmx <= 0x1p30f && (OX.e < OY.e ? OX.e : OY.e) > -(1 << 26) &&
                                   Zne > -(1 << 26) && RefIteration + 1 < count;
                // Quiet step: both parts of the new dz at least four binades below the orbit value it arrives at (whose
                // larger part is in [0.5, 2) 2^Zne), and that value below 4: |dz'| < 2^(Zne - 2.5) = 0.177 * 2^Zne <=
                // 0.354 |Z'|, so |z| = |Z' + dz'| is in [0.646, 1.354] |Z'| -- |z|^2 >= 3.3 |dz'|^2 (the rebase test cannot
                // fire) and |z|^2 < 59 for Zne <= 1 (nor the escape test).  z and its norm are then not formed at all; when
                // every stepping lane of the wave is in this state that is a quarter of the step's instructions.
                const bool quiet = imax(nxe, nye) <= Zne - 4 && Zne <= 1 && Zne >= -40 && RefIteration + 2 < count;
                if (__builtin_amdgcn_ballot_w64(!(ok_dz && quiet)) == 0ull) {
                    done_fast = true;
                    if (kStats) {
                        c_pt++;
                        c_single++;
                        c_q_step++;
                    }
                    ++RefIteration;
                    Zcached = hcplx<F>{zn4.x, zn4.y, Zne};
                    Zcached_at = RefIteration;
                    DeltaSubNX = hreal<F>{nxm, nxe};
                    DeltaSubNY = hreal<F>{nym, nye};
                    // |dz|^2 is read by the next table lookup only, and a table entry can only apply at orbit indices
                    // m = 1 (mod 4): three quiet steps in four leave it unformed (round 4; without a table nothing reads it)
                    if (kBla && __builtin_amdgcn_ballot_w64((RefIteration & 3u) == 1u) != 0ull) {
                        FS_STEP_DN()
                        const int db = __float_as_int(dnm);
                        DeltaNormSquared = hreal<F>{__int_as_float((db & 0x007FFFFF) | 0x3F800000),
                                                    dne + (int)__builtin_amdgcn_ubfe(db, 23, 8) - 127};
                    }
                    ++iter;
                    FS_PH(ph_step += __builtin_readcyclecounter() - ph_t);
                    continue;
                }
                FS_STEP_DN()
#undef FS_STEP_DN
                // z = Z' + n under ez
                const int ez = imax(imax(Zne, nxe), nye);
                const float zsZ = p2(Zne - ez);
                const f2 zsc = {p2(nxe - ez), p2(nye - ez)};
                const f2 Zt = (f2){zn4.x, zn4.y} * (f2){zsZ, zsZ} + (f2){nxm, nym} * zsc; // (tempZX.m, tempZY.m), exponent ez
                const f2 ZZ = Zt * Zt;
                const float nm = ZZ.x + ZZ.y; // exponent 2*ez
                const float smx = fmaxf(fabsf(Zt.x), fabsf(Zt.y));
                const float smn = fminf(fabsf(Zt.x), fabsf(Zt.y));
                const bool ok = ok_dz && smn >= 0x1p-30f && smx <= 0x1p30f;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
                    done_fast = true;
                    if (kStats) {
                        c_pt++;
                        c_single++;
                        c_z_step++;
                    }
                    ++RefIteration;
                    Zcached = hcplx<F>{zn4.x, zn4.y, Zne};
                    Zcached_at = RefIteration;
                    DeltaSubNX = hreal<F>{nxm, nxe};
                    DeltaSubNY = hreal<F>{nym, nye};
                    // Reduce(dn): dnm is in [1, 8]
                    {
                        const int db = __float_as_int(dnm);
                        DeltaNormSquared = hreal<F>{__int_as_float((db & 0x007FFFFF) | 0x3F800000),
                                                    dne + (int)__builtin_amdgcn_ubfe(db, 23, 8) - 127};
                    }
                    // Reduce(n) > 256 <=> nm * 2^(2 ez) > 2^8 (nm a positive normal float; ldexp saturates both ways)
                    if (__builtin_amdgcn_ldexpf(nm, imax((ez << 1) - 8, -400)) > 1.0f)
                        break;
                    // Reduce(n) < Reduce(dn) <=> nm * 2^(2 ez) < dnm * 2^dne  (dne <= 2 ez)
                    if (nm < __builtin_amdgcn_ldexpf(dnm, imax(dne - (ez << 1), -400)) || RefIteration >= count - 1) {
                        // dz = z in the literal representation: each part carries max(exponent of Z', exponent of its
                        // own dz' part) (exact rescaling of the sum formed under ez)
                        const int ex = imax(Zne, nxe), ey = imax(Zne, nye);
                        DeltaSubNX = hreal<F>{__builtin_amdgcn_ldexpf(Zt.x, ez - ex), ex};
                        DeltaSubNY = hreal<F>{__builtin_amdgcn_ldexpf(Zt.y, ez - ey), ey};
                        const int nb = __float_as_int(nm);
                        DeltaNormSquared = hreal<F>{__int_as_float((nb & 0x007FFFFF) | 0x3F800000),
                                                    (ez << 1) + (int)__builtin_amdgcn_ubfe(nb, 23, 8) - 127};
                        RefIteration = 0;
                        FS_CYCLE_CHECK()
                    }
                    ++iter;
                }
            }
            if (done_fast) {
                FS_PH(ph_step += __builtin_readcyclecounter() - ph_t);
                continue;
            }
            FS_PH(ph_n_literal++);

            // ---- generic single step, literal order of Fractal.cpp:2342-2466
            if (kStats && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) ==
                              (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
                atomicAdd((unsigned long long *)&A.stats[5], 1ull); // wave-trips through the literal step
            // Term4 == the inner sum of TermB1, Term3 == the inner sum of TermB2 (same operands, same order)
            const hreal<F> T4 = hr_add(hr_mul2(hc_re(Z)), OX);
            const hreal<F> T3 = hr_add(hr_mul2(hc_im(Z)), OY);
            const hreal<F> TermB1 = hr_mul(OX, T4);
            const hreal<F> TermB2 = hr_mul(OY, T3);
            DeltaSubNX = hr_sub(TermB1, TermB2);
            DeltaSubNX = hr_add(DeltaSubNX, DeltaSub0X);
            hr_reduce(DeltaSubNX);
            DeltaSubNY = hr_add(hr_mul(OX, T3), hr_mul(OY, T4));
            DeltaSubNY = hr_add(DeltaSubNY, DeltaSub0Y);
            hr_reduce(DeltaSubNY);
            if (kStats) {
                c_pt++;
                c_lit_step++;
            }

            ++RefIteration;
            if (RefIteration >= count)
                break;

            const hcplx<F> Z2 = zref_at(zr, RefIteration);
            Zcached = Z2;
            Zcached_at = RefIteration;
            const hreal<F> tempZX = hr_add(hc_re(Z2), DeltaSubNX);
            const hreal<F> tempZY = hr_add(hc_im(Z2), DeltaSubNY);
            const hreal<F> normSquared = hr_reduced(hr_add(hr_mul(tempZX, tempZX), hr_mul(tempZY, tempZY)));
            DeltaNormSquared = hr_reduced(hr_add(hr_mul(DeltaSubNX, DeltaSubNX), hr_mul(DeltaSubNY, DeltaSubNY)));
            if (hr_cmp_pos(normSquared, TwoFiftySix) > 0)
                break;
            if (hr_cmp_pos(normSquared, DeltaNormSquared) < 0 || RefIteration >= count - 1) {
                DeltaSubNX = tempZX;
                DeltaSubNY = tempZY;
                DeltaNormSquared = normSquared;
                RefIteration = 0;
                FS_CYCLE_CHECK()
            }
            ++iter;
            FS_PH(ph_literal += __builtin_readcyclecounter() - ph_t);
        }
#undef FS_CYCLE_CHECK
        if (finished) {
            if (kStats && !kBla && iter >= n_iterations)
                atomicAdd((unsigned long long *)&A.stats[6], 1ull); // probe: pixels that came back with the cap
            if (A.probe_out)
                A.probe_out[(size_t)L * A.probe_pitch + X] = (uint32_t)iter;
            else
                store_iter(A.out, A.frame, L, X, iter);
            have = false;
        }
    }
        if constexpr (!kRefill)
            break;
    }
    if (kStats) {
        add_stats(A.stats, c_single, c_la, c_pt, c_px);
        if (kBla) {
            const uint64_t v[5] = {c_q_step, c_z_step, c_lit_step, c_q_jump, c_z_jump};
            for (int i = 0; i < 5; i++) {
                uint64_t t = v[i];
                for (int off = 32; off > 0; off >>= 1)
                    t += __shfl_down(t, off);
                if ((threadIdx.x & 63) == 0)
                    atomicAdd((unsigned long long *)&A.stats[8 + i], (unsigned long long)t);
            }
        }
        if (!kBla) {
            atomicAdd((unsigned long long *)&A.stats[7], (unsigned long long)c_runs);
            atomicAdd((unsigned long long *)&A.stats[8], (unsigned long long)c_free_steps);
            atomicAdd((unsigned long long *)&A.stats[9], (unsigned long long)c_tested_blocks);
            atomicAdd((unsigned long long *)&A.stats[10], (unsigned long long)c_blk_violation);
            if ((threadIdx.x & 63) == 0) // (per wave: lane 0 of a tile is there from the first step to the wave's last)
                for (int i = 0; i < 5; i++)
                    atomicAdd((unsigned long long *)&A.stats[11 + i], (unsigned long long)c_end[i]);
        }
    }
#ifdef FS_TRACE_WAVES
    // measurement build (tools/c2_wave_trace.py): when and where every wave of a perturbation-only launch ran
    if (kStats && !kBla && !kRefill && A.stats) {
        uint64_t steps = c_pt;
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t o = __shfl_down(steps, off);
            steps = o > steps ? o : steps;
        }
        if ((threadIdx.x & 63) == 0) {
            uint32_t hw_id, xcc_id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
            const uint64_t wave = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
            uint64_t *t = A.stats + 16 + 4 * wave;
            t[0] = ps_trace_t0;
            t[1] = wall_clock64();
            t[2] = ((uint64_t)xcc_id << 32) | hw_id;
            t[3] = (steps & 0xFFFFFFFFull) | (((__builtin_readcyclecounter() - ps_trace_c0) >> 10) << 32);
        }
    }
#endif
#ifdef FS_PROFILE_CYCLES
    if (kStats && !kBla && !kRefill) {
        // (a lane accumulates while its pixel runs: the wave's figures are those of its longest-running lane)
        po_total = __builtin_readcyclecounter() - po_t0;
        uint64_t v[6] = {po_total, po_run, po_asm, po_quiet, po_n_run, po_n_asm};
        for (int i = 1; i < 6; i++)
            for (int off = 32; off > 0; off >>= 1) {
                const uint64_t o = __shfl_xor(v[i], off);
                v[i] = o > v[i] ? o : v[i];
            }
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < 6; i++)
                atomicAdd((unsigned long long *)&A.stats[16 + i], (unsigned long long)v[i]);
    }
    if (kStats && kBla && (threadIdx.x & 63) == 0) {
        // slots 16.. of the statistics buffer (fs_read_stats_raw; the renderer allocates them in this build)
        const uint64_t v[13] = {ph_lookup, ph_jump, ph_step, ph_literal, ph_n_lookup, ph_n_jump,
                                ph_n_step, ph_n_literal, ph_n_outer, ph_lanes_jump, ph_lanes_step, 1, ph_n_scaled};
        for (int i = 0; i < 13; i++)
            atomicAdd((unsigned long long *)&A.stats[16 + i], (unsigned long long)v[i]);
    }
#endif
#undef FS_PH
}

// ------------------------------------------------------------------------------------------------
// Direct double-precision escape time.  CPU twin: Fractal::CalcCpuHDR<uint32_t,double,double>
// (Fractal.cpp:2148-2183): cy = maxY - dy*(double)(float)y; cx starts at minX and is ACCUMULATED (cx += dx)
// along the row, so a lane at column x replays x additions; z0 = c; bailout sum > 4.
// Replaces mandel_1x_double (FractalSharkGpuLib/LowPrecisionKernels.cuh:79-171).
template <bool kStats, class IterT = uint32_t>
__global__ void __launch_bounds__(256) k_direct_f64(FsDirectArgs64 A)
{
    const uint32_t X = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t L = blockIdx.y * 4u + (threadIdx.x >> 6);
    uint64_t c_pt = 0, c_px = 0;
    const uint32_t Y = global_row(A.frame, L);
    const bool live = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    if (live) {
        c_px = 1;
        // Row prefix of cx: minX + dx + dx + ... (X times), rounded after every addition like the CPU loop.
        // The prefix for the wave's first column comes from a table built once per frame (A.cx_row).
        const double cx = A.cx_row[X];
        const double cy = A.maxY - A.dy * (double)((float)Y);
        double zx = cx, zy = cy;
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        IterT i;
        for (i = 0; i < n_iterations; i++) {
            const double zx2 = zx * zx;
            const double zy2 = zy * zy;
            const double sum = zx2 + zy2;
            if (sum > 4.0)
                break;
            zy = 2.0 * zx * zy;
            zx = zx2 - zy2;
            zx += cx;
            zy += cy;
        }
        if (kStats)
            c_pt = i;
        store_iter(A.out, A.frame, L, X, i);
    }
    if (kStats)
        add_stats(A.stats, 0, 0, c_pt, c_px);
}

// cx_row[x] = minX (+ dx) x times, sequentially rounded: a serial scan, done by one lane once per frame
// (W <= 61440 additions).
__global__ void k_direct_row_prefix_f64(double minX, double dx, uint32_t width, double *cx_row)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double cx = minX;
        for (uint32_t x = 0; x < width; x++) {
            cx_row[x] = cx;
            cx += dx;
        }
    }
}

// Orbit preparation for HDRFloat<double>.
__global__ void k_prepare_orbit_hdr64(const fs_orbit_hdr64 *__restrict__ in, FsZ64 *__restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const fs_orbit_hdr64 e = in[i];
    const hcplx64 c = hc_from_hr(hreal64{e.mx, e.ex}, hreal64{e.my, e.ey});
    FsZ64 z;
    z.re = c.re;
    z.im = c.im;
    z.e = c.e;
    z.pad_ = 0;
    z.w = ldexp(1.0, 8 - 2 * (c.e < -500 ? -500 : c.e));
    out[i] = z;
}

// ------------------------------------------------------------------------------------------------
// Direct escape time in HDRFloat<F>.  CPU twin: Fractal::CalcCpuHDR<uint32_t,HDRFloat<F>,F> (Fractal.cpp:2148-2183;
// CpuHDR32 / CpuHDR64): z0 = c, bailout Reduce(zx^2+zy^2) > 4, zy = (2*zx)*zy, zx = zx2 - zy2, += c, Reduce both.
// cx is the CPU's accumulated `cx += dx` (un-reduced HDR adds) from a 1-lane serial scan, like k_direct_f64.
// Replaces mandel_hdr_float (FractalSharkGpuLib/LowPrecisionKernels.cuh:682-777).
template <class F, bool kStats, class IterT = uint32_t>
__global__ void __launch_bounds__(256) k_direct_hdr(FsDirectHdrArgsT<F> A)
{
    const uint32_t X = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t L = blockIdx.y * 4u + (threadIdx.x >> 6);
    uint64_t c_pt = 0, c_px = 0;
    const uint32_t Y = global_row(A.frame, L);
    const bool live = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const hreal<F> cx = A.cx_row[X];
        // T{static_cast<float>(y)}: non-template HDRFloat(T mant) for float, templated (U = float) ctor for double
        const hreal<F> yh = sizeof(F) == 4 ? hr_from_mant<F>((F)(float)Y) : hr_from_number<F>((F)(float)Y);
        const hreal<F> cy = hr_sub(A.maxY, hr_mul(A.dy, yh));
        const hreal<F> Four{F(1), 2};
        const hreal<F> Two{F(1), 1};
        hreal<F> zx = cx, zy = cy;
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        IterT i;
        for (i = 0; i < n_iterations; i++) {
            const hreal<F> zx2 = hr_mul(zx, zx);
            const hreal<F> zy2 = hr_mul(zy, zy);
            const hreal<F> sum = hr_reduced(hr_add(zx2, zy2));
            if (hr_cmp_pos(sum, Four) > 0)
                break;
            zy = hr_mul(hr_mul(Two, zx), zy);
            zx = hr_sub(zx2, zy2);
            zx = hr_add(zx, cx);
            zy = hr_add(zy, cy);
            hr_reduce(zx);
            hr_reduce(zy);
        }
        if (kStats)
            c_pt = i;
        store_iter(A.out, A.frame, L, X, i);
    }
    if (kStats)
        add_stats(A.stats, 0, 0, c_pt, c_px);
}

template <class F> __global__ void k_direct_row_prefix_hdr(hreal<F> minX, hreal<F> dx, uint32_t width, hreal<F> *cx_row)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hreal<F> cx = minX;
        for (uint32_t x = 0; x < width; x++) {
            cx_row[x] = cx;
            cx = hr_add(cx, dx);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// PerturbExtras::SimpleCompression: expand a compressed reference orbit into the prepared form.
// The reference decompresses on the fly in every thread (GPUPerturbSingleResults::GetCompressedComplex*,
// FractalSharkGpuLib/Perturb.cuh:160-326; CPU twin RuntimeDecompressor, PerturbationResultsHelpers.h:46-161): the
// waypoint at or below the wanted index is advanced with z <- z^2 + c in HDRFloat arithmetic.  The value at an index is
// a pure function of the waypoints, and an MI355X has 288 GB of HBM, so the orbit is expanded ONCE per upload --
// one lane per waypoint segment, segments are independent -- and the iteration kernels stay the uncompressed ones.
__global__ void k_decompress_orbit_hdr32(const fs_orbit_hdr32_rc *__restrict__ wp, uint64_t n_wp, uint64_t n_uncompressed,
                                         fs_real_hdr32 cxLow, fs_real_hdr32 cyLow, float4 *__restrict__ out)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_wp)
        return;
    const uint64_t kIndexMask = 0x7FFFFFFFFFFFFFFFull;
    const uint64_t i0 = wp[k].index_and_rebase & kIndexMask;
    const uint64_t i1 = k + 1 < n_wp ? (wp[k + 1].index_and_rebase & kIndexMask) : n_uncompressed;
    hreal32 zx{wp[k].mx, wp[k].ex}, zy{wp[k].my, wp[k].ey};
    const hreal32 cx = ldr(cxLow), cy = ldr(cyLow);
    const hreal32 Two{1.0f, 1};
    for (uint64_t i = i0; i < i1 && i < n_uncompressed; i++) {
        const hcplx32 c = hc_from_hr(zx, zy);
        out[i] = make_float4(c.re, c.im, __int_as_float(c.e), ldexpf(1.0f, 8 - 2 * (c.e < -1000 ? -1000 : c.e)));
        // runOneIter, PerturbationResultsHelpers.h:51-58
        const hreal32 zx_old = zx;
        zx = hr_add(hr_sub(hr_mul(zx, zx), hr_mul(zy, zy)), cx);
        hr_reduce(zx);
        zy = hr_add(hr_mul(hr_mul(Two, zx_old), zy), cy);
        hr_reduce(zy);
    }
}

// ------------------------------------------------------------------------------------------------
// Plain double perturbation with BLA skipping.  CPU twin: Fractal::CalcCpuPerturbationFractalBLA<uint32_t,double,double>
// (Cpu64PerturbedBLA, Fractal.cpp:2266-2470 with T = double: no reductions, plain comparisons).
// Replaces mandel_1x_double_perturb_bla (FractalSharkGpuLib/BLAKernels.cuh:17-168).
namespace {
__device__ __forceinline__ const fs_bla_f64 *bla_lookup_f64(const FsBlaArgsF64 &A, uint32_t m, double z2)
{
    if (m == 0)
        return nullptr;
    const int32_t k = (int32_t)m - 1;
    if ((k & 1) == 1)
        return nullptr;
    int32_t zeros;
    uint32_t ix;
    if (k == 0) {
        if (z2 >= A.levels[2][0].r2)
            return nullptr;
        zeros = 32;
        ix = 0;
    } else {
        zeros = __ffs(k) - 1;
        ix = (uint32_t)k >> zeros;
    }
    const int32_t startLevel = zeros <= A.lm2 ? zeros : A.lm2;
    for (int32_t level = startLevel; level >= 2; --level) {
        const fs_bla_f64 *t = &A.levels[level][ix];
        if (z2 < t->r2)
            return t;
        ix <<= 1;
    }
    return nullptr;
}
} // namespace

template <bool kBla, bool kStats, class IterT = uint32_t>
__global__ void __launch_bounds__(256) k_perturb_bla_f64(FsBlaArgsF64 A)
{
    const uint32_t X = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t L = blockIdx.y * 4u + (threadIdx.x >> 6);
    uint64_t c_la = 0, c_pt = 0, c_px = 0;
    const uint32_t Y = global_row(A.frame, L);
    const bool live = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        const uint32_t count = A.orbit_count;
        const fs_orbit_f64 *__restrict__ orbit = A.orbit;
        IterT iter = 0;
        uint32_t RefIteration = 0;
        double deltaReal = A.dx * (double)X;
        deltaReal -= A.centerX;
        double deltaImaginary = -A.dy * (double)Y;
        deltaImaginary -= A.centerY;
        const double DeltaSub0X = deltaReal, DeltaSub0Y = deltaImaginary;
        double DeltaSubNX = 0, DeltaSubNY = 0, DeltaNormSquared = 0;
        while (iter < n_iterations) {
            if (kBla) {
                const fs_bla_f64 *b;
                while ((b = bla_lookup_f64(A, RefIteration, DeltaNormSquared)) != nullptr) {
                    const uint32_t l = (uint32_t)b->l;
                    if (RefIteration + l >= count)
                        break;
                    if (iter + l >= n_iterations)
                        break;
                    iter += l;
                    if (kStats)
                        c_la++;
                    {
                        const double Ax = b->Ax, Ay = b->Ay, Bx = b->Bx, By = b->By;
                        const double nx = Ax * DeltaSubNX - Ay * DeltaSubNY + Bx * DeltaSub0X - By * DeltaSub0Y;
                        const double ny = Ax * DeltaSubNY + Ay * DeltaSubNX + Bx * DeltaSub0Y + By * DeltaSub0X;
                        DeltaSubNX = nx;
                        DeltaSubNY = ny;
                    }
                    RefIteration += l;
                    const fs_orbit_f64 Z = orbit[RefIteration];
                    const double tempZX = Z.x + DeltaSubNX;
                    const double tempZY = Z.y + DeltaSubNY;
                    const double normSquared = tempZX * tempZX + tempZY * tempZY;
                    DeltaNormSquared = DeltaSubNX * DeltaSubNX + DeltaSubNY * DeltaSubNY;
                    if (normSquared > 256.0)
                        break;
                    if (normSquared < DeltaNormSquared || RefIteration >= count - 1) {
                        DeltaSubNX = tempZX;
                        DeltaSubNY = tempZY;
                        DeltaNormSquared = normSquared;
                        RefIteration = 0;
                    }
                }
                if (iter >= n_iterations)
                    break;
            }
            const double OX = DeltaSubNX, OY = DeltaSubNY;
            const fs_orbit_f64 Z = orbit[RefIteration];
            const double TermB1 = OX * (Z.x * 2 + OX);
            const double TermB2 = OY * (Z.y * 2 + OY);
            DeltaSubNX = TermB1 - TermB2;
            DeltaSubNX += DeltaSub0X;
            const double Term3 = Z.y * 2 + OY;
            const double Term4 = Z.x * 2 + OX;
            DeltaSubNY = OX * Term3 + OY * Term4;
            DeltaSubNY += DeltaSub0Y;
            if (kStats)
                c_pt++;
            ++RefIteration;
            if (RefIteration >= count)
                break;
            const fs_orbit_f64 Z2 = orbit[RefIteration];
            const double tempZX = Z2.x + DeltaSubNX;
            const double tempZY = Z2.y + DeltaSubNY;
            const double nT1 = tempZX * tempZX;
            const double nT2 = tempZY * tempZY;
            const double normSquared = nT1 + nT2;
            DeltaNormSquared = DeltaSubNX * DeltaSubNX + DeltaSubNY * DeltaSubNY;
            if (normSquared > 256.0)
                break;
            if (normSquared < DeltaNormSquared || RefIteration >= count - 1) {
                DeltaSubNX = tempZX;
                DeltaSubNY = tempZY;
                DeltaNormSquared = normSquared;
                RefIteration = 0;
            }
            ++iter;
        }
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, 0, c_la, c_pt, c_px);
}

// ------------------------------------------------------------------------------------------------
// Host-callable launchers (called from renderer.cpp through kernels.h).
static dim3 frame_grid(const FsFrame &f) { return dim3((f.width + 63) / 64, (f.local_rows + 3) / 4, 1); }
static dim3 tile_grid(const FsFrame &f) { return dim3((f.width + 31) / 32, (f.local_rows + 7) / 8, 1); } // tile_pixel()

void fsk_prepare_orbit_hdr32(const fs_orbit_hdr32 *in, float4 *out, uint64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_prepare_orbit_hdr32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
}

void fsk_make_quiet_orbit(const float4 *zref, float4 *zq, float2 *zs2, float4 *zqb, uint64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_make_quiet_orbit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, zref, zq, zs2, zqb, n);
}

static unsigned lds_pad()
{
    static const unsigned v = [] {
        const char *e = getenv("FSMI355_LDS_PAD"); // occupancy-cap experiment: dynamic LDS bytes
        const unsigned v = e ? (unsigned)atoi(e) : 0u;
        return v <= 65536u ? v : 0u;
    }();
    return v;
}

static unsigned lav2_block_size()
{
    static const unsigned bs = [] {
        const char *e = getenv("FSMI355_BLOCK"); // launch-shape experiment: 64, 128 or 256
        const unsigned v = e ? (unsigned)atoi(e) : 256u;
        return v == 64u || v == 128u ? v : 256u;
    }();
    return bs;
}

uint32_t fsk_lav2_hdr32_slots(const FsFrame &f)
{
    if (lav2_block_size() != 256u)
        return 0;
    return ((f.width + 31u) / 32u) * 4u * ((f.local_rows + 7u) / 8u);
}

void fsk_lav2_hdr32(const FsLav2Args32 &A, int mode, bool stats, int variant, hipStream_t s)
{
    const unsigned pad = lds_pad();
    const unsigned bs = lav2_block_size();
    // A/B flag of fs_set_kernel_variant: the wave-uniform scaled runs' orbit entries through LDS (see the kernel)
    const bool lds_orbit = (variant & FS_VARIANT_FLAG_LDS_ORBIT) != 0;
    variant &= FS_VARIANT_BASE_MASK;
    const dim3 b(bs), g((A.frame.width + bs / 8 - 1) / (bs / 8), (A.frame.local_rows + 7) / 8, 1);
#define FS_LAUNCH_FAST(M, SC, LDS)                                                                                  \
    if (stats) {                                                                                                    \
        if (gs)                                                                                                     \
            hipLaunchKernelGGL((k_lav2_hdr32_fast<M, true, SC, LDS, true>), g, b, pad, s, A);                       \
        else                                                                                                        \
            hipLaunchKernelGGL((k_lav2_hdr32_fast<M, true, SC, LDS, false>), g, b, pad, s, A);                      \
    } else {                                                                                                        \
        if (gs)                                                                                                     \
            hipLaunchKernelGGL((k_lav2_hdr32_fast<M, false, SC, LDS, true>), g, b, pad, s, A);                      \
        else                                                                                                        \
            hipLaunchKernelGGL((k_lav2_hdr32_fast<M, false, SC, LDS, false>), g, b, pad, s, A);                     \
    }
#define FS_LAUNCH(M)                                                                                                \
    do {                                                                                                            \
        if (variant == FS_VARIANT_LITERAL) {                                                                        \
            if (stats)                                                                                              \
                hipLaunchKernelGGL((k_lav2_lit<float, M, true>), g, b, 0, s, A);                                         \
            else                                                                                                    \
                hipLaunchKernelGGL((k_lav2_lit<float, M, false>), g, b, 0, s, A);                                        \
        } else {                                                                                                    \
            const bool gs = A.parity == FS_PARITY_GPUSTAGE;                                                         \
            if (variant == FS_VARIANT_TUNED_NOSCALE) {                                                              \
                FS_LAUNCH_FAST(M, false, false)                                                                     \
            } else if (lds_orbit) {                                                                                 \
                FS_LAUNCH_FAST(M, true, true)                                                                       \
            } else {                                                                                                \
                FS_LAUNCH_FAST(M, true, false)                                                                      \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
    if (mode == FS_MODE_FULL)
        FS_LAUNCH(FS_MODE_FULL);
    else if (mode == FS_MODE_PO)
        FS_LAUNCH(FS_MODE_PO);
    else
        FS_LAUNCH(FS_MODE_LAO);
#undef FS_LAUNCH
#undef FS_LAUNCH_FAST
}

// Grid of the persistent (lane-refilling) launch: as many workgroups as the device holds at once, never more than one
// wave per tile.  The pixel queue counter is zeroed on the stream right before the launch.
template <class K> static dim3 persistent_grid(K kernel, const FsFrame &f)
{
    int dev = 0, cus = 256, per_cu = 2;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1)
        per_cu = 2;
    if (const char *e = getenv("FSMI355_PERSIST_PER_CU")) // launch-shape experiment
        per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
    const uint64_t tiles = (uint64_t)((f.width + 7u) >> 3) * ((f.local_rows + 7u) >> 3);
    uint64_t blocks = (uint64_t)cus * (uint64_t)per_cu;
    const uint64_t need = (tiles + 3u) / 4u;
    if (blocks > need)
        blocks = need;
    return dim3((unsigned)(blocks ? blocks : 1u), 1, 1);
}

// FS_VARIANT_FLAG_REFILL (fs_set_kernel_variant; A/B) selects the persistent, lane-refilling launch.
// It is OFF by default: measured on C5 (7680x4320) it raises the loop's lane utilisation from 0.74 to 0.98 and still
// loses, 333 ms against 263 ms -- re-packed lanes are due for different actions (jump / step) and sit at unrelated orbit
// and table positions: rocprofv3 counts 1.8x the vector instructions at 37 % active lanes with the reference-shaped
// loop (505 ms), and with the action loop, which removes that divergence, 2.5x the L2 requests remain (every per-lane
// load of a wave touches 64 different lines).
template <class F>
static void launch_perturb_scalar(const FsBlaArgsT<F> &A_in, bool use_bla, bool stats, int variant, hipStream_t s)
{
    const FsBlaArgsT<F> &A = A_in;
    // a probe launch covers one lane per tile of the frame
    const uint32_t ptx = (A.frame.width + 7u) >> 3, pty = (A.frame.local_rows + 7u) >> 3;
    const dim3 g = A.probe_out ? dim3((ptx + 31) / 32, (pty + 7) / 8, 1) : tile_grid(A.frame), b(256);
    if (A.frame.wide != 0u) { // iteration cap of 2^32 or above: the instantiation that counts in 64 bits
        if (use_bla) {
            if (stats)
                hipLaunchKernelGGL((k_perturb_scalar<F, true, true, false, false, uint64_t>), g, b, 0, s, A);
            else
                hipLaunchKernelGGL((k_perturb_scalar<F, true, false, false, false, uint64_t>), g, b, 0, s, A);
        } else {
            if (stats)
                hipLaunchKernelGGL((k_perturb_scalar<F, false, true, false, false, uint64_t>), g, b, 0, s, A);
            else
                hipLaunchKernelGGL((k_perturb_scalar<F, false, false, false, false, uint64_t>), g, b, 0, s, A);
        }
        return;
    }
    if (use_bla && (variant & FS_VARIANT_FLAG_REFILL) != 0) {
        (void)hipMemsetAsync(A.queue, 0, sizeof(uint32_t), s);
        if (stats)
            hipLaunchKernelGGL((k_perturb_scalar<F, true, true, true>),
                               persistent_grid(k_perturb_scalar<F, true, true, true>, A.frame), b, 0, s, A);
        else
            hipLaunchKernelGGL((k_perturb_scalar<F, true, false, true>),
                               persistent_grid(k_perturb_scalar<F, true, false, true>, A.frame), b, 0, s, A);
    } else if (use_bla) {
        if constexpr (std::is_same<F, float>::value) {
            if (A.nrec != nullptr && (variant & FS_VARIANT_BASE_MASK) != FS_VARIANT_LITERAL) {
                if (stats)
                    hipLaunchKernelGGL((k_perturb_scalar<F, true, true, false, true>), g, b, 0, s, A);
                else
                    hipLaunchKernelGGL((k_perturb_scalar<F, true, false, false, true>), g, b, 0, s, A);
                return;
            }
        }
        if (stats)
            hipLaunchKernelGGL((k_perturb_scalar<F, true, true, false>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_perturb_scalar<F, true, false, false>), g, b, 0, s, A);
    } else {
        if (stats)
            hipLaunchKernelGGL((k_perturb_scalar<F, false, true, false>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_perturb_scalar<F, false, false, false>), g, b, 0, s, A);
    }
}

void fsk_perturb_scalar_hdr32(const FsBlaArgs32 &A, bool use_bla, bool stats, int variant, hipStream_t s)
{
    // the default BLA frame: the hand-written kernel (kernels_bla_fast.hip).  The compiled kernel below keeps the step-counting
    // launches, the 64-bit counters, the refill variant, probes, and variants 1 / 2 (A/B references).
    if (use_bla && !stats && A.hrec != nullptr && A.frame.wide == 0u && (variant & FS_VARIANT_BASE_MASK) == FS_VARIANT_TUNED &&
        (variant & FS_VARIANT_FLAG_REFILL) == 0 && A.probe_out == nullptr &&
        A.tile_order == nullptr && A.frame.iter_u64 == 0u) {
        fsk_bla_hdr32_fast(A, (variant & FS_VARIANT_FLAG_BLA_POOL) != 0, s);
        return;
    }
    launch_perturb_scalar<float>(A, use_bla, stats, variant, s);
}

__global__ void k_decompress_orbit_hdr64(const fs_orbit_hdr64_rc *__restrict__ wp, uint64_t n_wp, uint64_t n_uncompressed,
                                         fs_real_hdr64 cxLow, fs_real_hdr64 cyLow, FsZ64 *__restrict__ out)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_wp)
        return;
    const uint64_t kIndexMask = 0x7FFFFFFFFFFFFFFFull;
    const uint64_t i0 = wp[k].index_and_rebase & kIndexMask;
    const uint64_t i1 = k + 1 < n_wp ? (wp[k + 1].index_and_rebase & kIndexMask) : n_uncompressed;
    hreal64 zx{wp[k].mx, wp[k].ex}, zy{wp[k].my, wp[k].ey};
    const hreal64 cx = ldr(cxLow), cy = ldr(cyLow);
    const hreal64 Two{1.0, 1};
    for (uint64_t i = i0; i < i1 && i < n_uncompressed; i++) {
        const hcplx64 c = hc_from_hr(zx, zy);
        FsZ64 z;
        z.re = c.re;
        z.im = c.im;
        z.e = c.e;
        z.pad_ = 0;
        z.w = ldexp(1.0, 8 - 2 * (c.e < -500 ? -500 : c.e));
        out[i] = z;
        const hreal64 zx_old = zx;
        zx = hr_add(hr_sub(hr_mul(zx, zx), hr_mul(zy, zy)), cx);
        hr_reduce(zx);
        zy = hr_add(hr_mul(hr_mul(Two, zx_old), zy), cy);
        hr_reduce(zy);
    }
}

void fsk_decompress_orbit_hdr64(const fs_orbit_hdr64_rc *wp, uint64_t n_wp, uint64_t n_uncompressed, fs_real_hdr64 cxLow,
                                fs_real_hdr64 cyLow, FsZ64 *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_decompress_orbit_hdr64, dim3((unsigned)((n_wp + 63) / 64)), dim3(64), 0, s, wp, n_wp,
                       n_uncompressed, cxLow, cyLow, out);
}

void fsk_decompress_orbit_hdr32(const fs_orbit_hdr32_rc *wp, uint64_t n_wp, uint64_t n_uncompressed, fs_real_hdr32 cxLow,
                                fs_real_hdr32 cyLow, float4 *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_decompress_orbit_hdr32, dim3((unsigned)((n_wp + 63) / 64)), dim3(64), 0, s, wp, n_wp,
                       n_uncompressed, cxLow, cyLow, out);
}

void fsk_prepare_orbit_hdr64(const fs_orbit_hdr64 *in, FsZ64 *out, uint64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_prepare_orbit_hdr64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
}

// PerformAT of every pixel in a pass of its own (see FsLav2ArgsT::at_res): the same pixel delta, the same isValid test, the same
// at_perform and the same dz = z * InvZCoeff as k_lav2_lit<double> -- stored instead of used.
__global__ void __launch_bounds__(256) k_at_pass64(FsLav2ArgsT<double> A)
{
    using F = double;
    uint32_t X, L;
    if (A.pixel_order)
        ordered_pixel(A.frame, A.pixel_order, X, L);
    else
        tile_pixel(X, L);
    const bool in_buffer = X < A.frame.width && L < A.frame.local_rows;
    const uint32_t Y = in_buffer ? global_row(A.frame, L) : 0xFFFFFFFFu;
    if (!(in_buffer && Y < A.frame.height))
        return;
    hreal<F> deltaReal, deltaImaginary;
    pixel_delta<F>(A.coords, X, Y, deltaReal, deltaImaginary);
    const hcplx<F> DeltaSub0 = hc_from_hr(deltaReal, deltaImaginary);
    FsAtRes out{0.0, 0.0, 0, 0xFFFFFFFFu};
    uint32_t own = 0;
    if (A.la_valid && A.use_at && hr_cmp_pos(hc_cheb(DeltaSub0), ldr(A.at.ThresholdC)) <= 0) {
        const uint32_t ATMaxIt = A.n_iterations / A.at.StepLength;
        hcplx<F> c = hc_add(hc_mul(DeltaSub0, ldc(A.at.CCoeff)), ldc(A.at.RefC));
        hc_reduce(c);
        hcplx<F> z;
        uint32_t i, i_exec = 0, i_own = 0;
        at_perform<F, uint32_t>(c, ldr(A.at.SqrEscapeRadius), ATMaxIt, z, i, &i_exec, &i_own);
        hcplx<F> dz = hc_mul(z, ldc(A.at.InvZCoeff));
        hc_reduce(dz);
        out = FsAtRes{dz.re, dz.im, dz.e, i};
        own = i_own;
    }
    const size_t idx = (size_t)L * A.frame.rounded_width + X;
    A.at_res[idx] = out;
    if (A.at_cost)
        A.at_cost[idx] = own;
    if (A.pixel_cost) // the frame's own order (round 6): the AT iteration count is the leading part of the pixel's final count
        A.pixel_cost[idx] = out.i == 0xFFFFFFFFu ? 0u : out.i;
}

void fsk_at_pass64(const FsLav2ArgsT<double> &A, hipStream_t s)
{
    hipLaunchKernelGGL(k_at_pass64, tile_grid(A.frame), dim3(256), 0, s, A);
}

void fsk_lav2_hdr64(const FsLav2ArgsT<double> &A, int mode, bool stats, hipStream_t s)
{
    const dim3 g = tile_grid(A.frame), b(256);
#define FS_LAUNCH64(M)                                                                                              \
    do {                                                                                                            \
        if (stats)                                                                                                  \
            hipLaunchKernelGGL((k_lav2_lit<double, M, true>), g, b, 0, s, A);                                       \
        else                                                                                                        \
            hipLaunchKernelGGL((k_lav2_lit<double, M, false>), g, b, 0, s, A);                                      \
    } while (0)
    if (mode == FS_MODE_FULL)
        FS_LAUNCH64(FS_MODE_FULL);
    else if (mode == FS_MODE_PO)
        FS_LAUNCH64(FS_MODE_PO);
    else
        FS_LAUNCH64(FS_MODE_LAO);
#undef FS_LAUNCH64
}

void fsk_lav2_wide(const FsLav2Args32 *A32, const FsLav2ArgsT<double> *A64, int mode, bool stats, hipStream_t s)
{
    const dim3 b(256), g = tile_grid(A32 ? A32->frame : A64->frame);
#define FS_LAUNCH_WIDE(M)                                                                                           \
    do {                                                                                                            \
        if (A32) {                                                                                                  \
            if (stats)                                                                                              \
                hipLaunchKernelGGL((k_lav2_lit<float, M, true, uint64_t>), g, b, 0, s, *A32);                       \
            else                                                                                                    \
                hipLaunchKernelGGL((k_lav2_lit<float, M, false, uint64_t>), g, b, 0, s, *A32);                      \
        } else {                                                                                                    \
            if (stats)                                                                                              \
                hipLaunchKernelGGL((k_lav2_lit<double, M, true, uint64_t>), g, b, 0, s, *A64);                      \
            else                                                                                                    \
                hipLaunchKernelGGL((k_lav2_lit<double, M, false, uint64_t>), g, b, 0, s, *A64);                     \
        }                                                                                                           \
    } while (0)
    if (mode == FS_MODE_FULL)
        FS_LAUNCH_WIDE(FS_MODE_FULL);
    else if (mode == FS_MODE_PO)
        FS_LAUNCH_WIDE(FS_MODE_PO);
    else
        FS_LAUNCH_WIDE(FS_MODE_LAO);
#undef FS_LAUNCH_WIDE
}

void fsk_lav2_seq(const FsLav2Args32 *A32, const FsLav2ArgsT<double> *A64, int mode, bool stats, hipStream_t s)
{
    const FsFrame &f = A32 ? A32->frame : A64->frame;
    const dim3 g = tile_grid(f), b(256);
    const bool wide = (A32 ? A32->frame.wide : A64->frame.wide) != 0u;
#define FS_LAUNCH_SEQ(M)                                                                                             \
    do {                                                                                                            \
        if (wide) {                                                                                                 \
            if (A32)                                                                                                \
                hipLaunchKernelGGL((k_lav2_lit<float, M, false, uint64_t, true>), g, b, 0, s, *A32);                 \
            else                                                                                                    \
                hipLaunchKernelGGL((k_lav2_lit<double, M, false, uint64_t, true>), g, b, 0, s, *A64);                \
        } else if (A32) {                                                                                           \
            if (stats)                                                                                              \
                hipLaunchKernelGGL((k_lav2_lit<float, M, true, uint32_t, true>), g, b, 0, s, *A32);                  \
            else                                                                                                    \
                hipLaunchKernelGGL((k_lav2_lit<float, M, false, uint32_t, true>), g, b, 0, s, *A32);                 \
        } else {                                                                                                    \
            if (stats)                                                                                              \
                hipLaunchKernelGGL((k_lav2_lit<double, M, true, uint32_t, true>), g, b, 0, s, *A64);                 \
            else                                                                                                    \
                hipLaunchKernelGGL((k_lav2_lit<double, M, false, uint32_t, true>), g, b, 0, s, *A64);                \
        }                                                                                                           \
    } while (0)
    if (mode == FS_MODE_FULL)
        FS_LAUNCH_SEQ(FS_MODE_FULL);
    else if (mode == FS_MODE_PO)
        FS_LAUNCH_SEQ(FS_MODE_PO);
    else
        FS_LAUNCH_SEQ(FS_MODE_LAO);
#undef FS_LAUNCH_SEQ
}

void fsk_seq_cursor_probe(bool is64, bool wide_pos, const void *wp, uint32_t n_wp, const void *cx, const void *cy,
                          uint64_t start, uint32_t n, void *out, hipStream_t s)
{
    if (is64) {
        const fs_real_hdr64 *x = (const fs_real_hdr64 *)cx, *y = (const fs_real_hdr64 *)cy;
        const hreal<double> hx{x->m, x->e}, hy{y->m, y->e};
        if (wide_pos)
            hipLaunchKernelGGL((k_seq_cursor_probe<double, uint64_t>), dim3(1), dim3(64), 0, s, wp, n_wp, hx, hy, start, n,
                               (hcplx<double> *)out);
        else
            hipLaunchKernelGGL((k_seq_cursor_probe<double, uint32_t>), dim3(1), dim3(64), 0, s, wp, n_wp, hx, hy, start, n,
                               (hcplx<double> *)out);
    } else {
        const fs_real_hdr32 *x = (const fs_real_hdr32 *)cx, *y = (const fs_real_hdr32 *)cy;
        const hreal<float> hx{x->m, x->e}, hy{y->m, y->e};
        if (wide_pos)
            hipLaunchKernelGGL((k_seq_cursor_probe<float, uint64_t>), dim3(1), dim3(64), 0, s, wp, n_wp, hx, hy, start, n,
                               (hcplx<float> *)out);
        else
            hipLaunchKernelGGL((k_seq_cursor_probe<float, uint32_t>), dim3(1), dim3(64), 0, s, wp, n_wp, hx, hy, start, n,
                               (hcplx<float> *)out);
    }
}

void fsk_perturb_scalar_hdr64(const FsBlaArgsT<double> &A, bool use_bla, bool stats, int variant, hipStream_t s)
{
    launch_perturb_scalar<double>(A, use_bla, stats, variant, s);
}

void fsk_perturb_bla_f64(const FsBlaArgsF64 &A, bool use_bla, bool stats, hipStream_t s)
{
    const dim3 g = frame_grid(A.frame), b(256);
    if (A.frame.wide != 0u) { // 64-bit counting (iteration caps of 2^32 and above)
        if (use_bla)
            hipLaunchKernelGGL((k_perturb_bla_f64<true, false, uint64_t>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_perturb_bla_f64<false, false, uint64_t>), g, b, 0, s, A);
        return;
    }
    if (use_bla) {
        if (stats)
            hipLaunchKernelGGL((k_perturb_bla_f64<true, true>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_perturb_bla_f64<true, false>), g, b, 0, s, A);
    } else {
        if (stats)
            hipLaunchKernelGGL((k_perturb_bla_f64<false, true>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_perturb_bla_f64<false, false>), g, b, 0, s, A);
    }
}

template <class F>
static void launch_direct_hdr(const FsDirectHdrArgsT<F> &A, hreal<F> minX, hreal<F> dx, bool stats, hipStream_t s)
{
    hipLaunchKernelGGL((k_direct_row_prefix_hdr<F>), dim3(1), dim3(64), 0, s, minX, dx, A.frame.width, A.cx_row);
    const dim3 g = frame_grid(A.frame), b(256);
    if (A.frame.wide != 0u)
        hipLaunchKernelGGL((k_direct_hdr<F, false, uint64_t>), g, b, 0, s, A);
    else if (stats)
        hipLaunchKernelGGL((k_direct_hdr<F, true>), g, b, 0, s, A);
    else
        hipLaunchKernelGGL((k_direct_hdr<F, false>), g, b, 0, s, A);
}
void fsk_direct_hdr32(const FsDirectHdrArgsT<float> &A, hreal<float> minX, hreal<float> dx, bool stats, hipStream_t s)
{
    launch_direct_hdr<float>(A, minX, dx, stats, s);
}
void fsk_direct_hdr64(const FsDirectHdrArgsT<double> &A, hreal<double> minX, hreal<double> dx, bool stats, hipStream_t s)
{
    launch_direct_hdr<double>(A, minX, dx, stats, s);
}

void fsk_direct_f64(const FsDirectArgs64 &A, double minX, double dx, bool stats, hipStream_t s)
{
    hipLaunchKernelGGL(k_direct_row_prefix_f64, dim3(1), dim3(64), 0, s, minX, dx, A.frame.width, A.cx_row);
    const dim3 g = frame_grid(A.frame), b(256);
    if (A.frame.wide != 0u)
        hipLaunchKernelGGL((k_direct_f64<false, uint64_t>), g, b, 0, s, A);
    else if (stats)
        hipLaunchKernelGGL((k_direct_f64<true>), g, b, 0, s, A);
    else
        hipLaunchKernelGGL((k_direct_f64<false>), g, b, 0, s, A);
}

