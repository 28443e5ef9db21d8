// kernels.hip -- gfx950 device code of libfsmi355.so: orbit preparation, the waypoint cursor, the LITERAL LAv2 kernel (k_lav2_lit: the
// operation-by-operation A/B reference, 64-bit counters, waypoint-resident orbits), the AT pass, the direct kernels, the plain-double
// BLA kernel and their launchers.  Compiled with -ffp-contract=off (see hdr_math.hpp).  Round 6: the tuned HDRFloat<float> LAv2 kernel
// is in kernels_lav2_hdr32.hip, the scalar perturbation / BLA kernel in kernels_perturb.hip (what they share: scaled_runs.hpp), the
// HDRFloat<double> LAv2 kernel in kernels_hdr64.hip.
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
#include "scaled_runs.hpp"

using namespace fs;


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

// the literal HDRFloat<float> kernel (FS_VARIANT_LITERAL of fsk_lav2_hdr32, kernels_lav2_hdr32.hip), launch shape given by the caller
void fsk_lav2_lit32(const FsLav2Args32 &A, int mode, bool stats, dim3 g, dim3 b, hipStream_t s)
{
#define FS_LAUNCH_LIT32(M)                                                                                          \
    do {                                                                                                            \
        if (stats)                                                                                                  \
            hipLaunchKernelGGL((k_lav2_lit<float, M, true>), g, b, 0, s, A);                                        \
        else                                                                                                        \
            hipLaunchKernelGGL((k_lav2_lit<float, M, false>), g, b, 0, s, A);                                       \
    } while (0)
    if (mode == FS_MODE_FULL)
        FS_LAUNCH_LIT32(FS_MODE_FULL);
    else if (mode == FS_MODE_PO)
        FS_LAUNCH_LIT32(FS_MODE_PO);
    else
        FS_LAUNCH_LIT32(FS_MODE_LAO);
#undef FS_LAUNCH_LIT32
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

