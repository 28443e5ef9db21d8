// kernels_perturb.hip -- scalar-HDRFloat perturbation with optional BLA skipping (k_perturb_scalar: the perturbation-only kernel of
// configuration C2, the compiled BLA kernel that counts steps for C5, the HDRFloat<double> forms) and its launchers.  Compiled with
// -ffp-contract=off (see hdr_math.hpp).  (Round 6: a translation unit of its own; the text is unchanged.)
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
// Host-callable launchers (called from renderer.cpp through kernels.h).
static dim3 tile_grid(const FsFrame &f) { return dim3((f.width + 31) / 32, (f.local_rows + 7) / 8, 1); } // tile_pixel()

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


void fsk_perturb_scalar_hdr64(const FsBlaArgsT<double> &A, bool use_bla, bool stats, int variant, hipStream_t s)
{
    launch_perturb_scalar<double>(A, use_bla, stats, variant, s);
}
