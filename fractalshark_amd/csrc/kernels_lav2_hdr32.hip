// kernels_lav2_hdr32.hip -- LAv2 for T = HDRFloat<float>, the tuned kernel of the headline configuration (k_lav2_hdr32_fast) and its
// launcher.  Compiled with -ffp-contract=off (see hdr_math.hpp).  (Round 6: a translation unit of its own; the text is unchanged.)
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
                    // (round 6, as in k_lav2_hdr64: with dc CCoeff 120 binades and more below newDz ZCoeff in every lane of the wave
                    // plus_mutable returns its first operand, and the second product is not formed)
                    if (__builtin_amdgcn_ballot_w64(fs::clamp_exp(newDz.e + LAj->ZCoeff.e) - fs::clamp_exp(DeltaSub0.e + LAj->CCoeff.e) >=
                                                    fs::kExpDiffIgnored) == __builtin_amdgcn_ballot_w64(true))
                        DeltaSubN = hc_mul(newDz, ldc(LAj->ZCoeff));
                    else
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
// Host-callable launchers (called from renderer.cpp through kernels.h).
static dim3 tile_grid(const FsFrame &f) { return dim3((f.width + 31) / 32, (f.local_rows + 7) / 8, 1); } // tile_pixel()

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
            fsk_lav2_lit32(A, M, stats, g, b, s); /* k_lav2_lit<float> lives in kernels.hip */                      \
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

