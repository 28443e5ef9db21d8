// kernels_scaled.hip -- scaled perturbation, T = HDRFloat<float>: RenderAlgorithm GpuHDRx32PerturbedScaled
// (GPURenderer::RenderPerturbBLAScaled, GPU_Render.cu:1302-1376).  Compiled with -ffp-contract=off.
//
// Restates the reference's CUDA kernel mandel_1x_float_perturb_scaled<IterType, HDRFloat<float>>
// (FractalSharkGpuLib/ScaledKernels.cuh:3-239): the perturbation w = dz / S is iterated in plain binary32 against a
// binary32 copy of the orbit; when |w|^2 grows past sqrt(1e30), when the pixel rebases, or when the orbit entry is
// flagged `bad` (its binary32 form underflows), the step is finished in HDRFloat<float> and the scale S renewed.
// There is no CPU RenderAlgorithm for this algorithm and the reference's binary32 expressions are open to nvcc's FMA
// contraction, so the rounding behaviour is fixed here by convention: source order, one IEEE operation per operator.
// Checker: orc_gpu_scaled_hdr32 (oracle/cpu_ref.cpp), parity unpinned.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"
#include "kernels.h"
#include "kernel_common.hpp"

using namespace fs;

namespace {

using H = hreal<float>;

// HdrSqrt, HDRFloat.h:1358-1383 (binary32 sqrt is correctly rounded: hipcc's default)
__device__ __forceinline__ H hr_sqrt_dev(H a)
{
    const bool odd = (a.e & 1) != 0;
    return H{__builtin_sqrtf(odd ? 2.0f * a.m : a.m), odd ? (a.e - 1) / 2 : a.e / 2};
}
__device__ __forceinline__ H orb_x(const fs_orbit_hdr32_bad *__restrict__ o, uint32_t i) { return H{o[i].mx, o[i].ex}; }
__device__ __forceinline__ H orb_y(const fs_orbit_hdr32_bad *__restrict__ o, uint32_t i) { return H{o[i].my, o[i].ey}; }

// steps of the next run of the tuned kernel: 256 / 64 / 16, the most that every running lane still has before its
// iteration limit (three votes per run instead of a counter per step; the same scheme as scaled_run_length in scaled_runs.hpp)
__device__ __forceinline__ uint32_t scaled_run_length_dev(uint32_t left)
{
    if (__builtin_amdgcn_ballot_w64(left < 256u) == 0ull)
        return 256u;
    if (__builtin_amdgcn_ballot_w64(left < 64u) == 0ull)
        return 64u;
    return __builtin_amdgcn_ballot_w64(left < 16u) == 0ull ? 16u : 0u;
}

// MI355X shape of the loop (the arithmetic is the reference's, operation by operation -- there is no CPU twin to pin a
// re-association against): a wave covers an 8 x 8 pixel tile like the other perturbation kernels (neighbours in two
// dimensions stay on the same orbit entry longer than 64 pixels of a row do, so the per-lane 16-byte entry loads of a
// wave mostly hit one cache line); the binary32 orbit entry a step tests against is the entry the next step multiplies
// by, so it is carried in registers instead of being loaded twice (one load per binary32 step); the common outcome of a
// step ("none": no rebase, no rescale, no escape) is the fall-through, everything else is cold.
template <bool kStats, class IterT = uint32_t> __global__ void __launch_bounds__(256) k_scaled_hdr32(FsScaledArgs32 A)
{
    uint32_t X, L;
    tile_pixel(X, L);
    uint64_t c_rescale = 0, c_full = 0, c_float = 0, c_px = 0;
    const uint32_t Y = global_row(A.frame, L);
    const bool live = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        const uint32_t MaxRefIteration = A.orbit_count - 1;
        const fs_orbit_hdr32_bad *__restrict__ ot = A.orbit_t;
        const fs_orbit_f32_bad *__restrict__ of = A.orbit_f;
        IterT iter = 0;
        uint32_t RefIteration = 0;
        // :35-39  `dx * X`: the int becomes a float and goes through HDRFloat(T mant)
        H DeltaReal = hr_sub(hr_mul(A.coords.dx, hr_from_mant<float>((float)(int)X)), A.coords.centerX);
        hr_reduce(DeltaReal);
        H DeltaImaginary = hr_sub(hr_mul(hr_neg(A.coords.dy), hr_from_mant<float>((float)(int)Y)), A.coords.centerY);
        hr_reduce(DeltaImaginary);
        H S = hr_sqrt_dev(hr_add(hr_mul(DeltaReal, DeltaReal), hr_mul(DeltaImaginary, DeltaImaginary)));
        hr_reduce(S);
        float DeltaSub0DX = hr_to_native(hr_div(DeltaReal, S));
        float DeltaSub0DY = hr_to_native(hr_div(DeltaImaginary, S));
        float wX = 0.0f, wY = 0.0f;
        float s = hr_to_native(S);
        float twos = 2 * s;
        const float w2threshold = A.w2threshold;
        const H Two = hr_from_mant<float>(2.0f);

#define FS_RESCALE(NX, NY)                                                                                              \
    do {                                                                                                                \
        S = hr_sqrt_dev(hr_add(hr_mul((NX), (NX)), hr_mul((NY), (NY))));                                                \
        hr_reduce(S);                                                                                                   \
        s = hr_to_native(S);                                                                                            \
        twos = 2 * s;                                                                                                   \
        DeltaSub0DX = hr_to_native(hr_div(DeltaReal, S));                                                               \
        DeltaSub0DY = hr_to_native(hr_div(DeltaImaginary, S));                                                          \
        wX = hr_to_native(hr_div((NX), S));                                                                             \
        wY = hr_to_native(hr_div((NY), S));                                                                             \
    } while (0)

        fs_orbit_f32_bad cf = of[0];
        uint32_t cf_at = 0; // orbit index cf was loaded from
        while (iter < n_iterations) {
            if (cf_at != RefIteration) {
                cf = of[RefIteration];
                cf_at = RefIteration;
            }
            if (cf.bad == 0) {
                // :78-94 binary32 step
                const float ox = wX, oy = wY;
                wX = ox * cf.x * 2 - oy * cf.y * 2 + s * ox * ox - s * oy * oy + DeltaSub0DX;
                wY = ox * (cf.y * 2 + twos * oy) + oy * cf.x * 2 + DeltaSub0DY;
                if (kStats)
                    c_float++;
                ++RefIteration;
                const fs_orbit_f32_bad nf = of[RefIteration];
                cf = nf; // the entry of the next step (unless this step rebases: cf_at then no longer matches)
                cf_at = RefIteration;
                const float tempZX = nf.x + wX * s;
                const float tempZY = nf.y + wY * s;
                const float zn_size = tempZX * tempZX + tempZY * tempZY;
                const float w2 = wX * wX + wY * wY;
                const float normDeltaSubN = w2 * s * s;
                const bool zn_size_OK = zn_size < 256.0f;
                const bool test1a = zn_size < normDeltaSubN;
                const bool test1b = RefIteration == MaxRefIteration;
                const bool test1ab = test1a || (test1b && zn_size_OK);
                const bool testw2 = (w2 >= w2threshold) && zn_size_OK;
                const bool none = !test1ab && !testw2 && zn_size_OK;
                if (none) {
                    ++iter;
                    continue;
                } else if (test1ab) {
                    const H ZX = hr_add(orb_x(ot, RefIteration), hr_mul(hr_from_mant<float>(wX), S));
                    const H ZY = hr_add(orb_y(ot, RefIteration), hr_mul(hr_from_mant<float>(wY), S));
                    RefIteration = 0;
                    FS_RESCALE(ZX, ZY);
                    if (kStats)
                        c_rescale++;
                    ++iter;
                    continue;
                } else if (testw2) {
                    const H ZX = hr_mul(hr_from_mant<float>(wX), S);
                    const H ZY = hr_mul(hr_from_mant<float>(wY), S);
                    FS_RESCALE(ZX, ZY);
                    if (kStats)
                        c_rescale++;
                    ++iter;
                    continue;
                } else {
                    break;
                }
            } else {
                // :160-233 the whole step in T
                const H ox = hr_from_mant<float>(wX), oy = hr_from_mant<float>(wY);
                const H cxr = orb_x(ot, RefIteration), cyr = orb_y(ot, RefIteration);
                H nX = hr_mul(hr_mul(ox, cxr), Two);
                nX = hr_sub(nX, hr_mul(hr_mul(oy, cyr), Two));
                nX = hr_add(nX, hr_mul(hr_mul(S, ox), ox));
                nX = hr_sub(nX, hr_mul(hr_mul(S, oy), oy));
                nX = hr_add(nX, hr_div(DeltaReal, S));
                hr_reduce(nX);
                H nY = hr_mul(ox, hr_add(hr_mul(cyr, Two), hr_mul(hr_mul(hr_from_number<float>(2.0f), S), oy)));
                nY = hr_add(nY, hr_mul(hr_mul(oy, cxr), Two));
                nY = hr_add(nY, hr_div(DeltaImaginary, S));
                hr_reduce(nY);
                if (kStats)
                    c_full++;
                ++RefIteration;
                const H tempZX = hr_add(orb_x(ot, RefIteration), hr_mul(nX, S));
                const H tempZY = hr_add(orb_y(ot, RefIteration), hr_mul(nY, S));
                H zn_size = hr_add(hr_mul(tempZX, tempZX), hr_mul(tempZY, tempZY));
                hr_reduce(zn_size);
                // !HdrCompareToBothPositiveReducedLT<T,256>(zn_size), HDRFloat.h:1169-1184
                const bool below = zn_size.e < 1 || (zn_size.e == 1 && !(zn_size.m >= 256.0f));
                if (!below)
                    break;
                const H TwoS = hr_mul(S, S);
                H normDeltaSubN = hr_add(hr_mul(hr_mul(nX, nX), TwoS), hr_mul(hr_mul(nY, nY), TwoS));
                hr_reduce(normDeltaSubN);
                H NewX, NewY;
                if (hr_cmp_pos(zn_size, normDeltaSubN) < 0 || RefIteration == MaxRefIteration) {
                    NewX = hr_add(orb_x(ot, RefIteration), hr_mul(nX, S));
                    NewY = hr_add(orb_y(ot, RefIteration), hr_mul(nY, S));
                    RefIteration = 0;
                } else {
                    NewX = hr_mul(nX, S);
                    NewY = hr_mul(nY, S);
                }
                FS_RESCALE(NewX, NewY);
            }
            ++iter;
        }
#undef FS_RESCALE
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, c_rescale, c_full, c_float, c_px);
}

// One wave-voted run of binary32 steps (see the tuned kernel below): from orbit entry `ref` (whose binary32 value is
// cfx, cfy) at most run_len steps, stopping before the first step that ANY running lane of the wave fails.  Returns the
// number of steps taken; w and the entry values are advanced.  Every running lane of the wave must call it together.
__device__ __forceinline__ uint32_t scaled_run(const fs_orbit_f32_bad *__restrict__ of, uint32_t ref, uint32_t run_len,
                                               float &wX, float &wY, float &cfx, float &cfy, float s, float twos, float dcX,
                                               float dcY)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const uint32_t lane_off = (ref + 1u) * 16u;
    const f4 *zp = (const f4 *)of;
    uint32_t c = 0;
    f2 o = {wX, wY}, e = {cfx, cfy};
    const f2 two2 = {2.0f, 2.0f}, s2 = {s, s}, dc2 = {dcX, dcY};
#define FS_SC_LOAD(OFS, T, PIN)                                                                                     \
    asm volatile("global_load_dwordx4 %0, %2, %3 offset:" OFS : "=v"(ent_##T), "+v"(PIN) : "v"(lane_off), "s"(zp));
    // One step from O against entry E into N; WAIT orders the arrival entry's load.  The reference's expressions,
    // operation by operation, with the operations that come in pairs issued as packed ones (the same IEEE operation on
    // each half):
    //   wX' = ((ox ex) 2 - (oy ey) 2 + (s ox) ox - (s oy) oy) + dX      wY' = (ox (ey 2 + twos oy) + (oy ex) 2) + dY
    //   P = (o * e) * 2 = (t1, t2),  Q = (s * o) * o = (t3, t4),  wX' = ((t1 - t2) + t3) - t4 + dX
#define FS_SC_STEP(O, E, N, T, WAIT)                                                                                 \
    const f2 P_##T = (O * E) * two2;                                                                                 \
    const f2 Q_##T = (s2 * O) * O;                                                                                   \
    const float c_##T = ((P_##T.x - P_##T.y) + Q_##T.x) - Q_##T.y;                                                   \
    const float u_##T = O.x * (E.y * 2 + twos * O.y) + O.y * E.x * 2;                                                \
    const f2 N = (f2){c_##T, u_##T} + dc2;                                                                           \
    float mx_##T = __builtin_fmaxf(__builtin_fabsf(N.x), __builtin_fabsf(N.y));                                      \
    WAIT;                                                                                                            \
    const bool ok_##T = mx_##T * s <= ent_##T.y && mx_##T < 0x1p24f;
    f4 ent_a, ent_b, ent_c, ent_d; // {bad, bound, x, y}
    for (;;) {
        FS_SC_LOAD("0", a, o.x)
        FS_SC_LOAD("16", b, o.x)
        FS_SC_LOAD("32", c, o.x)
        FS_SC_LOAD("48", d, o.x)
        FS_SC_STEP(o, e, n1, a, asm volatile("s_waitcnt vmcnt(3)" : "+v"(ent_a), "+v"(mx_a)))
        if (__builtin_amdgcn_ballot_w64(!ok_a) != 0ull)
            break;
        const f2 e1 = {ent_a.z, ent_a.w};
        FS_SC_STEP(n1, e1, n2, b, asm volatile("s_waitcnt vmcnt(2)" : "+v"(ent_b), "+v"(mx_b)))
        if (__builtin_amdgcn_ballot_w64(!ok_b) != 0ull) {
            o = n1, e = e1, c += 1;
            break;
        }
        const f2 e2 = {ent_b.z, ent_b.w};
        FS_SC_STEP(n2, e2, n3, c, asm volatile("s_waitcnt vmcnt(1)" : "+v"(ent_c), "+v"(mx_c)))
        if (__builtin_amdgcn_ballot_w64(!ok_c) != 0ull) {
            o = n2, e = e2, c += 2;
            break;
        }
        const f2 e3 = {ent_c.z, ent_c.w};
        FS_SC_STEP(n3, e3, n4, d, asm volatile("s_waitcnt vmcnt(0)" : "+v"(ent_d), "+v"(mx_d)))
        if (__builtin_amdgcn_ballot_w64(!ok_d) != 0ull) {
            o = n3, e = e3, c += 3;
            break;
        }
        o = n4, e = (f2){ent_d.z, ent_d.w}, c += 4;
        zp += 4;
        if (c >= run_len)
            break;
    }
    // a run that ends early leaves loads in flight: they land before anything else happens
    asm volatile("s_waitcnt vmcnt(0) ; scaled-kernel run, loop exit" ::"v"(ent_a), "v"(ent_b), "v"(ent_c), "v"(ent_d));
#undef FS_SC_STEP
#undef FS_SC_LOAD
    wX = o.x, wY = o.y, cfx = e.x, cfy = e.y;
    return c;
}

// ------------------------------------------------------------------------------------------------
// Tuned form of the same kernel (the default; the kernel above stays as FS_VARIANT_LITERAL, the A/B reference).
//
// The binary32 step's ARITHMETIC is executed exactly as above -- same operations, same order, nothing re-associated, so
// there is nothing new to pin -- but its five outcome tests (|z|^2 < 256, |z|^2 < |dz|^2, orbit end, w^2 over the
// threshold, next entry `bad`) are not evaluated when a cheaper sufficient condition proves that all of them come out as
// "none", which is what 98 % of the steps do.  With M = max(|Z'.x|, |Z'.y|) of the entry a step arrives at and
// m = max(|w'.x|, |w'.y|):
//     m * s <= M / 4   =>   |dz'| <= 0.354 |Z'|,  |z| = |Z' + dz'| in [0.646, 1.354] |Z'|
//                      =>   |z|^2 >= 3.3 |dz'|^2 (no rebase: the reference's float evaluation of both norms moves them by
//                           parts in 10^7; w2 * s * s cannot overflow: (m s) m <= 1.4 * 2^24), and
//                           |z|^2 < 115 for M < 5.6 (no escape, and zn_size_OK holds)
//     m < 2^24         =>   w2 = w.x^2 + w.y^2 < 2^49 < w2threshold (no rescale)
// The per-entry bound M / 4 sits in the padding word of the binary32 orbit entry (k_scaled_bounds, after the upload):
// -1 ("never") for an entry with M outside [2^-40, 5.6), for a `bad` entry (the NEXT step could not be a binary32 one) and
// for the last entry (the reference rebases there).  A NaN or infinity anywhere fails both comparisons.
// A run of such steps is wave-voted: it continues while EVERY running lane passes, the step that fails is dropped for the
// whole wave and every lane takes one step through the literal code, which decides exactly.  Lanes whose iteration limit
// is near are kept out of the runs by a vote on the steps left (runs of 256 / 64 / 16 steps, scaled_run_length in
// scaled_runs.hpp does the same).  The entry of a step is one 16-byte load per lane, requested four steps ahead from a
// wave-uniform base plus a per-lane byte offset that is fixed for the run.  (Entries through the scalar cache when the lanes
// share their orbit position, which pays in k_lav2_hdr32_fast, does not here: 325 vs 315 ms on View 14, whose runs
// average 37 steps -- profiles/patches/r02_t_*.)
template <bool kStats> __global__ void __launch_bounds__(256) k_scaled_hdr32_fast(FsScaledArgs32 A)
{
    uint32_t X, L;
    tile_pixel(X, L);
    uint64_t c_rescale = 0, c_full = 0, c_float = 0, c_px = 0, c_fast = 0, c_runs = 0;
    const uint32_t Y = global_row(A.frame, L);
    const bool live = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const uint32_t n_iterations = A.n_iterations;
        const uint32_t MaxRefIteration = A.orbit_count - 1;
        const fs_orbit_hdr32_bad *__restrict__ ot = A.orbit_t;
        const fs_orbit_f32_bad *__restrict__ of = A.orbit_f;
        uint32_t iter = 0, RefIteration = 0;
        H DeltaReal = hr_sub(hr_mul(A.coords.dx, hr_from_mant<float>((float)(int)X)), A.coords.centerX);
        hr_reduce(DeltaReal);
        H DeltaImaginary = hr_sub(hr_mul(hr_neg(A.coords.dy), hr_from_mant<float>((float)(int)Y)), A.coords.centerY);
        hr_reduce(DeltaImaginary);
        H S = hr_sqrt_dev(hr_add(hr_mul(DeltaReal, DeltaReal), hr_mul(DeltaImaginary, DeltaImaginary)));
        hr_reduce(S);
        float DeltaSub0DX = hr_to_native(hr_div(DeltaReal, S));
        float DeltaSub0DY = hr_to_native(hr_div(DeltaImaginary, S));
        float wX = 0.0f, wY = 0.0f;
        float s = hr_to_native(S);
        float twos = 2 * s;
        const float w2threshold = A.w2threshold;
        const H Two = hr_from_mant<float>(2.0f);

#define FS_RESCALE(NX, NY)                                                                                              \
    do {                                                                                                                \
        S = hr_sqrt_dev(hr_add(hr_mul((NX), (NX)), hr_mul((NY), (NY))));                                                \
        hr_reduce(S);                                                                                                   \
        s = hr_to_native(S);                                                                                            \
        twos = 2 * s;                                                                                                   \
        DeltaSub0DX = hr_to_native(hr_div(DeltaReal, S));                                                               \
        DeltaSub0DY = hr_to_native(hr_div(DeltaImaginary, S));                                                          \
        wX = hr_to_native(hr_div((NX), S));                                                                             \
        wY = hr_to_native(hr_div((NY), S));                                                                             \
    } while (0)

        float cfx = of[0].x, cfy = of[0].y; // the entry the next step multiplies by ...
        uint32_t cf_bad = of[0].bad;
        uint32_t cf_at = 0;                 // ... and the orbit index it was loaded from
        while (iter < n_iterations) {
            if (cf_at != RefIteration) {
                const fs_orbit_f32_bad e = of[RefIteration];
                cfx = e.x, cfy = e.y, cf_bad = e.bad;
                cf_at = RefIteration;
            }
            // ---- runs of binary32 steps whose tests are implied (see above)
            {
                const uint32_t run_len = scaled_run_length_dev(n_iterations - iter);
                if (__builtin_amdgcn_ballot_w64(cf_bad != 0u) == 0ull && run_len != 0u) {
                    const uint32_t c = scaled_run(of, RefIteration, run_len, wX, wY, cfx, cfy, s, twos, DeltaSub0DX, DeltaSub0DY);
                    if (c != 0u) {
                        cf_bad = 0u; // an entry a run arrived at is not `bad`
                        RefIteration += c;
                        cf_at = RefIteration;
                        iter += c;
                        if (kStats) {
                            c_float += c;
                            c_fast += c;
                            c_runs++;
                        }
                        if (iter >= n_iterations)
                            break;
                    }
                }
            }
            // ---- one step through the literal code
            if (cf_bad == 0) {
                const float ox = wX, oy = wY;
                wX = ox * cfx * 2 - oy * cfy * 2 + s * ox * ox - s * oy * oy + DeltaSub0DX;
                wY = ox * (cfy * 2 + twos * oy) + oy * cfx * 2 + DeltaSub0DY;
                if (kStats)
                    c_float++;
                ++RefIteration;
                const fs_orbit_f32_bad nf = of[RefIteration];
                cfx = nf.x, cfy = nf.y, cf_bad = nf.bad;
                cf_at = RefIteration;
                const float tempZX = nf.x + wX * s;
                const float tempZY = nf.y + wY * s;
                const float zn_size = tempZX * tempZX + tempZY * tempZY;
                const float w2 = wX * wX + wY * wY;
                const float normDeltaSubN = w2 * s * s;
                const bool zn_size_OK = zn_size < 256.0f;
                const bool test1a = zn_size < normDeltaSubN;
                const bool test1b = RefIteration == MaxRefIteration;
                const bool test1ab = test1a || (test1b && zn_size_OK);
                const bool testw2 = (w2 >= w2threshold) && zn_size_OK;
                const bool none = !test1ab && !testw2 && zn_size_OK;
                if (none) {
                    ++iter;
                    continue;
                } else if (test1ab) {
                    const H ZX = hr_add(orb_x(ot, RefIteration), hr_mul(hr_from_mant<float>(wX), S));
                    const H ZY = hr_add(orb_y(ot, RefIteration), hr_mul(hr_from_mant<float>(wY), S));
                    RefIteration = 0;
                    FS_RESCALE(ZX, ZY);
                    if (kStats)
                        c_rescale++;
                    ++iter;
                    continue;
                } else if (testw2) {
                    const H ZX = hr_mul(hr_from_mant<float>(wX), S);
                    const H ZY = hr_mul(hr_from_mant<float>(wY), S);
                    FS_RESCALE(ZX, ZY);
                    if (kStats)
                        c_rescale++;
                    ++iter;
                    continue;
                } else {
                    break;
                }
            } else {
                const H ox = hr_from_mant<float>(wX), oy = hr_from_mant<float>(wY);
                const H cxr = orb_x(ot, RefIteration), cyr = orb_y(ot, RefIteration);
                H nX = hr_mul(hr_mul(ox, cxr), Two);
                nX = hr_sub(nX, hr_mul(hr_mul(oy, cyr), Two));
                nX = hr_add(nX, hr_mul(hr_mul(S, ox), ox));
                nX = hr_sub(nX, hr_mul(hr_mul(S, oy), oy));
                nX = hr_add(nX, hr_div(DeltaReal, S));
                hr_reduce(nX);
                H nY = hr_mul(ox, hr_add(hr_mul(cyr, Two), hr_mul(hr_mul(hr_from_number<float>(2.0f), S), oy)));
                nY = hr_add(nY, hr_mul(hr_mul(oy, cxr), Two));
                nY = hr_add(nY, hr_div(DeltaImaginary, S));
                hr_reduce(nY);
                if (kStats)
                    c_full++;
                ++RefIteration;
                const H tempZX = hr_add(orb_x(ot, RefIteration), hr_mul(nX, S));
                const H tempZY = hr_add(orb_y(ot, RefIteration), hr_mul(nY, S));
                H zn_size = hr_add(hr_mul(tempZX, tempZX), hr_mul(tempZY, tempZY));
                hr_reduce(zn_size);
                const bool below = zn_size.e < 1 || (zn_size.e == 1 && !(zn_size.m >= 256.0f));
                if (!below)
                    break;
                const H TwoS = hr_mul(S, S);
                H normDeltaSubN = hr_add(hr_mul(hr_mul(nX, nX), TwoS), hr_mul(hr_mul(nY, nY), TwoS));
                hr_reduce(normDeltaSubN);
                H NewX, NewY;
                if (hr_cmp_pos(zn_size, normDeltaSubN) < 0 || RefIteration == MaxRefIteration) {
                    NewX = hr_add(orb_x(ot, RefIteration), hr_mul(nX, S));
                    NewY = hr_add(orb_y(ot, RefIteration), hr_mul(nY, S));
                    RefIteration = 0;
                } else {
                    NewX = hr_mul(nX, S);
                    NewY = hr_mul(nY, S);
                }
                FS_RESCALE(NewX, NewY);
            }
            ++iter;
        }
#undef FS_RESCALE
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats) {
        add_stats(A.stats, c_rescale, c_full, c_float, c_px);
        atomicAdd((unsigned long long *)&A.stats[6], (unsigned long long)c_fast); // probes (tools/scaled_kernel_probe.py):
        atomicAdd((unsigned long long *)&A.stats[7], (unsigned long long)c_runs); // lane-steps inside runs, runs
    }
}

// The bound word of the binary32 orbit entries (their padding field): M / 4, or -1 where no run may arrive.
__global__ void k_scaled_bounds(fs_orbit_f32_bad *__restrict__ of, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const fs_orbit_f32_bad e = of[i];
    const float m = __builtin_fmaxf(__builtin_fabsf(e.x), __builtin_fabsf(e.y));
    const bool usable = e.bad == 0u && i + 1 < n && m >= 0x1p-40f && m < 5.6f;
    of[i].padding = __float_as_uint(usable ? m * 0.25f : -1.0f);
}

// ------------------------------------------------------------------------------------------------
// The same kernel for T = double (RenderAlgorithm Gpu1x32PerturbedScaled): HdrReduce / HdrSqrt / the HDR comparisons
// collapse to plain double arithmetic (ScaledKernels.cuh:3-239 with T = double; HdrCompareToBothPositiveReducedLT<T,256>
// is `zn_size < 256.0`, HDRFloat.h:1584).
template <bool kStats, class IterT = uint32_t> __global__ void __launch_bounds__(256) k_scaled_f64(FsScaledArgsF64 A)
{
    const uint32_t X = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t L = blockIdx.y * 4u + (threadIdx.x >> 6);
    uint64_t c_rescale = 0, c_full = 0, c_float = 0, c_px = 0, c_fast = 0, c_runs = 0;
    const uint32_t Y = global_row(A.frame, L);
    const bool live = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        const uint32_t MaxRefIteration = A.orbit_count - 1;
        const fs_orbit_f64_bad *__restrict__ ot = A.orbit_t;
        const fs_orbit_f32_bad *__restrict__ of = A.orbit_f;
        IterT iter = 0;
        uint32_t RefIteration = 0;
        const double DeltaReal = A.dx * (double)(int)X - A.centerX;
        const double DeltaImaginary = -A.dy * (double)(int)Y - A.centerY;
        double S = __builtin_sqrt(DeltaReal * DeltaReal + DeltaImaginary * DeltaImaginary);
        float DeltaSub0DX = (float)(DeltaReal / S);
        float DeltaSub0DY = (float)(DeltaImaginary / S);
        float wX = 0.0f, wY = 0.0f;
        float s = (float)S;
        float twos = 2 * s;
        const float w2threshold = A.w2threshold;

#define FS_RESCALE64(NX, NY)                                                                                            \
    do {                                                                                                                \
        S = __builtin_sqrt((NX) * (NX) + (NY) * (NY));                                                                  \
        s = (float)S;                                                                                                   \
        twos = 2 * s;                                                                                                   \
        DeltaSub0DX = (float)(DeltaReal / S);                                                                           \
        DeltaSub0DY = (float)(DeltaImaginary / S);                                                                      \
        wX = (float)((NX) / S);                                                                                         \
        wY = (float)((NY) / S);                                                                                         \
    } while (0)

        while (iter < n_iterations) {
            const fs_orbit_f32_bad cf = of[RefIteration];
            if (cf.bad == 0) {
                const float ox = wX, oy = wY;
                wX = ox * cf.x * 2 - oy * cf.y * 2 + s * ox * ox - s * oy * oy + DeltaSub0DX;
                wY = ox * (cf.y * 2 + twos * oy) + oy * cf.x * 2 + DeltaSub0DY;
                if (kStats)
                    c_float++;
                ++RefIteration;
                const fs_orbit_f32_bad nf = of[RefIteration];
                const float tempZX = nf.x + wX * s;
                const float tempZY = nf.y + wY * s;
                const float zn_size = tempZX * tempZX + tempZY * tempZY;
                const float w2 = wX * wX + wY * wY;
                const float normDeltaSubN = w2 * s * s;
                const bool zn_size_OK = zn_size < 256.0f;
                const bool test1a = zn_size < normDeltaSubN;
                const bool test1b = RefIteration == MaxRefIteration;
                const bool test1ab = test1a || (test1b && zn_size_OK);
                const bool testw2 = (w2 >= w2threshold) && zn_size_OK;
                const bool none = !test1ab && !testw2 && zn_size_OK;
                if (none) {
                    ++iter;
                    continue;
                } else if (test1ab) {
                    const double ZX = ot[RefIteration].x + (double)wX * S;
                    const double ZY = ot[RefIteration].y + (double)wY * S;
                    RefIteration = 0;
                    FS_RESCALE64(ZX, ZY);
                    if (kStats)
                        c_rescale++;
                    ++iter;
                    continue;
                } else if (testw2) {
                    const double ZX = (double)wX * S;
                    const double ZY = (double)wY * S;
                    FS_RESCALE64(ZX, ZY);
                    if (kStats)
                        c_rescale++;
                    ++iter;
                    continue;
                } else {
                    break;
                }
            } else {
                const double ox = (double)wX, oy = (double)wY;
                const double cxr = ot[RefIteration].x, cyr = ot[RefIteration].y;
                double nX = ox * cxr * 2;
                nX -= oy * cyr * 2;
                nX += S * ox * ox;
                nX -= S * oy * oy;
                nX += DeltaReal / S;
                double nY = ox * (cyr * 2 + 2.0 * S * oy);
                nY += oy * cxr * 2;
                nY += DeltaImaginary / S;
                if (kStats)
                    c_full++;
                ++RefIteration;
                const double tempZX = ot[RefIteration].x + nX * S;
                const double tempZY = ot[RefIteration].y + nY * S;
                const double zn_size = tempZX * tempZX + tempZY * tempZY;
                if (!(zn_size < 256.0))
                    break;
                const double TwoS = S * S;
                const double normDeltaSubN = nX * nX * TwoS + nY * nY * TwoS;
                double NewX, NewY;
                if (zn_size < normDeltaSubN || RefIteration == MaxRefIteration) {
                    NewX = ot[RefIteration].x + nX * S;
                    NewY = ot[RefIteration].y + nY * S;
                    RefIteration = 0;
                } else {
                    NewX = nX * S;
                    NewY = nY * S;
                }
                FS_RESCALE64(NewX, NewY);
            }
            ++iter;
        }
#undef FS_RESCALE64
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, c_rescale, c_full, c_float, c_px);
}

// Tuned form of k_scaled_f64: the same wave-voted runs of binary32 steps as k_scaled_hdr32_fast (the binary32 step does not
// depend on T), 8 x 8 pixel tiles per wave; the literal kernel above stays as FS_VARIANT_LITERAL.
template <bool kStats> __global__ void __launch_bounds__(256) k_scaled_f64_fast(FsScaledArgsF64 A)
{
    uint32_t X, L;
    tile_pixel(X, L);
    uint64_t c_rescale = 0, c_full = 0, c_float = 0, c_px = 0, c_fast = 0, c_runs = 0;
    const uint32_t Y = global_row(A.frame, L);
    const bool live = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const uint32_t n_iterations = A.n_iterations;
        const uint32_t MaxRefIteration = A.orbit_count - 1;
        const fs_orbit_f64_bad *__restrict__ ot = A.orbit_t;
        const fs_orbit_f32_bad *__restrict__ of = A.orbit_f;
        uint32_t iter = 0, RefIteration = 0;
        const double DeltaReal = A.dx * (double)(int)X - A.centerX;
        const double DeltaImaginary = -A.dy * (double)(int)Y - A.centerY;
        double S = __builtin_sqrt(DeltaReal * DeltaReal + DeltaImaginary * DeltaImaginary);
        float DeltaSub0DX = (float)(DeltaReal / S);
        float DeltaSub0DY = (float)(DeltaImaginary / S);
        float wX = 0.0f, wY = 0.0f;
        float s = (float)S;
        float twos = 2 * s;
        const float w2threshold = A.w2threshold;

#define FS_RESCALE64(NX, NY)                                                                                            \
    do {                                                                                                                \
        S = __builtin_sqrt((NX) * (NX) + (NY) * (NY));                                                                  \
        s = (float)S;                                                                                                   \
        twos = 2 * s;                                                                                                   \
        DeltaSub0DX = (float)(DeltaReal / S);                                                                           \
        DeltaSub0DY = (float)(DeltaImaginary / S);                                                                      \
        wX = (float)((NX) / S);                                                                                         \
        wY = (float)((NY) / S);                                                                                         \
    } while (0)

        float cfx = of[0].x, cfy = of[0].y; // the entry the next step multiplies by ...
        uint32_t cf_bad = of[0].bad;
        uint32_t cf_at = 0;                 // ... and the orbit index it was loaded from
        while (iter < n_iterations) {
            if (cf_at != RefIteration) {
                const fs_orbit_f32_bad e = of[RefIteration];
                cfx = e.x, cfy = e.y, cf_bad = e.bad;
                cf_at = RefIteration;
            }
            // ---- runs of binary32 steps whose tests are implied (k_scaled_hdr32_fast)
            {
                const uint32_t run_len = scaled_run_length_dev(n_iterations - iter);
                if (__builtin_amdgcn_ballot_w64(cf_bad != 0u) == 0ull && run_len != 0u) {
                    const uint32_t c = scaled_run(of, RefIteration, run_len, wX, wY, cfx, cfy, s, twos, DeltaSub0DX, DeltaSub0DY);
                    if (c != 0u) {
                        cf_bad = 0u;
                        RefIteration += c;
                        cf_at = RefIteration;
                        iter += c;
                        if (kStats)
                            c_float += c;
                        if (iter >= n_iterations)
                            break;
                    }
                }
            }
            // ---- one step through the literal code
            if (cf_bad == 0) {
                const float ox = wX, oy = wY;
                wX = ox * cfx * 2 - oy * cfy * 2 + s * ox * ox - s * oy * oy + DeltaSub0DX;
                wY = ox * (cfy * 2 + twos * oy) + oy * cfx * 2 + DeltaSub0DY;
                if (kStats)
                    c_float++;
                ++RefIteration;
                const fs_orbit_f32_bad nf = of[RefIteration];
                cfx = nf.x, cfy = nf.y, cf_bad = nf.bad;
                cf_at = RefIteration;
                const float tempZX = nf.x + wX * s;
                const float tempZY = nf.y + wY * s;
                const float zn_size = tempZX * tempZX + tempZY * tempZY;
                const float w2 = wX * wX + wY * wY;
                const float normDeltaSubN = w2 * s * s;
                const bool zn_size_OK = zn_size < 256.0f;
                const bool test1a = zn_size < normDeltaSubN;
                const bool test1b = RefIteration == MaxRefIteration;
                const bool test1ab = test1a || (test1b && zn_size_OK);
                const bool testw2 = (w2 >= w2threshold) && zn_size_OK;
                const bool none = !test1ab && !testw2 && zn_size_OK;
                if (none) {
                    ++iter;
                    continue;
                } else if (test1ab) {
                    const double ZX = ot[RefIteration].x + (double)wX * S;
                    const double ZY = ot[RefIteration].y + (double)wY * S;
                    RefIteration = 0;
                    FS_RESCALE64(ZX, ZY);
                    if (kStats)
                        c_rescale++;
                    ++iter;
                    continue;
                } else if (testw2) {
                    const double ZX = (double)wX * S;
                    const double ZY = (double)wY * S;
                    FS_RESCALE64(ZX, ZY);
                    if (kStats)
                        c_rescale++;
                    ++iter;
                    continue;
                } else {
                    break;
                }
            } else {
                const double ox = (double)wX, oy = (double)wY;
                const double cxr = ot[RefIteration].x, cyr = ot[RefIteration].y;
                double nX = ox * cxr * 2;
                nX -= oy * cyr * 2;
                nX += S * ox * ox;
                nX -= S * oy * oy;
                nX += DeltaReal / S;
                double nY = ox * (cyr * 2 + 2.0 * S * oy);
                nY += oy * cxr * 2;
                nY += DeltaImaginary / S;
                if (kStats)
                    c_full++;
                ++RefIteration;
                const double tempZX = ot[RefIteration].x + nX * S;
                const double tempZY = ot[RefIteration].y + nY * S;
                const double zn_size = tempZX * tempZX + tempZY * tempZY;
                if (!(zn_size < 256.0))
                    break;
                const double TwoS = S * S;
                const double normDeltaSubN = nX * nX * TwoS + nY * nY * TwoS;
                double NewX, NewY;
                if (zn_size < normDeltaSubN || RefIteration == MaxRefIteration) {
                    NewX = ot[RefIteration].x + nX * S;
                    NewY = ot[RefIteration].y + nY * S;
                    RefIteration = 0;
                } else {
                    NewX = nX * S;
                    NewY = nY * S;
                }
                FS_RESCALE64(NewX, NewY);
            }
            ++iter;
        }
#undef FS_RESCALE64
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, c_rescale, c_full, c_float, c_px);
}

} // namespace

void fsk_scaled_hdr32(const FsScaledArgs32 &A, bool stats, int variant, hipStream_t s)
{
    const dim3 b(256);
    const dim3 g((A.frame.width + 31) / 32, (A.frame.local_rows + 7) / 8); // tile_pixel(): four 8 x 8 tiles per workgroup
    if (A.frame.wide != 0u) { // iteration cap of 2^32 or above: the literal kernel counting in 64 bits
        hipLaunchKernelGGL((k_scaled_hdr32<false, uint64_t>), g, b, 0, s, A);
        return;
    }
    if (variant == FS_VARIANT_LITERAL) {
        if (stats)
            hipLaunchKernelGGL((k_scaled_hdr32<true>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_scaled_hdr32<false>), g, b, 0, s, A);
    } else {
        if (stats)
            hipLaunchKernelGGL((k_scaled_hdr32_fast<true>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_scaled_hdr32_fast<false>), g, b, 0, s, A);
    }
}

void fsk_scaled_bounds(fs_orbit_f32_bad *of, uint64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_scaled_bounds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, of, n);
}

void fsk_scaled_f64(const FsScaledArgsF64 &A, bool stats, int variant, hipStream_t s)
{
    const dim3 b(256);
    if (A.frame.wide != 0u) {
        const dim3 g((A.frame.width + 63) / 64, (A.frame.local_rows + 3) / 4);
        hipLaunchKernelGGL((k_scaled_f64<false, uint64_t>), g, b, 0, s, A);
        return;
    }
    if (variant == FS_VARIANT_LITERAL) {
        const dim3 g((A.frame.width + 63) / 64, (A.frame.local_rows + 3) / 4);
        if (stats)
            hipLaunchKernelGGL((k_scaled_f64<true>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_scaled_f64<false>), g, b, 0, s, A);
    } else {
        const dim3 g((A.frame.width + 31) / 32, (A.frame.local_rows + 7) / 8); // tile_pixel()
        if (stats)
            hipLaunchKernelGGL((k_scaled_f64_fast<true>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_scaled_f64_fast<false>), g, b, 0, s, A);
    }
}
