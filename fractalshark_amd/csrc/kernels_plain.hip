// kernels_plain.hip -- LAv2 for the non-HDR numeric types: T = float (RenderAlgorithm Gpu1x32PerturbedLAv2[PO|LAO],
// GPU_Render.cu:1025-1043), double (Gpu1x64PerturbedLAv2*, :1077-1100) and CudaDblflt (Gpu2x32PerturbedLAv2*, :1044-1076).
// Fractal's AUTO mode renders zoom factors 1e4 .. 1e34 with the float kernel (Fractal.cpp:958-966).
// Compiled with -ffp-contract=off: every operation below is one IEEE operation, as written.
//
// No CPU RenderAlgorithm runs LAv2 on a plain T, so the semantics restated here are those of the CUDA kernel itself,
// mandel_1xHDR_float_perturb_lav2<IterType, T, T, Mode, PExtras> (FractalSharkGpuLib/LAKernel.cuh:3-315) with its
// `else` arms for non-HDR types: FloatComplex<T> arithmetic (FloatComplex.h:188-268,327-331,413-419), HdrReduce a no-op,
// compares by operator< / >= (HDRFloat.h:1536-1586; GPU_LAReference.h:238-254; GPU_LAInfoDeep.h:90-106), bailout
// |z|^2 < T(256), the AT shortcut of ATInfo.h:126-188.  The checker is oracle/gpu_ref_plain.cpp (parity unpinned: see
// its header).
//
// Decomposition: one lane per (sub)pixel, a wave covers an 8 x 8 pixel tile (kernel_common.hpp), the orbit and the LA
// table are read straight from L2 (the lanes of a wave read the same or neighbouring entries).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/fs_layout.h"
#include "df32_math.hpp"
#include "kernels.h"
#include "kernel_common.hpp"

using namespace fs;

namespace {

template <class T> struct Plain;
template <> struct Plain<float> {
    using Orbit = fs_orbit_f32;
    using LA = fs_la_f32_u32;
    using AT = fs_at_f32_u32;
    using Real = float;
    static __device__ __forceinline__ float from_int(int v) { return (float)v; }
};
template <> struct Plain<double> {
    using Orbit = fs_orbit_f64;
    using LA = fs_la_f64_u32;
    using AT = fs_at_f64_u32;
    using Real = double;
    static __device__ __forceinline__ double from_int(int v) { return (double)v; }
};
template <> struct Plain<df32> {
    using Orbit = fs_orbit_p2x32;
    using LA = fs_la_p2x32_u32;
    using AT = fs_at_p2x32_u32;
    using Real = fs_real_p2x32;
    // on the device only CudaDblflt(float) is viable for T(X) with an int X (CudaDblflt.h:52-68)
    static __device__ __forceinline__ df32 from_int(int v) { return df32((float)v); }
};

template <class T> struct cx {
    T re, im;
};
template <class T> __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return cx<T>{a.re + b.re, a.im + b.im}; }
// times_mutable(FloatComplex), FloatComplex.h:198-211
template <class T> __device__ __forceinline__ cx<T> operator*(cx<T> a, cx<T> b)
{
    const T re = (a.re * b.re) - (a.im * b.im);
    const T im = (a.re * b.im) + (a.im * b.re);
    return cx<T>{re, im};
}
template <class T> __device__ __forceinline__ cx<T> mul_real(cx<T> a, T f) { return cx<T>{a.re * f, a.im * f}; }
template <class T> __device__ __forceinline__ T norm2(cx<T> a) { return a.re * a.re + a.im * a.im; }
template <class T> __device__ __forceinline__ T cheb(cx<T> a)
{
    const T ar = fabs_bits<T>(a.re), ai = fabs_bits<T>(a.im);
    return ar > ai ? ar : ai;
}

// zx * T{2} + d (LAKernel.cuh:143-149).  For binary32 / binary64 the product by two is exact, so the fused form performs
// the same single rounding of 2z + d as the separate multiply and add: identical bits, one instruction.  (Orbit values
// are bounded by the bailout radius, far from overflow.)
template <class T> __device__ __forceinline__ T twice_plus(T z, T d, T Two) { return z * Two + d; }
template <> __device__ __forceinline__ float twice_plus<float>(float z, float d, float) { return __builtin_fmaf(z, 2.0f, d); }
template <> __device__ __forceinline__ double twice_plus<double>(double z, double d, double) { return __builtin_fma(z, 2.0, d); }

__device__ __forceinline__ float ld(float v) { return v; }
__device__ __forceinline__ double ld(double v) { return v; }
__device__ __forceinline__ df32 ld(const fs_real_p2x32 &v) { return df32(v.head, v.tail); }
__device__ __forceinline__ cx<float> ld(const fs_cplx_f32 &c) { return cx<float>{c.re, c.im}; }
__device__ __forceinline__ cx<double> ld(const fs_cplx_f64 &c) { return cx<double>{c.re, c.im}; }
__device__ __forceinline__ cx<df32> ld(const fs_cplx_p2x32 &c)
{
    return cx<df32>{df32(c.re_head, c.re_tail), df32(c.im_head, c.im_tail)};
}
__device__ __forceinline__ cx<float> ldz(const fs_orbit_f32 *__restrict__ o, uint32_t i)
{
    const float2 v = *reinterpret_cast<const float2 *>(o + i);
    return cx<float>{v.x, v.y};
}
__device__ __forceinline__ cx<double> ldz(const fs_orbit_f64 *__restrict__ o, uint32_t i)
{
    const double2 v = *reinterpret_cast<const double2 *>(o + i);
    return cx<double>{v.x, v.y};
}
__device__ __forceinline__ cx<df32> ldz(const fs_orbit_p2x32 *__restrict__ o, uint32_t i)
{
    const float4 v = *reinterpret_cast<const float4 *>(o + i);
    return cx<df32>{df32(v.x, v.y), df32(v.z, v.w)};
}

// Sequential access to a SimpleCompression orbit that stays compressed in HBM (fs_set_compressed_orbit_mode 1), T = float /
// double / CudaDblflt: the plain-type twin of SeqOrbit in kernels.hip (GPUPerturbSingleResults::SeqWorkspace / GetIterSeq /
// BinarySearch, Perturb.cuh:160-326).  seek() starts at the last waypoint at or before the index and iterates z = z^2 + c
// forward in T arithmetic, step() moves one index on: the next waypoint when its index comes up, one iteration otherwise --
// the operations, in the order, of k_decompress_plain / k_decompress_p2x32, which expand the same waypoints once per upload.
template <class T> struct PlainRc;
template <> struct PlainRc<float> {
    using Rc = fs_orbit_f32_rc;
    static __device__ __forceinline__ void load(const Rc &w, float &x, float &y) { x = w.x, y = w.y; }
    static __device__ __forceinline__ float low(const uint8_t *p) { return *reinterpret_cast<const float *>(p); }
};
template <> struct PlainRc<double> {
    using Rc = fs_orbit_f64_rc;
    static __device__ __forceinline__ void load(const Rc &w, double &x, double &y) { x = w.x, y = w.y; }
    static __device__ __forceinline__ double low(const uint8_t *p) { return *reinterpret_cast<const double *>(p); }
};
template <> struct PlainRc<df32> {
    using Rc = fs_orbit_p2x32_rc;
    static __device__ __forceinline__ void load(const Rc &w, df32 &x, df32 &y)
    {
        x = df32(w.x_head, w.x_tail), y = df32(w.y_head, w.y_tail);
    }
    static __device__ __forceinline__ df32 low(const uint8_t *p)
    {
        const fs_real_p2x32 *r = reinterpret_cast<const fs_real_p2x32 *>(p);
        return df32(r->head, r->tail);
    }
};

template <class T> struct SeqPlain {
    const typename PlainRc<T>::Rc *__restrict__ wp;
    uint32_t n_wp;
    T cx, cy;
    uint32_t idx, next, next_index;
    T zx, zy;
    __device__ __forceinline__ uint32_t index_of(uint32_t k) const { return (uint32_t)(wp[k].index_and_rebase & 0x7FFFFFFFFFFFFFFFull); }
    __device__ __forceinline__ void step()
    {
        idx++;
        if (idx == next_index) {
            PlainRc<T>::load(wp[next], zx, zy);
            next++;
            next_index = next < n_wp ? index_of(next) : 0xFFFFFFFFu;
        } else {
            const T zx_old = zx;
            zx = zx * zx - zy * zy + cx;
            zy = T(2.0f) * zx_old * zy + cy;
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
        PlainRc<T>::load(wp[lo], zx, zy);
        idx = index_of(lo);
        next = lo + 1u;
        next_index = next < n_wp ? index_of(next) : 0xFFFFFFFFu;
        while (idx < i)
            step();
    }
};

typedef float f2 __attribute__((ext_vector_type(2)));

// The perturbation loop of LAKernel.cuh:133-235 for T = float, on packed binary32 pairs.  Same IEEE operations as the
// generic loop below, in the same association:
//   s  = 2z + d                         one v_pk_fma_f32 (exact doubling, see twice_plus)
//   pa = dX * (sX, sY), pb = dY * (sY, sX)
//   n  = (pa.x - pb.x, pa.y + pb.y) + d0   one v_pk_add_f32 with a lane-wise negate, one v_pk_add_f32
//   t  = z' + n, |t|^2, |n|^2
// Per-lane state is {d, z, byte offset of the orbit entry, iter}; the select block that rebases is skipped with one
// wave-uniform branch on the steps where no lane of the wave rebases.
template <bool kStats, class IterT>
__device__ __forceinline__ void perturb_f32(const fs_orbit_f32 *__restrict__ orb, uint32_t orbit_count, f2 d, f2 d0,
                                            uint32_t &RefIteration, IterT &iter, IterT n_iterations, uint64_t &c_pt)
{
    const char *base = reinterpret_cast<const char *>(orb);
    const uint32_t max_off = (orbit_count - 1) * 8u;
    uint32_t off = RefIteration * 8u;
    const f2 z0 = *reinterpret_cast<const f2 *>(base);
    f2 z = *reinterpret_cast<const f2 *>(base + off);
    const f2 two = {2.0f, 2.0f};
    // one step: state (D, Z) -> (ND, NZ); two copies with the names swapped make the loop body, so that no step ends in
    // register copies
#define FS_PLAIN_STEP(D, Z, ND, NZ)                                                                                     \
    {                                                                                                                   \
        const f2 s_ = __builtin_elementwise_fma(Z, two, D);                                                             \
        off += 8u;                                                                                                      \
        NZ = *reinterpret_cast<const f2 *>(base + off); /* the entry this step ends on */                               \
        const f2 pa_ = D.xx * s_;                                                                                       \
        const f2 pb_ = D.yy * s_.yx;                                                                                    \
        f2 r_;                                                                                                          \
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r_) : "v"(pa_), "v"(pb_));                       \
        ND = r_ + d0;                                                                                                   \
        if (kStats)                                                                                                     \
            c_pt++;                                                                                                     \
        const f2 t_ = NZ + ND;                                                                                          \
        const f2 tt_ = t_ * t_;                                                                                         \
        const float normSquared_ = tt_.x + tt_.y;                                                                       \
        if (!(normSquared_ < 256.0f && iter < n_iterations))                                                            \
            break;                                                                                                      \
        const f2 nn_ = ND * ND;                                                                                         \
        const float DeltaNormSquared_ = nn_.x + nn_.y;                                                                  \
        const bool rebase_ = normSquared_ < DeltaNormSquared_ || off >= max_off;                                        \
        if (__builtin_amdgcn_ballot_w64(rebase_) != 0) {                                                                \
            /* the empty volatile asm keeps this a real (wave-uniform) branch instead of unconditional selects */       \
            asm volatile("" : "+v"(ND), "+v"(NZ), "+v"(off));                                                           \
            ND = rebase_ ? t_ : ND;                                                                                     \
            NZ = rebase_ ? z0 : NZ;                                                                                     \
            off = rebase_ ? 0u : off;                                                                                   \
        }                                                                                                               \
        ++iter;                                                                                                         \
    }
    f2 d2, z2;
    for (;;) {
        FS_PLAIN_STEP(d, z, d2, z2)
        FS_PLAIN_STEP(d2, z2, d, z)
    }
#undef FS_PLAIN_STEP
    RefIteration = off >> 3;
}

// IterT: the reference's IterType for the counters (LAKernel.cuh:3): uint32_t, or uint64_t for caps of 2^32 and above.
// kSeq: the orbit stays compressed (A.wp): every entry comes from a SeqPlain cursor (the generic loop for all three types).
template <class T, int Mode, bool kStats, class IterT = uint32_t, bool kSeq = false>
__global__ void __launch_bounds__(256) k_lav2_plain(FsLav2ArgsPlain A)
{
    using P = Plain<T>;
    uint32_t X, L;
    tile_pixel(X, L);
    uint64_t c_at = 0, c_la = 0, c_pt = 0, c_px = 0;
    const uint32_t Y = global_row(A.frame, L);
    const bool live = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    if (live) {
        c_px = 1;
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        const typename P::Real *co = reinterpret_cast<const typename P::Real *>(A.coords);
        const typename P::AT &at = *reinterpret_cast<const typename P::AT *>(A.at);
        const typename P::LA *__restrict__ las = reinterpret_cast<const typename P::LA *>(A.las);
        const typename P::Orbit *__restrict__ orb = reinterpret_cast<const typename P::Orbit *>(A.orbit);
        const T Two = P::from_int(2);
        // LAKernel.cuh:39-63
        const T DeltaSub0X = ld(co[0]) * P::from_int((int)X) - ld(co[2]);
        const T DeltaSub0Y = -ld(co[1]) * P::from_int((int)Y) - ld(co[3]);
        const cx<T> DeltaSub0{DeltaSub0X, DeltaSub0Y};
        cx<T> DeltaSubN{P::from_int(0), P::from_int(0)};
        IterT iter = 0;
        uint32_t RefIteration = 0;

        if (Mode != FS_MODE_PO) {
            // :66-71 + ATInfo::isValid / PerformAT (plain arms), ATInfo.h:126-188.  For T = CudaDblflt `<=` is the
            // reference's operator as written (CudaDblflt.h:218-222, df32_math.hpp).
            if (A.la_valid && A.use_at && cheb(DeltaSub0) <= ld(at.ThresholdC)) {
                const IterT ATMaxIt = n_iterations / at.StepLength;
                const cx<T> c = DeltaSub0 * ld(at.CCoeff) + ld(at.RefC);
                cx<T> z{P::from_int(0), P::from_int(0)};
                const T esc = ld(at.SqrEscapeRadius);
                IterT i;
                for (i = 0; i < ATMaxIt; i++) {
                    if (norm2(z) > esc)
                        break;
                    z = z * z + c;
                }
                DeltaSubN = z * ld(at.InvZCoeff);
                iter = i * at.StepLength;
                if (kStats)
                    c_at = i;
            }
            // :73-131 (complex0's value before the stage loop is dead)
            uint32_t CurrentLAStage = A.la_valid ? A.stage_count : 0;
            const T dcCheb = cheb(DeltaSub0);
            while (CurrentLAStage > 0) {
                CurrentLAStage--;
                const uint32_t LAIndex = A.stages[CurrentLAStage].LAIndex;
                if (dcCheb >= ld(las[LAIndex].LAThresholdC)) // GPU_LAReference.h:238-254
                    continue;
                const uint32_t MacroItCount = A.stages[CurrentLAStage].MacroItCount;
                uint32_t j = RefIteration;
                while (iter < n_iterations) {
                    const typename P::LA *LAj = &las[LAIndex + j]; // getLA, GPU_LAReference.h:271-303
                    const uint32_t l = LAj->StepLength;
                    bool unusable = true;
                    cx<T> newDz{P::from_int(0), P::from_int(0)};
                    if (iter + l <= n_iterations) {
                        // Prepare, GPU_LAInfoDeep.h:90-106
                        newDz = DeltaSubN * (mul_real(ld(LAj->Ref), Two) + DeltaSubN);
                        unusable = cheb(newDz) >= ld(LAj->LAThreshold);
                    }
                    if (unusable) {
                        RefIteration = LAj->NextStageLAIndex;
                        break;
                    }
                    iter += l;
                    if (kStats)
                        c_la++;
                    // Evaluate GPU_LAInfoDeep.h:120-124, getZ LAstep.h:181-185
                    DeltaSubN = newDz * ld(LAj->ZCoeff) + DeltaSub0 * ld(LAj->CCoeff);
                    const cx<T> complex0 = ld(LAj[1].Ref) + DeltaSubN;
                    j++;
                    if (cheb(complex0) < cheb(DeltaSubN) || j >= MacroItCount) {
                        DeltaSubN = complex0;
                        j = 0;
                    }
                }
                if (iter >= n_iterations)
                    break;
            }
        }

        if (Mode != FS_MODE_LAO) {
            // :133-235.  perturbLoop(maxRefIteration) at :254-276 reads the block's previous results, which are zero on
            // the cleared buffer every caller passes (Fractal.cpp:2822), so only perturbLoop(n_iterations) runs.
            if constexpr (std::is_same<T, float>::value && !kSeq) {
                perturb_f32<kStats, IterT>(orb, A.orbit_count, f2{DeltaSubN.re, DeltaSubN.im}, f2{DeltaSub0X, DeltaSub0Y},
                                    RefIteration, iter, n_iterations, c_pt);
            } else {
            const uint32_t MaxRef = A.orbit_count - 1;
            const T TwoFiftySix = P::from_int(256);
            T dX = DeltaSubN.re, dY = DeltaSubN.im;
            SeqPlain<T> seq;
            cx<T> z0, z;
            if constexpr (kSeq) {
                seq.wp = reinterpret_cast<const typename PlainRc<T>::Rc *>(A.wp);
                seq.n_wp = A.n_wp;
                seq.cx = PlainRc<T>::low(A.c_low[0]), seq.cy = PlainRc<T>::low(A.c_low[1]);
                seq.seek(0);
                z0 = cx<T>{seq.zx, seq.zy};
                if (RefIteration != 0)
                    seq.seek(RefIteration);
                z = cx<T>{seq.zx, seq.zy};
            } else {
                z0 = ldz(orb, 0);
                z = ldz(orb, RefIteration);
            }
            for (;;) {
                const T sumY = twice_plus(z.im, dY, Two); // tempSum1 = zy * T{2} + dY
                const T sumX = twice_plus(z.re, dX, Two); // tempSum2
                ++RefIteration;
                if constexpr (kSeq) {
                    seq.step();
                    z = cx<T>{seq.zx, seq.zy};
                } else {
                    z = ldz(orb, RefIteration); // GetIterSeq: the entry this step ends on, requested before the arithmetic
                }
                const T nX = dX * sumX - dY * sumY + DeltaSub0X;
                const T nY = dX * sumY + dY * sumX + DeltaSub0Y;
                if (kStats)
                    c_pt++;
                const T tX = z.re + nX;
                const T tY = z.im + nY;
                const T normSquared = tX * tX + tY * tY;
                if (!(normSquared < TwoFiftySix && iter < n_iterations))
                    break;
                const T DeltaNormSquared = nX * nX + nY * nY;
                const bool rebase = normSquared < DeltaNormSquared || RefIteration >= MaxRef;
                dX = nX;
                dY = nY;
                if (__builtin_amdgcn_ballot_w64(rebase) != 0) {
                    // (the empty volatile asm keeps this a real wave-uniform branch, skipped on the steps where no lane
                    // of the wave rebases, instead of unconditional selects)
                    asm volatile("" : "+v"(RefIteration));
                    dX = rebase ? tX : dX;
                    dY = rebase ? tY : dY;
                    RefIteration = rebase ? 0u : RefIteration;
                    z.re = rebase ? z0.re : z.re;
                    z.im = rebase ? z0.im : z.im;
                    if constexpr (kSeq) {
                        if (rebase)
                            seq.seek(0); // a new SeqWorkspace at the start of the orbit
                    }
                }
                ++iter;
            }
            }
        }
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, c_at, c_la, c_pt, c_px);
}

template <class T> void launch(const FsLav2ArgsPlain &A, int mode, bool stats, hipStream_t s)
{
    const dim3 b(256);
    const dim3 g((A.frame.width + 31) / 32, (A.frame.local_rows + 7) / 8); // tile_pixel()
#define FS_LAUNCH(M)                                                                                                    \
    do {                                                                                                                \
        if (A.wp != nullptr && A.frame.wide != 0u)                                                                      \
            hipLaunchKernelGGL((k_lav2_plain<T, M, false, uint64_t, true>), g, b, 0, s, A);                             \
        else if (A.wp != nullptr)                                                                                       \
            hipLaunchKernelGGL((k_lav2_plain<T, M, false, uint32_t, true>), g, b, 0, s, A);                             \
        else if (A.frame.wide != 0u)                                                                                    \
            hipLaunchKernelGGL((k_lav2_plain<T, M, false, uint64_t>), g, b, 0, s, A);                                   \
        else if (stats)                                                                                                 \
            hipLaunchKernelGGL((k_lav2_plain<T, M, true>), g, b, 0, s, A);                                              \
        else                                                                                                            \
            hipLaunchKernelGGL((k_lav2_plain<T, M, false>), g, b, 0, s, A);                                             \
    } while (0)
    if (mode == FS_MODE_PO)
        FS_LAUNCH(FS_MODE_PO);
    else if (mode == FS_MODE_LAO)
        FS_LAUNCH(FS_MODE_LAO);
    else
        FS_LAUNCH(FS_MODE_FULL);
#undef FS_LAUNCH
}

} // namespace

// kind: 0 = float, 1 = double, 2 = CudaDblflt
void fsk_lav2_plain(const FsLav2ArgsPlain &A, int kind, int mode, bool stats, hipStream_t s)
{
    if (kind == 0)
        launch<float>(A, mode, stats, s);
    else if (kind == 1)
        launch<double>(A, mode, stats, s);
    else
        launch<df32>(A, mode, stats, s);
}
