// kernels_tables.hip -- BLA table construction on the device.
//
// Replaces the host-side BLAS<IterType,T>::Init (FractalSharkLib/BLAS.cpp:212-255), which the reference re-runs on the
// CPU for every BLA render (Fractal.cpp:2739-2740) and then copies to the GPU.  Here the table is built in HBM straight
// from the uploaded orbit and never crosses PCIe.  Results are bit-identical to the host builder
// (fractalshark_amd/host/refinputs.cpp BlaBuilder, itself pinned through the golden CRCs of the Cpu*PerturbedBLAHDR
// algorithms): same operations in the same order, -ffp-contract=off, correctly rounded sqrt and divide.
//
//   first materialised level (m_FirstLevel = 2, BLAS.h:22): element m = CreateLStep(2, m) (BLAS.cpp:49-72) = the merge
//       tree over up to four single-step records CreateOneStep (BLAS.cpp:74-93) -- one lane per element;
//   every further level: element k = MergeTwoBlas(src[2k], src[2k+1]) or a copy of src[2k] at a ragged end
//       (BLAS.cpp:141-210) -- one lane per element, one launch per level (log2(orbit) launches, each at least halving).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"
#include "bla_math.hpp"
#include "kernels.h"

using namespace fs;

namespace {

template <class F> using Rec = BlaRec<F>;

__device__ __forceinline__ hcplx<float> zat(const float4 *__restrict__ z, uint32_t i)
{
    const float4 v = z[i];
    return hcplx<float>{v.x, v.y, __float_as_int(v.z)};
}
__device__ __forceinline__ hcplx<double> zat(const FsZ64 *__restrict__ z, uint32_t i)
{
    return hcplx<double>{z[i].re, z[i].im, z[i].e};
}

// The record arithmetic itself is csrc/bla_math.hpp (shared with the host builder and the known-answer harness):
// BLAS::CreateOneStep (BLAS.cpp:74-93) at orbit entry m, BLAS::MergeTwoBlas (BLAS.cpp:25-47).
template <class F, class Z> __device__ __forceinline__ Rec<F> one_step(const Z *__restrict__ zref, uint32_t m, hreal<F> epsilon)
{
    return bla_one_step<F>(zat(zref, m), epsilon);
}
template <class F> __device__ __forceinline__ Rec<F> merge(const Rec<F> &x, const Rec<F> &y, hreal<F> blaSize)
{
    return bla_merge<F>(x, y, blaSize);
}

template <class F> __device__ __forceinline__ Rec<F> ld_rec(const typename FsDev<F>::BLA &b)
{
    return Rec<F>{hreal<F>{b.r2.m, b.r2.e}, hreal<F>{b.Ax.m, b.Ax.e}, hreal<F>{b.Ay.m, b.Ay.e}, hreal<F>{b.Bx.m, b.Bx.e},
                  hreal<F>{b.By.m, b.By.e}, b.l};
}
template <class F> __device__ __forceinline__ void st_rec(typename FsDev<F>::BLA *dst, const Rec<F> &s)
{
    typename FsDev<F>::BLA o;
    memset(&o, 0, sizeof(o)); // the double record has padding: keep it deterministic
    o.r2.m = s.r2.m, o.r2.e = s.r2.e;
    o.Ax.m = s.Ax.m, o.Ax.e = s.Ax.e;
    o.Ay.m = s.Ay.m, o.Ay.e = s.Ay.e;
    o.Bx.m = s.Bx.m, o.Bx.e = s.Bx.e;
    o.By.m = s.By.m, o.By.e = s.By.e;
    o.l = s.l;
    *dst = o;
}

// level-1 element k (1-based), BLAS::CreateLStep(1, k)
template <class F, class Z>
__device__ __forceinline__ Rec<F> level1(const Z *__restrict__ zref, uint64_t k, uint64_t epl0, hreal<F> blaSize, hreal<F> eps)
{
    const uint64_t m2 = k << 1;
    const Rec<F> x = one_step<F>(zref, (uint32_t)(m2 - 1), eps);
    if (m2 <= epl0)
        return merge(x, one_step<F>(zref, (uint32_t)m2, eps), blaSize);
    return x;
}

// first materialised level: dst[m-1] = CreateLStep(2, m), m = 1..n
template <class F, class Z>
__global__ void __launch_bounds__(256) k_bla_first_level(const Z *__restrict__ zref, typename FsDev<F>::BLA *__restrict__ dst,
                                                         uint64_t n, uint64_t epl0, uint64_t epl1, hreal<F> blaSize,
                                                         hreal<F> eps)
{
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (m > n)
        return;
    const uint64_t m2 = m << 1;
    Rec<F> x = level1<F>(zref, m2 - 1, epl0, blaSize, eps);
    if (m2 <= epl1)
        x = merge(x, level1<F>(zref, m2, epl0, blaSize, eps), blaSize);
    st_rec<F>(&dst[m - 1], x);
}

// BLAS::Merge, BLAS.cpp:141-210: dst[k] = merge(src[2k], src[2k+1]) or src[2k] at the ragged end
template <class F>
__global__ void __launch_bounds__(256) k_bla_merge(const typename FsDev<F>::BLA *__restrict__ src, uint64_t n_src,
                                                   typename FsDev<F>::BLA *__restrict__ dst, uint64_t n_dst, hreal<F> blaSize)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_dst)
        return;
    const uint64_t mx = k << 1, my = mx + 1;
    if (my < n_src)
        st_rec<F>(&dst[k], merge(ld_rec<F>(src[mx]), ld_rec<F>(src[my]), blaSize));
    else
        dst[k] = src[mx];
}

} // namespace

template <class F, class Z>
static void build_levels(const Z *zref, void *const *levels, const uint64_t *epl, int n_levels, hreal<F> blaSize,
                         hipStream_t s)
{
    using B = typename FsDev<F>::BLA;
    // T(1) / T{1L << 23}: templated ctor for the int, HDRFloat(T mant) for the scalar (BLAS.cpp:216)
    const hreal<F> eps = hr_div(hr_from_number<F>(F(1)), hr_from_mant<F>(F(8388608)));
    const int first = 2;
    if (n_levels <= first)
        return;
    const uint64_t n = epl[first];
    hipLaunchKernelGGL((k_bla_first_level<F, Z>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, zref, (B *)levels[first],
                       n, epl[0], epl[1], blaSize, eps);
    for (int src = first; src + 1 < n_levels && epl[src] > 1; src++) {
        const uint64_t nd = epl[src + 1];
        hipLaunchKernelGGL((k_bla_merge<F>), dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, (const B *)levels[src],
                           epl[src], (B *)levels[src + 1], nd, blaSize);
    }
}

// ---- device-native form of an HDRFloat<float> table (see FsBlaRec in kernels.h)
namespace {

struct NativeGeom {
    uint32_t level_off[kBlaMaxLevels];
    uint32_t level_n[kBlaMaxLevels];
    int32_t n_levels;
    uint32_t total;
};

__device__ __forceinline__ long long r2_key(fs_real_hdr32 r2, uint32_t *bad)
{
    const int bits = __float_as_int(r2.m);
    // the integer compare equals the reference's float compare of the mantissas only for non-negative, non-NaN values
    if (bits < 0 || (bits & 0x7F800000) == 0x7F800000) {
        atomicOr(bad, 1u);
        return (long long)0x8000000000000000ull;
    }
    return (long long)(((unsigned long long)(unsigned)r2.e << 32) | (unsigned)bits);
}

__global__ void __launch_bounds__(256) k_bla_make_native(const fs_bla_hdr32 *const *__restrict__ levels, NativeGeom G,
                                                         const float4 *__restrict__ zref, uint32_t orbit_count,
                                                         FsBlaRec *__restrict__ rec, int4 *__restrict__ lad,
                                                         uint32_t *__restrict__ bad)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= G.total)
        return;
    int32_t L = 2;
    for (int32_t l = 3; l < G.n_levels; l++)
        if (G.level_n[l] != 0u && p >= G.level_off[l])
            L = l;
    const uint32_t ix = p - G.level_off[L];
    const fs_bla_hdr32 b = levels[L][ix];
    FsBlaRec o;
    o.Axm = b.Ax.m, o.Aym = b.Ay.m, o.Bxm = b.Bx.m, o.Bym = b.By.m;
    o.Axe = b.Ax.e, o.Aye = b.Ay.e, o.Bxe = b.Bx.e, o.Bye = b.By.e;
    o.l = (uint32_t)b.l;
    // the only orbit index this element is looked up at is (ix << L) + 1 (BLAS.cpp:283-300: ix = k >> zeros with
    // zeros >= L, and the probes below the start level shift it back), so the entry the jump arrives at is fixed
    const uint64_t arrive = ((uint64_t)ix << L) + 1u + (uint64_t)o.l;
    const float4 z = arrive < orbit_count ? zref[arrive] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    o.Zre = z.x, o.Zim = z.y, o.Ze = __float_as_int(z.z);
    rec[p] = o;
    long long k[4];
    for (int j = 0; j < 4; j++) {
        const int32_t Lj = L - j;
        // (Lj, ix << j) is the first element of the left sub-tree j levels down: it exists whenever (L, ix) does
        k[j] = Lj >= 2 ? r2_key(levels[Lj][(size_t)ix << j].r2, bad) : (long long)0x8000000000000000ull;
    }
    lad[2 * (size_t)p] = make_int4((int)(unsigned long long)k[0], (int)((unsigned long long)k[0] >> 32),
                                   (int)(unsigned long long)k[1], (int)((unsigned long long)k[1] >> 32));
    lad[2 * (size_t)p + 1] = make_int4((int)(unsigned long long)k[2], (int)((unsigned long long)k[2] >> 32),
                                       (int)(unsigned long long)k[3], (int)((unsigned long long)k[3] >> 32));
}

// Pre-test of the lookup: kmax[q] = the largest key among the elements BLAS::LookupBackwards probes at orbit index
// m = 4 q + 1 (k = 4 q: start level min(zeros(k), lm2), element index k >> zeros(k) -- from zeros, not from the capped level,
// as BLAS.cpp:283-300 has it -- then one level down and the index doubled, until level 2).  |dz|^2 >= kmax[q] means that
// no probe of that walk can hold: the lookup is over after one 8-byte load and one compare, for every lane of a wave whose
// lookup finds nothing -- which is how every outer trip of the kernel ends.  An element index outside its level (the walk
// would read beyond the table) disables the pre-test for that q (INT64_MAX: never rejects).
__global__ void __launch_bounds__(256) k_bla_make_kmax(const int4 *__restrict__ lad, NativeGeom G, int32_t lm2,
                                                       long long *__restrict__ kmax, uint32_t n)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n)
        return;
    const uint32_t k = q << 2;
    const int32_t zeros = k == 0u ? 32 : (int32_t)__ffs((int)k) - 1;
    uint32_t ix = k == 0u ? 0u : k >> zeros;
    long long best = (long long)0x8000000000000000ull;
    for (int32_t L = zeros <= lm2 ? zeros : lm2; L >= 2; L--, ix <<= 1) {
        if (L >= G.n_levels || ix >= G.level_n[L]) {
            best = 0x7FFFFFFFFFFFFFFFll;
            break;
        }
        const int4 a = lad[2u * ((size_t)G.level_off[L] + ix)];
        const long long key = (long long)(((unsigned long long)(unsigned)a.y << 32) | (unsigned)a.x);
        best = key > best ? key : best;
    }
    kmax[q] = best;
}

} // namespace

void fsk_bla_make_native(const fs_bla_hdr32 *const *levels, const uint32_t *level_off, const uint64_t *epl, int n_levels,
                         const float4 *zref, uint32_t orbit_count, FsBlaRec *rec, int4 *lad, uint32_t *bad, int32_t lm2,
                         long long *kmax, uint32_t n_kmax, hipStream_t s)
{
    NativeGeom G;
    memset(&G, 0, sizeof(G));
    G.n_levels = n_levels;
    uint32_t total = 0;
    for (int l = 2; l < n_levels && l < kBlaMaxLevels; l++) {
        G.level_off[l] = level_off[l];
        G.level_n[l] = (uint32_t)epl[l];
        total = level_off[l] + (uint32_t)epl[l];
    }
    G.total = total;
    if (total == 0)
        return;
    hipLaunchKernelGGL(k_bla_make_native, dim3((total + 255u) / 256u), dim3(256), 0, s, levels, G, zref, orbit_count, rec, lad,
                       bad);
    if (n_kmax != 0u)
        hipLaunchKernelGGL(k_bla_make_kmax, dim3((n_kmax + 255u) / 256u), dim3(256), 0, s, lad, G, lm2, kmax, n_kmax);
}

void fsk_bla_build_hdr32(const float4 *zref, void *const *levels, const uint64_t *epl, int n_levels, fs_real_hdr32 bla_size,
                         hipStream_t s)
{
    build_levels<float>(zref, levels, epl, n_levels, hreal<float>{bla_size.m, bla_size.e}, s);
}
void fsk_bla_build_hdr64(const FsZ64 *zref, void *const *levels, const uint64_t *epl, int n_levels, fs_real_hdr64 bla_size,
                         hipStream_t s)
{
    build_levels<double>(zref, levels, epl, n_levels, hreal<double>{bla_size.m, bla_size.e}, s);
}
