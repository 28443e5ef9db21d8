// kernels_decompress.hip -- PerturbExtras::SimpleCompression orbits for T = float, double, CudaDblflt and
// HDRFloat<CudaDblflt>: expansion of the waypoint list into the full orbit, once per upload.
//
// The reference's *RC* kernels rebuild each entry while they iterate (GPUPerturbSingleResults::GetCompressedComplex /
// GetCompressedComplexSeq, FractalSharkGpuLib/Perturb.cuh:272-326): the waypoint at or below the wanted index, advanced
// with  zx' = zx*zx - zy*zy + OrbitXLow;  zy' = Type{2}*zx_old*zy + OrbitYLow  in T arithmetic, HdrReduce after each
// (the identity for a non-HDR T).  The value at an index is therefore a pure function of the waypoints, and the
// expansion below -- one lane per waypoint segment -- hands the uncompressed kernels exactly the entries the
// reference's kernels would have rebuilt.  (HDRFloat<float> / HDRFloat<double>: k_decompress_orbit_hdr32/64 in kernels.hip.)
// Compiled with -ffp-contract=off: one IEEE operation per source operation, in source order; for float / double that is
// also what the host's RuntimeDecompressor computes (PerturbationResultsHelpers.h:51-58).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fs_layout.h"
#include "df32_math.hpp"
#include "../../include/fsmi355.h"
#include "kernels.h"

using namespace fs;

namespace {

constexpr uint64_t kIndexMask = 0x7FFFFFFFFFFFFFFFull;

template <class Rc>
__device__ __forceinline__ bool segment(const Rc *wp, uint64_t n_wp, uint64_t n_uncompressed, uint64_t &k, uint64_t &i0, uint64_t &i1)
{
    k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_wp)
        return false;
    i0 = wp[k].index_and_rebase & kIndexMask;
    i1 = k + 1 < n_wp ? (wp[k + 1].index_and_rebase & kIndexMask) : n_uncompressed;
    if (i1 > n_uncompressed)
        i1 = n_uncompressed;
    return true;
}

template <class T, class Rc, class Out>
__global__ void k_decompress_plain(const Rc *__restrict__ wp, uint64_t n_wp, uint64_t n_uncompressed, T cxLow, T cyLow,
                                   Out *__restrict__ out)
{
    uint64_t k, i0, i1;
    if (!segment(wp, n_wp, n_uncompressed, k, i0, i1))
        return;
    T zx = wp[k].x, zy = wp[k].y;
    for (uint64_t i = i0; i < i1; i++) {
        out[i] = Out{zx, zy};
        const T zx_old = zx;
        zx = zx * zx - zy * zy + cxLow;
        zy = T(2.0f) * zx_old * zy + cyLow;
    }
}

__global__ void k_decompress_p2x32(const fs_orbit_p2x32_rc *__restrict__ wp, uint64_t n_wp, uint64_t n_uncompressed,
                                   fs_real_p2x32 cxLow, fs_real_p2x32 cyLow, fs_orbit_p2x32 *__restrict__ out)
{
    uint64_t k, i0, i1;
    if (!segment(wp, n_wp, n_uncompressed, k, i0, i1))
        return;
    df32 zx(wp[k].x_head, wp[k].x_tail), zy(wp[k].y_head, wp[k].y_tail);
    const df32 cx(cxLow.head, cxLow.tail), cy(cyLow.head, cyLow.tail);
    const df32 Two(2.0f); // Type{2}: only CudaDblflt(float) is viable on the device (CudaDblflt.h:52-68)
    for (uint64_t i = i0; i < i1; i++) {
        out[i] = fs_orbit_p2x32{zx.head, zx.tail, zy.head, zy.tail};
        const df32 zx_old = zx;
        zx = zx * zx - zy * zy + cx;
        zy = Two * zx_old * zy + cy;
    }
}

__global__ void k_decompress_hdr2x32(const fs_orbit_2x32_rc *__restrict__ wp, uint64_t n_wp, uint64_t n_uncompressed,
                                     fs_real_2x32 cxLow, fs_real_2x32 cyLow, fs_orbit_2x32 *__restrict__ out)
{
    uint64_t k, i0, i1;
    if (!segment(wp, n_wp, n_uncompressed, k, i0, i1))
        return;
    hreal<df32> zx{df32(wp[k].x_head, wp[k].x_tail), wp[k].ex}, zy{df32(wp[k].y_head, wp[k].y_tail), wp[k].ey};
    const hreal<df32> cx{df32(cxLow.head, cxLow.tail), cxLow.e}, cy{df32(cyLow.head, cyLow.tail), cyLow.e};
    const hreal<df32> Two = hr2_from_float(2.0f);
    for (uint64_t i = i0; i < i1; i++) {
        out[i] = fs_orbit_2x32{zx.m.head, zx.m.tail, zx.e, zy.e, zy.m.head, zy.m.tail};
        const hreal<df32> zx_old = zx;
        zx = hr_add(hr_sub(hr_mul(zx, zx), hr_mul(zy, zy)), cx);
        hr_reduce(zx);
        zy = hr_add(hr_mul(hr_mul(Two, zx_old), zy), cy);
        hr_reduce(zy);
    }
}

} // namespace

void fsk_decompress_orbit_plain(int type_tag, const void *wp, uint64_t n_wp, uint64_t n_uncompressed, const void *cxLow,
                                const void *cyLow, void *out, hipStream_t s)
{
    const dim3 g((unsigned)((n_wp + 63) / 64)), b(64);
    if (type_tag == FS_T_F32)
        hipLaunchKernelGGL((k_decompress_plain<float, fs_orbit_f32_rc, fs_orbit_f32>), g, b, 0, s, (const fs_orbit_f32_rc *)wp,
                           n_wp, n_uncompressed, *(const float *)cxLow, *(const float *)cyLow, (fs_orbit_f32 *)out);
    else if (type_tag == FS_T_F64)
        hipLaunchKernelGGL((k_decompress_plain<double, fs_orbit_f64_rc, fs_orbit_f64>), g, b, 0, s, (const fs_orbit_f64_rc *)wp,
                           n_wp, n_uncompressed, *(const double *)cxLow, *(const double *)cyLow, (fs_orbit_f64 *)out);
    else if (type_tag == FS_T_2X32)
        hipLaunchKernelGGL(k_decompress_p2x32, g, b, 0, s, (const fs_orbit_p2x32_rc *)wp, n_wp, n_uncompressed,
                           *(const fs_real_p2x32 *)cxLow, *(const fs_real_p2x32 *)cyLow, (fs_orbit_p2x32 *)out);
    else
        hipLaunchKernelGGL(k_decompress_hdr2x32, g, b, 0, s, (const fs_orbit_2x32_rc *)wp, n_wp, n_uncompressed,
                           *(const fs_real_2x32 *)cxLow, *(const fs_real_2x32 *)cyLow, (fs_orbit_2x32 *)out);
}
