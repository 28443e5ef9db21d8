// lav2_common.hpp -- the few device helpers the LAv2 translation units share (kernels.hip, kernels_hdr64.hip): record ->
// hreal / hcplx loads, the prepared-orbit access and the pixel -> delta-c mapping.  One definition (round 6: moved out of
// kernels.hip when the HDRFloat<double> kernel got a translation unit of its own).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"
#include "kernels.h"

namespace {

using fs::hcplx;
using fs::hcplx32;
using fs::hcplx64;
using fs::hreal;
using fs::hreal32;
using fs::hreal64;

__device__ __forceinline__ hreal32 ldr(const fs_real_hdr32 &r) { return hreal32{r.m, r.e}; }
__device__ __forceinline__ hcplx32 ldc(const fs_cplx_hdr32 &c) { return hcplx32{c.re, c.im, c.e}; }
__device__ __forceinline__ hreal64 ldr(const fs_real_hdr64 &r) { return hreal64{r.m, r.e}; }
__device__ __forceinline__ hcplx64 ldc(const fs_cplx_hdr64 &c) { return hcplx64{c.re, c.im, c.e}; }

// Reference-orbit entry in device form: PerturbationResults::GetComplex (PerturbationResults.h:174-185)
// builds HDRFloatComplex{x, y} on *every* access; it is a pure function of the entry, so it is evaluated
// once per upload (k_prepare_orbit_*).  One access is one 16-byte (float) / 32-byte (double) record.
__device__ __forceinline__ hcplx32 zref_at(const float4 *__restrict__ z, uint32_t i)
{
    const float4 v = z[i];
    return hcplx32{v.x, v.y, __float_as_int(v.z)};
}
__device__ __forceinline__ hcplx64 zref_at(const FsZ64 *__restrict__ z, uint32_t i)
{
    return hcplx64{z[i].re, z[i].im, z[i].e};
}

// Pixel -> delta c, Fractal.cpp:2553-2562 (== 2272-2281): `dx * (float)x` goes through HDRFloat(T mant).
template <class F>
__device__ __forceinline__ void pixel_delta(const FsCoordsT<F> &c, uint32_t x, uint32_t y, hreal<F> &dRe, hreal<F> &dIm)
{
    using namespace fs;
    hreal<F> a = hr_mul(c.dx, hr_from_mant<F>((F)x)); // `dx * (SubType)x`
    hr_reduce(a);
    a = hr_sub(a, c.centerX);
    hreal<F> b = hr_mul(hr_neg(c.dy), hr_from_mant<F>((F)y));
    hr_reduce(b);
    b = hr_sub(b, c.centerY);
    hr_reduce(a);
    hr_reduce(b);
    dRe = a;
    dIm = b;
}

} // namespace
