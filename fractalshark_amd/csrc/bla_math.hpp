// bla_math.hpp -- bilinear-approximation (BLA) record arithmetic, shared by the device table builder
// (kernels_tables.hip), the host builder (host/refinputs.cpp) and the known-answer harness (tests/kat): one
// definition, compiled by g++ and by hipcc for gfx950.  Follows the reference's HpSharkFloatLib/BLA.cuh (record
// operations) and FractalSharkLib/BLAS.cpp (how records are made and merged), file:line per function; -ffp-contract=off
// like everything that must agree with the CPU functions.  F = float | double: the records hold HDRFloat<F> values.
#pragma once

#include "hdr_math.hpp"

namespace fs {

// BLA<T> (BLA.cuh:7-19): validity radius squared, A = (Ax, Ay), B = (Bx, By), steps skipped
template <class F> struct BlaRec {
    hreal<F> r2, Ax, Ay, Bx, By;
    int32_t l;
};

// HdrSqrt, HDRFloat.h:1358-1383 (correctly rounded square root on either compiler)
template <class F> FS_HD hreal<F> bla_sqrt(hreal<F> a)
{
    const bool odd = (a.e & 1) != 0;
    const F v = odd ? F(2) * a.m : a.m;
    F s;
    if (sizeof(F) == 4)
        s = (F)__builtin_sqrtf((float)v);
    else
        s = (F)__builtin_sqrt((double)v);
    return hreal<F>{s, odd ? (a.e - 1) / 2 : a.e / 2};
}

// BLA<T>::hypotA / hypotB, BLA.cuh:40-56
template <class F> FS_HD hreal<F> bla_hypot(hreal<F> a, hreal<F> b)
{
    return hr_reduced(bla_sqrt(hr_add(hr_mul(a, a), hr_mul(b, b))));
}

// BLA<T>::getValue, BLA.cuh:21-38: dz' = A dz + B dc, each part a chain of scalar HDRFloat operations in this order
template <class F>
FS_HD void bla_get_value(const BlaRec<F> &b, hreal<F> &dzx, hreal<F> &dzy, hreal<F> dcx, hreal<F> dcy)
{
    const hreal<F> nx = hr_sub(hr_add(hr_sub(hr_mul(b.Ax, dzx), hr_mul(b.Ay, dzy)), hr_mul(b.Bx, dcx)), hr_mul(b.By, dcy));
    const hreal<F> ny = hr_add(hr_add(hr_add(hr_mul(b.Ax, dzy), hr_mul(b.Ay, dzx)), hr_mul(b.Bx, dcy)), hr_mul(b.By, dcx));
    dzx = nx;
    dzy = ny;
}

// BLA<T>::getNewA, BLA.cuh:65-76: A of (y after x) = y.A x.A
template <class F> FS_HD void bla_new_a(const BlaRec<F> &x, const BlaRec<F> &y, hreal<F> &ax, hreal<F> &ay)
{
    ax = hr_reduced(hr_sub(hr_mul(y.Ax, x.Ax), hr_mul(y.Ay, x.Ay)));
    ay = hr_reduced(hr_add(hr_mul(y.Ax, x.Ay), hr_mul(y.Ay, x.Ax)));
}

// BLA<T>::getNewB, BLA.cuh:78-91: B of (y after x) = y.A x.B + y.B
template <class F> FS_HD void bla_new_b(const BlaRec<F> &x, const BlaRec<F> &y, hreal<F> &bx, hreal<F> &by)
{
    bx = hr_reduced(hr_add(hr_sub(hr_mul(y.Ax, x.Bx), hr_mul(y.Ay, x.By)), y.Bx));
    by = hr_reduced(hr_add(hr_add(hr_mul(y.Ax, x.By), hr_mul(y.Ay, x.Bx)), y.By));
}

// BLAS::CreateOneStep, BLAS.cpp:74-93: the record of one perturbation step at orbit value z
template <class F> FS_HD BlaRec<F> bla_one_step(hcplx<F> z, hreal<F> epsilon)
{
    const hreal<F> RealA = hr_mul2(hc_re(z));
    const hreal<F> ImagA = hr_mul2(hc_im(z));
    const hreal<F> mA = bla_sqrt(hr_add(hr_mul(RealA, RealA), hr_mul(ImagA, ImagA)));
    const hreal<F> r = hr_mul(mA, epsilon);
    return BlaRec<F>{hr_mul(r, r), RealA, ImagA, hr_from_number<F>(F(1)), hr_from_number<F>(F(0)), 1};
}

// BLAS::MergeTwoBlas, BLAS.cpp:25-47 (getNewA / getNewB + the merged validity radius)
template <class F> FS_HD BlaRec<F> bla_merge(const BlaRec<F> &x, const BlaRec<F> &y, hreal<F> blaSize)
{
    BlaRec<F> o;
    o.l = x.l + y.l;
    bla_new_a(x, y, o.Ax, o.Ay);
    bla_new_b(x, y, o.Bx, o.By);
    const hreal<F> xA = bla_hypot(x.Ax, x.Ay);
    const hreal<F> xB = bla_hypot(x.Bx, x.By);
    const hreal<F> tempR = hr_reduced(hr_div(hr_sub(bla_sqrt(y.r2), hr_mul(xB, blaSize)), xA));
    const hreal<F> zero = hr_from_number<F>(F(0));
    const hreal<F> mx = hr_cmp(zero, tempR) > 0 ? zero : tempR; // HdrMaxReduced(T(0), tempR)
    const hreal<F> sx = bla_sqrt(x.r2);
    const hreal<F> r = hr_cmp_pos(sx, mx) < 0 ? sx : mx; // HdrMinPositiveReduced
    o.r2 = hr_mul(r, r);
    return o;
}

} // namespace fs
