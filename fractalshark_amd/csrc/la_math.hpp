// la_math.hpp -- the arithmetic of one LAv2 table record, shared by the host input builder (host/refinputs.cpp, the
// golden-pinned one) and the device builder (csrc/kernels_la.hip, fs_build_la): LAInfoDeep construction, Step, Composite,
// DetectPeriod (LAInfoDeep.h:108-391), CreateAT (:456-506), ATInfo::Usable (ATInfo.h:101-116), LAParameters defaults
// (LAParameters.h:66-75).  ONE source for both, so that a record built on the device is the record the host builder
// builds, operation for operation; what differs between the two builders is only how the segment boundaries are found.
// Number families: F = float | double -> HDRFloat<F> / HDRFloatComplex<F> (host and device); F = plain<float|double> ->
// the non-HDR arms (host only).
#pragma once

#include <algorithm>
#include <cmath>

#include "hdr_math.hpp"

namespace fs {
namespace la {

// LAParameters defaults, LAParameters.h:66-75 / LAParameters.cpp:61-71 (floats are exp2 of the exponents).
struct LAParams {
    int detectionMethod = 1;
    int laThresholdScaleExp = -24;
    int laThresholdCScaleExp = -24;
    int stage0PeriodDetectionThreshold2Exp = -6;
    int periodDetectionThreshold2Exp = -3;
    int stage0PeriodDetectionThresholdExp = -10;
    int periodDetectionThresholdExp = -10;
};

// ---- number families.  The LA builder below is written once against real_t<F> / cplx_t<F>:
//   F = float | double           T = HDRFloat<F>, complex = HDRFloatComplex<F>   (hdr_math.hpp)
//   F = plain<float|double>      T = float | double itself, complex = FloatComplex<T> (FloatComplex.h:7-420): the
//                                `else` arms of every `if constexpr (IsHDR)` in LAInfoDeep.h / ATInfo.h, HdrReduce
//                                a no-op (HDRFloat.h:1406-1419), min = std::min, compare = operator<.
template <class T> struct plain {};
template <class T> struct preal {
    T m; // named like hreal's mantissa so that `x.m == 0` reads the same for both families
};
template <class T> struct pcplx {
    T re, im;
};
template <class F> struct num {
    using R = hreal<F>;
    using C = hcplx<F>;
    using S = F;
    static constexpr bool is_plain = false;
};
template <class T> struct num<plain<T>> {
    using R = preal<T>;
    using C = pcplx<T>;
    using S = T;
    static constexpr bool is_plain = true;
};
template <class F> using real_t = typename num<F>::R;
template <class F> using cplx_t = typename num<F>::C;
template <class F> using scalar_t = typename num<F>::S;

template <class T> preal<T> hr_mul(preal<T> a, preal<T> b) { return preal<T>{a.m * b.m}; }
template <class T> preal<T> hr_div(preal<T> a, preal<T> b) { return preal<T>{a.m / b.m}; }
template <class T> preal<T> hr_square(preal<T> a) { return preal<T>{a.m * a.m}; }
template <class T> void hr_reduce(preal<T> &) {}
template <class T> preal<T> hr_reduced(preal<T> a) { return a; }
template <class T> preal<T> hr_min_pos(preal<T> a, preal<T> b) { return preal<T>{std::min(a.m, b.m)}; }
template <class T> int hr_cmp_pos(preal<T> a, preal<T> b) { return a.m < b.m ? -1 : (a.m > b.m ? 1 : 0); }
template <class T> pcplx<T> hc_from_hr(preal<T> re, preal<T> im) { return pcplx<T>{re.m, im.m}; }
template <class T> pcplx<T> hc_reduced(pcplx<T> a) { return a; }
// chebychevNorm, FloatComplex.h:413-419
template <class T> preal<T> hc_cheb(pcplx<T> a)
{
    const T ar = std::fabs(a.re), ai = std::fabs(a.im);
    return preal<T>{ar > ai ? ar : ai};
}
// times_mutable(FloatComplex), FloatComplex.h:198-211
template <class T> pcplx<T> hc_mul(pcplx<T> a, pcplx<T> b)
{
    const T re = (a.re * b.re) - (a.im * b.im);
    const T im = (a.re * b.im) + (a.im * b.re);
    return pcplx<T>{re, im};
}
template <class T> pcplx<T> hc_mul_real(pcplx<T> a, preal<T> f) { return pcplx<T>{a.re * f.m, a.im * f.m}; } // :245-251
template <class T> pcplx<T> hc_add(pcplx<T> a, pcplx<T> b) { return pcplx<T>{a.re + b.re, a.im + b.im}; }  // :188-195
template <class T> pcplx<T> hc_add_real(pcplx<T> a, preal<T> r) { return pcplx<T>{a.re + r.m, a.im}; }      // :263-268
template <class T> preal<T> hc_norm2(pcplx<T> a) { return preal<T>{a.re * a.re + a.im * a.im}; }            // :327-331
// reciprocal(), FloatComplex.h:339-344
template <class T> pcplx<T> hc_recip(pcplx<T> a)
{
    const T temp = T(1) / (a.re * a.re + a.im * a.im);
    return pcplx<T>{a.re * temp, -a.im * temp};
}

// constructors spelled per family
template <class F> struct mk {
    static FS_HD hreal<F> zero() { return hr_zero<F>(); }
    static FS_HD hcplx<F> czero() { return hc_zero<F>(); }
    static FS_HD hcplx<F> cnative(double re, double im) { return hc_from_native<F>(F(re), F(im)); }
    static FS_HD hreal<F> number(double v) { return hr_from_number<F>(F(v)); }
    static FS_HD hreal<F> mant(double v) { return hr_from_mant<F>(F(v)); }
    static FS_HD hreal<F> raw_pow2(int e) { return hr_raw<F>(e, F(1)); }
};
template <class T> struct mk<plain<T>> {
    static preal<T> zero() { return preal<T>{T(0)}; }
    static pcplx<T> czero() { return pcplx<T>{T(0), T(0)}; }
    static pcplx<T> cnative(double re, double im) { return pcplx<T>{T(re), T(im)}; }
    static preal<T> number(double v) { return preal<T>{T(v)}; }
    static preal<T> mant(double v) { return preal<T>{T(v)}; }
    static preal<T> raw_pow2(int e) { return preal<T>{T(std::ldexp(1.0, e))}; }
};

// `HDRFloat * float` for a power-of-two float goes through HDRFloat(T mant) -> {1.0, exp}; for a plain T it is the
// float constant itself (LAParameters.cpp:61-71), exact in either width.
template <class F> FS_HD real_t<F> pow2_hr(int e)
{
    if constexpr (num<F>::is_plain)
        return real_t<F>{scalar_t<F>(std::ldexp(1.0f, e))};
    else
        return real_t<F>{F(1), e};
}

template <class F> struct LAInfo {
    cplx_t<F> Ref = mk<F>::czero();
    cplx_t<F> ZCoeff = mk<F>::czero();
    cplx_t<F> CCoeff = mk<F>::czero();
    real_t<F> LAThreshold = mk<F>::zero();
    real_t<F> LAThresholdC = mk<F>::zero();
    real_t<F> MinMag = mk<F>::zero();
    uint32_t StepLength = 0;
    uint32_t NextStageLAIndex = 0;
};

// LAInfoDeep(la_parameters, z), LAInfoDeep.h:108-131
template <class F> FS_HD LAInfo<F> la_init(const LAParams &p, cplx_t<F> z)
{
    LAInfo<F> r;
    r.Ref = z;
    r.ZCoeff = mk<F>::cnative(1, 0);
    r.CCoeff = mk<F>::cnative(1, 0);
    r.LAThreshold = mk<F>::number(1);
    r.LAThresholdC = mk<F>::number(1);
    if (p.detectionMethod == 1)
        r.MinMag = mk<F>::number(4);
    return r;
}

// LAInfoDeep::Step(params, out, z), LAInfoDeep.h:178-246 -- `out` keeps whatever it held in the fields
// Step does not write (LAi; MinMag when detectionMethod != 1).
template <class F> FS_HD bool la_step(const LAParams &p, const LAInfo<F> &self, LAInfo<F> &out, cplx_t<F> z)
{
    const real_t<F> ChebyMagz = hc_cheb(z);
    const real_t<F> ChebyMagZCoeff = hc_cheb(self.ZCoeff);
    const real_t<F> ChebyMagCCoeff = hc_cheb(self.CCoeff);
    if (p.detectionMethod == 1)
        out.MinMag = hr_min_pos(ChebyMagz, self.MinMag);

    real_t<F> temp1 = hr_mul(hr_div(ChebyMagz, ChebyMagZCoeff), pow2_hr<F>(p.laThresholdScaleExp));
    hr_reduce(temp1);
    real_t<F> temp2 = hr_mul(hr_div(ChebyMagz, ChebyMagCCoeff), pow2_hr<F>(p.laThresholdCScaleExp));
    hr_reduce(temp2);
    out.LAThreshold = hr_min_pos(self.LAThreshold, temp1);
    out.LAThresholdC = hr_min_pos(self.LAThresholdC, temp2);

    const cplx_t<F> z2 = hc_mul_real(z, mk<F>::number(2));
    out.ZCoeff = hc_reduced(hc_mul(z2, self.ZCoeff));
    out.CCoeff = hc_reduced(hc_add_real(hc_mul(z2, self.CCoeff), mk<F>::number(1)));
    out.Ref = self.Ref;

    if (p.detectionMethod == 1)
        return hr_cmp_pos(out.MinMag, hr_mul(self.MinMag, pow2_hr<F>(p.stage0PeriodDetectionThreshold2Exp))) < 0;
    return hr_cmp_pos(out.LAThreshold, hr_mul(self.LAThreshold, pow2_hr<F>(p.stage0PeriodDetectionThresholdExp))) < 0;
}

// LAInfoDeep::Step(params, z) returning a fresh record, LAInfoDeep.h:268-277
template <class F> FS_HD LAInfo<F> la_step_new(const LAParams &p, const LAInfo<F> &self, cplx_t<F> z)
{
    LAInfo<F> r;
    la_step(p, self, r, z);
    return r;
}

// LAInfoDeep::DetectPeriod, LAInfoDeep.h:133-155
template <class F> FS_HD bool la_detect_period(const LAParams &p, const LAInfo<F> &self, cplx_t<F> z)
{
    if (p.detectionMethod == 1)
        return hr_cmp_pos(hc_cheb(z), hr_mul(self.MinMag, pow2_hr<F>(p.periodDetectionThreshold2Exp))) < 0;
    const real_t<F> lhs =
        hr_mul(hr_div(hc_cheb(z), hc_cheb(self.ZCoeff)), pow2_hr<F>(p.laThresholdScaleExp));
    return hr_cmp_pos(lhs, hr_mul(self.LAThreshold, pow2_hr<F>(p.periodDetectionThresholdExp))) < 0;
}

// LAInfoDeep::Composite(params, out, LA), LAInfoDeep.h:279-369
template <class F> FS_HD bool la_composite(const LAParams &p, const LAInfo<F> &self, LAInfo<F> &out, const LAInfo<F> &LA)
{
    const cplx_t<F> z = LA.Ref;
    const real_t<F> ChebyMagz = hc_cheb(z);
    real_t<F> ChebyMagZCoeff = hc_cheb(self.ZCoeff);
    real_t<F> ChebyMagCCoeff = hc_cheb(self.CCoeff);

    real_t<F> temp1 = hr_mul(hr_div(ChebyMagz, ChebyMagZCoeff), pow2_hr<F>(p.laThresholdScaleExp));
    hr_reduce(temp1);
    real_t<F> temp2 = hr_mul(hr_div(ChebyMagz, ChebyMagCCoeff), pow2_hr<F>(p.laThresholdCScaleExp));
    hr_reduce(temp2);
    real_t<F> outLAThreshold = hr_min_pos(self.LAThreshold, temp1);
    real_t<F> outLAThresholdC = hr_min_pos(self.LAThresholdC, temp2);

    const cplx_t<F> z2 = hc_mul_real(z, mk<F>::number(2));
    cplx_t<F> outZCoeff = hc_reduced(hc_mul(z2, self.ZCoeff));
    cplx_t<F> outCCoeff = hc_reduced(hc_mul(z2, self.CCoeff));
    ChebyMagZCoeff = hc_cheb(outZCoeff);
    ChebyMagCCoeff = hc_cheb(outCCoeff);
    real_t<F> temp = outLAThreshold;

    temp1 = hr_div(LA.LAThreshold, ChebyMagZCoeff);
    hr_reduce(temp1);
    temp2 = hr_div(LA.LAThreshold, ChebyMagCCoeff);
    hr_reduce(temp2);
    outLAThreshold = hr_min_pos(outLAThreshold, temp1);
    outLAThresholdC = hr_min_pos(outLAThresholdC, temp2);
    outZCoeff = hc_reduced(hc_mul(outZCoeff, LA.ZCoeff));
    outCCoeff = hc_reduced(hc_add(hc_mul(outCCoeff, LA.ZCoeff), LA.CCoeff));

    out.LAThreshold = outLAThreshold;
    out.LAThresholdC = outLAThresholdC;
    out.ZCoeff = outZCoeff;
    out.CCoeff = outCCoeff;
    out.Ref = self.Ref;

    if (p.detectionMethod == 1) {
        temp = hr_min_pos(ChebyMagz, self.MinMag);
        out.MinMag = hr_min_pos(temp, LA.MinMag);
        return hr_cmp_pos(temp, hr_mul(self.MinMag, pow2_hr<F>(p.periodDetectionThreshold2Exp))) < 0;
    }
    return hr_cmp_pos(temp, hr_mul(self.LAThreshold, pow2_hr<F>(p.periodDetectionThresholdExp))) < 0;
}
template <class F> FS_HD LAInfo<F> la_composite_new(const LAParams &p, const LAInfo<F> &self, const LAInfo<F> &LA)
{
    LAInfo<F> r;
    la_composite(p, self, r, LA);
    return r;
}

template <class F> struct ATInfoT {
    uint32_t StepLength = 0;
    real_t<F> ThresholdC = mk<F>::zero(), SqrEscapeRadius = mk<F>::zero();
    cplx_t<F> RefC = mk<F>::czero(), ZCoeff = mk<F>::czero(), CCoeff = mk<F>::czero(), InvZCoeff = mk<F>::czero();
    cplx_t<F> CCoeffSqrInvZCoeff = mk<F>::czero(), CCoeffInvZCoeff = mk<F>::czero();
    real_t<F> CCoeffNormSqr = mk<F>::zero(), RefCNormSqr = mk<F>::zero();
    real_t<F> factor = mk<F>::number(4294967296.0); // HDRFloat(0x1.0p32), ATInfo.h:132
};

// LAInfoDeep::CreateAT, LAInfoDeep.h:456-506 (IsHDR branch; UseSmallExponents only matters for double).
template <class F> FS_HD void la_create_at(const LAInfo<F> &self, ATInfoT<F> &R, const LAInfo<F> &Next, bool useSmallExponents)
{
    R.ZCoeff = self.ZCoeff;
    R.CCoeff = hc_reduced(hc_mul(self.ZCoeff, self.CCoeff));
    R.InvZCoeff = hc_reduced(hc_recip(self.ZCoeff));
    R.CCoeffSqrInvZCoeff = hc_reduced(hc_mul(hc_mul(R.CCoeff, R.CCoeff), R.InvZCoeff));
    R.CCoeffInvZCoeff = hc_reduced(hc_mul(R.CCoeff, R.InvZCoeff));
    R.RefC = hc_reduced(hc_mul(Next.Ref, self.ZCoeff));
    R.CCoeffNormSqr = hr_reduced(hc_norm2(R.CCoeff));
    R.RefCNormSqr = hr_reduced(hc_norm2(R.RefC));

    real_t<F> lim = mk<F>::raw_pow2(32); // plain T: lim = 4294967296.0f, LAInfoDeep.h:499
    if constexpr (!num<F>::is_plain) {
        if (sizeof(F) == 8 && !useSmallExponents)
            lim.e = 256;
    }
    hr_reduce(lim);
    R.SqrEscapeRadius = hr_reduced(hr_min_pos(hr_mul(hc_norm2(self.ZCoeff), self.LAThreshold), lim));
    R.ThresholdC = hr_min_pos(self.LAThresholdC, hr_div(lim, hc_cheb(R.CCoeff)));
}

// ATInfo::Usable, ATInfo.h:101-116
template <class F> FS_HD bool at_usable(const ATInfoT<F> &at, real_t<F> SqrRadius)
{
    const real_t<F> result = hr_reduced(hr_mul(hr_mul(at.CCoeffNormSqr, SqrRadius), at.factor));
    const real_t<F> Four = mk<F>::mant(4);
    return hr_cmp_pos(result, at.RefCNormSqr) > 0 && hr_cmp_pos(at.SqrEscapeRadius, Four) > 0;
}


} // namespace la
} // namespace fs
