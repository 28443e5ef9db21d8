// hdr_math.hpp -- extended-exponent real / complex arithmetic for the MI355X per-pixel renderer.
//
// One header, compiled three ways: g++ (host input builders), hipcc host pass, hipcc gfx950 device
// pass.  Everything here must be built with -ffp-contract=off: the parity target is the reference's
// CPU RenderAlgorithm functions, which are compiled for baseline x86-64 (no FMA), so every a*b+c below
// is two correctly rounded IEEE operations.  gfx950 keeps f32/f64 denormals by default, like x86.
//
// Semantics follow the reference's HpSharkFloatLib/HDRFloat.h and HDRFloatComplex.h (file:line cited
// per function); the representation and API are this project's own:
//   hreal<F>  = { F m; int32 e }   value = m * 2^e          (reference: HDRFloat<F>,        8/16 B)
//   hcplx<F>  = { F re, im; int32 e }                        (reference: HDRFloatComplex<F>, 12/24 B)
// F is float or double.
#pragma once

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FS_HD __host__ __device__ __forceinline__
#else
#define FS_HD inline
#endif

namespace fs {

// INT32_MIN >> 3, HDRFloat.h:50-58
static constexpr int32_t kMinBigExp = -268435456;
// HDRFloat.h:122-123
static constexpr int32_t kExpDiffIgnored = 120;

template <class F> struct fbits;
template <> struct fbits<float> {
    using U = uint32_t;
    static constexpr U kExpMask = 0x7F800000u;
    static constexpr U kKeepMask = 0x807FFFFFu;
    static constexpr U kOneExp = 0x3F800000u;
    static constexpr int kShift = 23;
    static constexpr int kBias = 127;
    static constexpr int kMaxMulExp = 128; // getMultiplier saturates at >= 128
};
template <> struct fbits<double> {
    using U = uint64_t;
    static constexpr U kExpMask = 0x7FF0000000000000ull;
    static constexpr U kKeepMask = 0x800FFFFFFFFFFFFFull;
    static constexpr U kOneExp = 0x3FF0000000000000ull;
    static constexpr int kShift = 52;
    static constexpr int kBias = 1023;
    static constexpr int kMaxMulExp = 1024;
};

template <class F> FS_HD typename fbits<F>::U to_bits(F v)
{
    typename fbits<F>::U u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __builtin_bit_cast(typename fbits<F>::U, v);
#else
    memcpy(&u, &v, sizeof(u));
#endif
    return u;
}

template <class F> FS_HD F from_bits(typename fbits<F>::U u)
{
    F v;
#if defined(__HIP_DEVICE_COMPILE__)
    v = __builtin_bit_cast(F, u);
#else
    memcpy(&v, &u, sizeof(v));
#endif
    return v;
}

// Raw biased exponent field of v.
template <class F> FS_HD int32_t exp_field(F v)
{
    return (int32_t)((to_bits<F>(v) & fbits<F>::kExpMask) >> fbits<F>::kShift);
}

// 2^s as F for s inside the normal range; built from bits so that host and device agree exactly with
// scalbnf(1.0f, s).
template <class F> FS_HD F pow2_normal(int32_t s)
{
    using U = typename fbits<F>::U;
    return from_bits<F>((U)(s + fbits<F>::kBias) << fbits<F>::kShift);
}

template <class F> FS_HD F type_max();
template <> FS_HD float type_max<float>() { return 3.402823466e+38f; }
template <> FS_HD double type_max<double>() { return 1.7976931348623157e+308; }

// HDRFloat.h:497-521 getMultiplier: <= -bias -> 0, >= bias+1 -> max, else 2^s.
template <class F> FS_HD F multiplier(int32_t s)
{
    if (s <= -fbits<F>::kBias)
        return F(0);
    if (s >= fbits<F>::kMaxMulExp)
        return type_max<F>();
    return pow2_normal<F>(s);
}

// HDRFloat.h:523-551 getMultiplierNeg: <= -bias -> 0, else scalbn(1, s).  Only called with s <= 0
// on the paths restated here.
template <class F> FS_HD F multiplier_neg(int32_t s)
{
    if (s <= -fbits<F>::kBias)
        return F(0);
    return pow2_normal<F>(s);
}

template <class F> struct hreal {
    F m;
    int32_t e;
};

template <class F> struct hcplx {
    F re;
    F im;
    int32_t e;
};

using hreal32 = hreal<float>;
using hcplx32 = hcplx<float>;
using hreal64 = hreal<double>;
using hcplx64 = hcplx<double>;

FS_HD int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
FS_HD int32_t clamp_exp(int32_t e) { return e < kMinBigExp ? kMinBigExp : e; }

// ---------------------------------------------------------------- hreal

// HDRFloat.h:200-204 default constructor.
template <class F> FS_HD hreal<F> hr_zero() { return hreal<F>{F(0), kMinBigExp}; }

// HDRFloat.h:268-272 raw (exp, mantissa) constructor.
template <class F> FS_HD hreal<F> hr_raw(int32_t e, F m) { return hreal<F>{m, e}; }

// HDRFloat.h:414-457 Reduce(): zero mantissa leaves the exponent untouched; denormal mantissas are
// mis-normalised by construction (biased field 0) and that is kept.
template <class F> FS_HD void hr_reduce(hreal<F> &a)
{
    if (a.m == F(0))
        return;
    const auto bits = to_bits<F>(a.m);
    const int32_t fe = (int32_t)((bits & fbits<F>::kExpMask) >> fbits<F>::kShift) - fbits<F>::kBias;
    a.m = from_bits<F>((bits & fbits<F>::kKeepMask) | fbits<F>::kOneExp);
    a.e += fe;
}
template <class F> FS_HD hreal<F> hr_reduced(hreal<F> a)
{
    hr_reduce(a);
    return a;
}

// HDRFloat.h:206-212 `explicit HDRFloat(T mant)`: {mant, 0} then Reduce, so 0.0f -> {0, exp 0}.
template <class F> FS_HD hreal<F> hr_from_mant(F v)
{
    hreal<F> r{v, 0};
    hr_reduce(r);
    return r;
}

// HDRFloat.h:295-363 templated `HDRFloat(const U number)` (int / float / double argument): zero maps to
// {0, kMinBigExp}, everything else is normalised.
template <class F> FS_HD hreal<F> hr_from_number(F v)
{
    if (v == F(0))
        return hr_zero<F>();
    const auto bits = to_bits<F>(v);
    const int32_t fe = (int32_t)((bits & fbits<F>::kExpMask) >> fbits<F>::kShift) - fbits<F>::kBias;
    return hreal<F>{from_bits<F>((bits & fbits<F>::kKeepMask) | fbits<F>::kOneExp), fe};
}

// HDRFloat.h:829-840 multiply_mutable.
template <class F> FS_HD hreal<F> hr_mul(hreal<F> a, hreal<F> b)
{
    return hreal<F>{a.m * b.m, clamp_exp(a.e + b.e)};
}

// `HDRFloat * 2` as written in Fractal.cpp:2346-2356 / BLAS.cpp:78-79: the int becomes T(2.0), goes
// through HDRFloat(T mant) = {1.0, 1}, then multiply_mutable.
template <class F> FS_HD hreal<F> hr_mul2(hreal<F> a) { return hreal<F>{a.m * F(1), clamp_exp(a.e + 1)}; }

// HDRFloat.h:624-636 divide_mutable.
template <class F> FS_HD hreal<F> hr_div(hreal<F> a, hreal<F> b)
{
    return hreal<F>{a.m / b.m, clamp_exp(a.e - b.e)};
}

// HDRFloat.h:877-884 square() (no clamp).
template <class F> FS_HD hreal<F> hr_square(hreal<F> a) { return hreal<F>{a.m * a.m, a.e * 2}; }

// HDRFloat.h:974-1000 add_mutable.
template <class F> FS_HD hreal<F> hr_add(hreal<F> a, hreal<F> b)
{
    const int32_t d = a.e - b.e;
    if (d >= kExpDiffIgnored) {
        return a; // NB: returns before the zero-exponent reset, as the reference does
    } else if (d >= 0) {
        const F mul = multiplier_neg<F>(-d);
        a.m = a.m + b.m * mul;
    } else if (d > -kExpDiffIgnored) {
        const F mul = multiplier_neg<F>(d);
        a.e = b.e;
        a.m = a.m * mul + b.m;
    } else {
        a.e = b.e;
        a.m = b.m;
    }
    if (a.m == F(0))
        a.e = kMinBigExp;
    return a;
}

// HDRFloat.h:1039-1065 subtract_mutable.
template <class F> FS_HD hreal<F> hr_sub(hreal<F> a, hreal<F> b)
{
    const int32_t d = a.e - b.e;
    if (d >= kExpDiffIgnored) {
        return a;
    } else if (d >= 0) {
        const F mul = multiplier_neg<F>(-d);
        a.m = a.m - b.m * mul;
    } else if (d > -kExpDiffIgnored) {
        const F mul = multiplier_neg<F>(d);
        a.e = b.e;
        a.m = a.m * mul - b.m;
    } else {
        a.e = b.e;
        a.m = -b.m;
    }
    if (a.m == F(0))
        a.e = kMinBigExp;
    return a;
}

// HDRFloat.h:606-613 reciprocal(): raw, not reduced.
template <class F> FS_HD hreal<F> hr_recip(hreal<F> a) { return hreal<F>{F(1) / a.m, -a.e}; }

template <class F> FS_HD hreal<F> hr_neg(hreal<F> a) { return hreal<F>{-a.m, a.e}; }
// HDRFloat.h:1385-1404 HdrAbs.
template <class F> FS_HD F fabs_bits(F v)
{
    using U = typename fbits<F>::U;
    return from_bits<F>(to_bits<F>(v) & ~((U)1 << (sizeof(U) * 8 - 1)));
}
template <class F> FS_HD hreal<F> hr_abs(hreal<F> a) { return hreal<F>{fabs_bits<F>(a.m), a.e}; }

// HDRFloat.h:1150-1167 compareToBothPositiveReduced (also compareToBothPositive :1186-1204).
template <class F> FS_HD int hr_cmp_pos(hreal<F> a, hreal<F> b)
{
    if (a.e > b.e)
        return 1;
    if (a.e < b.e)
        return -1;
    if (a.m > b.m)
        return 1;
    if (a.m < b.m)
        return -1;
    return 0;
}

// HDRFloat.h:1207-1248 compareTo (general sign).
template <class F> FS_HD int hr_cmp(hreal<F> a, hreal<F> b)
{
    if (a.m == F(0) && b.m == F(0))
        return 0;
    if (a.m > F(0)) {
        if (b.m <= F(0))
            return 1;
        if (a.e > b.e)
            return 1;
        if (a.e < b.e)
            return -1;
        return a.m > b.m ? 1 : (a.m < b.m ? -1 : 0);
    } else {
        if (b.m > F(0))
            return -1;
        if (a.e > b.e)
            return -1;
        if (a.e < b.e)
            return 1;
        return a.m > b.m ? 1 : (a.m < b.m ? -1 : 0);
    }
}

template <class F> FS_HD hreal<F> hr_max_pos(hreal<F> a, hreal<F> b) { return hr_cmp_pos(a, b) > 0 ? a : b; }
template <class F> FS_HD hreal<F> hr_min_pos(hreal<F> a, hreal<F> b) { return hr_cmp_pos(a, b) < 0 ? a : b; }

// HDRFloat.h:1358-1383 HdrSqrt (host builders only).
template <class F> inline hreal<F> hr_sqrt(hreal<F> a)
{
    const bool odd = (a.e & 1) != 0;
    return hreal<F>{(F)::sqrt(odd ? F(2) * a.m : a.m), odd ? (a.e - 1) / 2 : a.e / 2};
}

// toDouble(), HDRFloat.h:557-561.
template <class F> FS_HD F hr_to_native(hreal<F> a) { return a.m * multiplier<F>(a.e); }

// ---------------------------------------------------------------- hcplx

// HDRFloatComplex.h:133-138 default.
template <class F> FS_HD hcplx<F> hc_zero() { return hcplx<F>{F(0), F(0), kMinBigExp}; }

// HDRFloatComplex.h:158,166-173 from two HDRFloat (setMantexp).
template <class F> FS_HD hcplx<F> hc_from_hr(hreal<F> re, hreal<F> im)
{
    hcplx<F> r;
    r.e = imax(re.e, im.e);
    r.re = re.m * multiplier<F>(re.e - r.e);
    r.im = im.m * multiplier<F>(im.e - r.e);
    return r;
}

// HDRFloatComplex.h:160-163 from two scalars, via HDRFloat(T mant): zeros carry exponent 0.
template <class F> FS_HD hcplx<F> hc_from_native(F re, F im)
{
    return hc_from_hr(hr_from_mant<F>(re), hr_from_mant<F>(im));
}

template <class F> FS_HD hreal<F> hc_re(hcplx<F> a) { return hreal<F>{a.re, a.e}; }
template <class F> FS_HD hreal<F> hc_im(hcplx<F> a) { return hreal<F>{a.im, a.e}; }

// HDRFloatComplex.h:219-247 plus_mutable(complex).
template <class F> FS_HD hcplx<F> hc_add(hcplx<F> a, hcplx<F> b)
{
    const int32_t d = a.e - b.e;
    if (d >= kExpDiffIgnored) {
        return a;
    } else if (d >= 0) {
        const F mul = multiplier<F>(-d);
        a.re = a.re + b.re * mul;
        a.im = a.im + b.im * mul;
    } else if (d > -kExpDiffIgnored) {
        const F mul = multiplier<F>(d);
        a.e = b.e;
        a.re = a.re * mul + b.re;
        a.im = a.im * mul + b.im;
    } else {
        a = b;
    }
    return a;
}

// HDRFloatComplex.h:361-388 plus_mutable(HDRFloat real).
template <class F> FS_HD hcplx<F> hc_add_real(hcplx<F> a, hreal<F> r)
{
    const int32_t d = a.e - r.e;
    if (d >= kExpDiffIgnored) {
        return a;
    } else if (d >= 0) {
        const F mul = multiplier<F>(-d);
        a.re = a.re + r.m * mul;
    } else if (d > -kExpDiffIgnored) {
        const F mul = multiplier<F>(d);
        a.e = r.e;
        a.re = a.re * mul + r.m;
        a.im = a.im * mul;
    } else {
        a.e = r.e;
        a.re = r.m;
        a.im = F(0);
    }
    return a;
}

// HDRFloatComplex.h:267-283 times_mutable(complex).
template <class F> FS_HD hcplx<F> hc_mul(hcplx<F> a, hcplx<F> b)
{
    const F re = (a.re * b.re) - (a.im * b.im);
    const F im = (a.re * b.im) + (a.im * b.re);
    return hcplx<F>{re, im, clamp_exp(a.e + b.e)};
}

// HDRFloatComplex.h:334-348 times_mutable(HDRFloat).
template <class F> FS_HD hcplx<F> hc_mul_real(hcplx<F> a, hreal<F> s)
{
    return hcplx<F>{a.re * s.m, a.im * s.m, clamp_exp(a.e + s.e)};
}

// z * HDRFloat(2): HDRFloat(int 2) = {1.0, 1}.
template <class F> FS_HD hcplx<F> hc_mul2(hcplx<F> a) { return hcplx<F>{a.re * F(1), a.im * F(1), clamp_exp(a.e + 1)}; }

// HDRFloatComplex.h:473-510 Reduce(): raw biased exponent fields, a zero component contributes 0.
template <class F> FS_HD void hc_reduce(hcplx<F> &a)
{
    if (a.re == F(0) && a.im == F(0))
        return;
    const int32_t d = imax(exp_field<F>(a.re), exp_field<F>(a.im)) - fbits<F>::kBias;
    const F mul = multiplier<F>(-d);
    a.re *= mul;
    a.im *= mul;
    a.e += d;
}
template <class F> FS_HD hcplx<F> hc_reduced(hcplx<F> a)
{
    hc_reduce(a);
    return a;
}

// HDRFloatComplex.h:544-548 norm_squared.
template <class F> FS_HD hreal<F> hc_norm2(hcplx<F> a) { return hreal<F>{a.re * a.re + a.im * a.im, a.e << 1}; }

// HDRFloatComplex.h:691-695 chebychevNorm: both parts share the exponent, so the larger |mantissa| wins
// (ties and the `> 0 ? a : b` orientation pick the imaginary part, which has the same value).
template <class F> FS_HD hreal<F> hc_cheb(hcplx<F> a)
{
    const F ar = fabs_bits<F>(a.re);
    const F ai = fabs_bits<F>(a.im);
    return hreal<F>{ar > ai ? ar : ai, a.e};
}

// HDRFloatComplex.h:551-555 norm(): the mantissas' Euclidean length under the shared exponent.
template <class F> FS_HD hreal<F> hc_norm(hcplx<F> a)
{
    const F n2 = a.re * a.re + a.im * a.im;
    return hreal<F>{sizeof(F) == 4 ? (F)__builtin_sqrtf((float)n2) : (F)__builtin_sqrt((double)n2), a.e};
}

// HDRFloatComplex.h:574-594 divide_mutable(complex).
template <class F> FS_HD hcplx<F> hc_div(hcplx<F> a, hcplx<F> b)
{
    const F t = F(1) / (b.re * b.re + b.im * b.im);
    const F re = (a.re * b.re + a.im * b.im) * t;
    const F im = (a.im * b.re - a.re * b.im) * t;
    return hcplx<F>{re, im, clamp_exp(a.e - b.e)};
}

// HDRFloatComplex.h:556-561 reciprocal.
template <class F> FS_HD hcplx<F> hc_recip(hcplx<F> a)
{
    const F t = F(1) / (a.re * a.re + a.im * a.im);
    return hcplx<F>{a.re * t, -a.im * t, -a.e};
}

} // namespace fs
