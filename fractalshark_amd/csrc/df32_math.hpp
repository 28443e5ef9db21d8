// df32_math.hpp -- the 2x32 numeric type: a float-float mantissa (head + tail, ~48 bits) that plugs into the
// extended-exponent templates of hdr_math.hpp as F = df32, giving
//   hreal<df32>  = reference HDRFloat<CudaDblflt<MattDblflt>>         (12 B: head, tail, exp)
//   hcplx<df32>  = reference HDRFloatComplex<CudaDblflt<MattDblflt>>  (20 B)
//
// The operation sequences are the published double-float algorithms the reference uses (Knuth two-sum; Thall's
// df64 add; the FMA-based product), HpSharkFloatLib/dblflt.cuh:86-215 and dblflt.h:19-29, and the comparison rules
// of CudaDblflt.h:197-255.  Each __fadd_rn / __fmul_rn / __fmaf_rn of the reference is one IEEE binary32 operation:
// this header is compiled with -ffp-contract=off, so a*b+c below is never fused and fused operations are written as
// fs::fma32().  gfx950 keeps binary32 denormals (v_fma_f32 / v_add_f32 / v_mul_f32 in the default mode), like the
// reference's non-fast-math build.
//
// There is no CPU twin of this type in the reference (CudaDblflt arithmetic only exists under __CUDACC__); the
// host-side converters (double -> head/tail) live in fractalshark_amd/host/refinputs.cpp.
#pragma once

#include "hdr_math.hpp"

namespace fs {

FS_HD float fma32(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return ::fmaf(a, b, c);
#endif
}

struct df32 {
    float head; // most significant part
    float tail; // |tail| <= ulp(head)/2 for normalised values

    df32() = default;
    // CudaDblflt(float) -> MattDblflt(float) -> two-sum(f, 0) = {f, 0}; CudaDblflt.h:64-65, dblflt.h:54-55
    FS_HD explicit constexpr df32(float f) : head(f), tail(0.0f) {}
    FS_HD explicit constexpr df32(int v) : head((float)v), tail(0.0f) {}
    FS_HD constexpr df32(float h, float t) : head(h), tail(t) {}
};

// add_dblflt, dblflt.cuh:116-132
FS_HD df32 operator+(df32 a, df32 b)
{
    float t1 = a.head + b.head;
    float t2 = t1 - a.head;
    float t3 = (a.head + (t2 - t1)) + (b.head - t2);
    float t4 = a.tail + b.tail;
    t2 = t4 - a.tail;
    const float t5 = (a.tail + (t2 - t4)) + (b.tail - t2);
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    const float e = t4 + t3;
    return df32(e, (t4 - e) + t3);
}

// sub_dblflt, dblflt.cuh:141-157.  `x + -(y)` and `x - y` are the same IEEE operation.
FS_HD df32 operator-(df32 a, df32 b)
{
    float t1 = a.head - b.head;
    float t2 = t1 - a.head;
    float t3 = (a.head + (t2 - t1)) - (b.head + t2);
    float t4 = a.tail - b.tail;
    t2 = t4 - a.tail;
    const float t5 = (a.tail + (t2 - t4)) - (b.tail + t2);
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    const float e = t4 + t3;
    return df32(e, (t4 - e) + t3);
}

// mul_dblflt, dblflt.cuh:164-176
FS_HD df32 operator*(df32 a, df32 b)
{
    const float th = a.head * b.head;
    float tt = fma32(a.head, b.head, -th);
    tt = fma32(a.tail, b.tail, tt);
    tt = fma32(a.head, b.tail, tt);
    tt = fma32(a.tail, b.head, tt);
    const float e = th + tt;
    return df32(e, (th - e) + tt);
}

FS_HD df32 operator-(df32 a) { return df32(-a.head, -a.tail); } // CudaDblflt.h:181-186
FS_HD df32 &operator*=(df32 &a, df32 b)
{
    a = a * b;
    return a;
}

// CudaDblflt.h:197-245
FS_HD bool operator<(df32 a, df32 b) { return a.head < b.head || (a.head == b.head && a.tail < b.tail); }
FS_HD bool operator==(df32 a, df32 b) { return a.head == b.head && a.tail == b.tail; }
FS_HD bool operator>(df32 a, df32 b) { return !(a < b) && !(b == a); }
FS_HD bool operator>=(df32 a, df32 b) { return !(a < b); }
FS_HD bool operator<=(df32 a, df32 b) { return !(b > a); }

// CudaDblflt::abs, CudaDblflt.h:247-255 (HdrAbs for the 2x32 HDRFloat, HDRFloat.h:1400-1402)
template <> FS_HD df32 fabs_bits<df32>(df32 v) { return v < df32(0.0f) ? -v : v; }

// getMultiplier / getMultiplierNeg, HDRFloat.h:497-551 (float / CudaDblflt branch): (T)scalbnf(1, s).
// std::numeric_limits<CudaDblflt>::max() is the unspecialised primary template, i.e. a zero; the >= 128 arm is
// unreachable on this path (scale factors are <= 0).
template <> FS_HD df32 multiplier<df32>(int32_t s)
{
    if (s <= -127 || s >= 128)
        return df32(0.0f);
    return df32(pow2_normal<float>(s));
}
template <> FS_HD df32 multiplier_neg<df32>(int32_t s)
{
    if (s <= -127)
        return df32(0.0f);
    return df32(pow2_normal<float>(s));
}

// Reduce(), HDRFloat.h:458-488: the head is renormalised to [1,2); the tail keeps its mantissa bits and gets the
// head's exponent shift applied to its *exponent field*, saturating at field 0 (so a zero tail under a head < 1
// turns into a tiny power of two -- kept, it is what the reference computes).
template <> FS_HD void hr_reduce<df32>(hreal<df32> &a)
{
    if (a.m.head == 0.0f && a.m.tail == 0.0f)
        return;
    const uint32_t by = to_bits<float>(a.m.head);
    const uint32_t bx = to_bits<float>(a.m.tail);
    const int32_t fy = (int32_t)((by & 0x7F800000u) >> 23) - 127;
    const int32_t fx = (int32_t)((bx & 0x7F800000u) >> 23);
    const int32_t ne = fx - fy;
    const int32_t se = ne <= 0 ? 0 : ne;
    a.m.head = from_bits<float>((by & 0x807FFFFFu) | 0x3F800000u);
    a.m.tail = from_bits<float>((bx & 0x807FFFFFu) | ((uint32_t)se << 23));
    a.e += fy;
}

// HDRFloat(const U number) for U = int / float, HDRFloat.h:293-363: zero -> {0, kMinBigExp}; otherwise the float's
// mantissa in the head, zero tail.
FS_HD hreal<df32> hr2_from_float(float v)
{
    if (v == 0.0f)
        return hreal<df32>{df32(0.0f), kMinBigExp};
    const uint32_t bits = to_bits<float>(v);
    const int32_t fe = (int32_t)((bits & 0x7F800000u) >> 23) - 127;
    return hreal<df32>{df32(from_bits<float>((bits & 0x807FFFFFu) | 0x3F800000u)), fe};
}

// HDRFloatComplex::Reduce(), CudaDblflt branch, HDRFloatComplex.h:472-500: each part through HDRFloat(T mant)
// (mantissa, exponent 0, Reduce), Reduce again, setMantexp, then the old exponent is added back.  The second Reduce is
// not issued: after the first one the head's exponent field is 127 (or head and tail are both zero and Reduce returns at
// once), so it would shift by 0 binades -- head, tail and exponent come back bit for bit.
template <> FS_HD void hc_reduce<df32>(hcplx<df32> &a)
{
    if (a.re == df32(0.0f) && a.im == df32(0.0f))
        return;
    hreal<df32> tr{a.re, 0}, ti{a.im, 0};
    hr_reduce(tr);
    hr_reduce(ti);
    const int32_t old = a.e;
    a = hc_from_hr(tr, ti);
    a.e += old;
}


#if defined(__HIPCC__)
// Two double-floats side by side (device code only): the SAME operation sequences as df32's operators above, applied to both
// halves of packed binary32 registers (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 are the IEEE operations of their scalar
// forms on each half), so .lo() / .hi() of a result are bit for bit what the df32 operator gives for the corresponding
// operands.  Used where the reference's arithmetic has two independent double-float operations of the same kind next to
// each other -- the real and imaginary part of a complex step.
struct df32x2 {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 head, tail;
    __device__ __forceinline__ df32x2() = default;
    __device__ __forceinline__ df32x2(f2 h, f2 t) : head(h), tail(t) {}
    __device__ __forceinline__ df32x2(df32 a, df32 b) : head{a.head, b.head}, tail{a.tail, b.tail} {}
    __device__ __forceinline__ df32 lo() const { return df32(head.x, tail.x); }
    __device__ __forceinline__ df32 hi() const { return df32(head.y, tail.y); }
    __device__ __forceinline__ df32x2 swapped() const { return df32x2(head.yx, tail.yx); }
    // (-lo, hi): negation of a double-float is exact, and `x - y` is the IEEE operation `x + (-y)` (operator- above)
    __device__ __forceinline__ df32x2 neg_lo() const { return df32x2((f2){-head.x, head.y}, (f2){-tail.x, tail.y}); }
};

// add_dblflt on both halves (operator+ above, line for line)
__device__ __forceinline__ df32x2 operator+(df32x2 a, df32x2 b)
{
    typedef df32x2::f2 f2;
    f2 t1 = a.head + b.head;
    f2 t2 = t1 - a.head;
    f2 t3 = (a.head + (t2 - t1)) + (b.head - t2);
    f2 t4 = a.tail + b.tail;
    t2 = t4 - a.tail;
    const f2 t5 = (a.tail + (t2 - t4)) + (b.tail - t2);
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    const f2 e = t4 + t3;
    return df32x2(e, (t4 - e) + t3);
}

// mul_dblflt on both halves (operator* above, line for line)
__device__ __forceinline__ df32x2 operator*(df32x2 a, df32x2 b)
{
    typedef df32x2::f2 f2;
    const f2 th = a.head * b.head;
    f2 tt = __builtin_elementwise_fma(a.head, b.head, -th);
    tt = __builtin_elementwise_fma(a.tail, b.tail, tt);
    tt = __builtin_elementwise_fma(a.head, b.tail, tt);
    tt = __builtin_elementwise_fma(a.tail, b.head, tt);
    const f2 e = th + tt;
    return df32x2(e, (th - e) + tt);
}

// a * {m, 0} on both halves for a multiplier that is a plain float (a power of two wherever this is used): mul_dblflt with
// the two fused operations on the multiplier's zero tail left out -- they return their addend: x * 0 + t = t for a finite x
// unless t is -0, and t, the exact error of the head product (or what the previous fused operation made of it), is never -0.
__device__ __forceinline__ df32x2 mul_by_float(df32x2 a, df32x2::f2 m)
{
    typedef df32x2::f2 f2;
    const f2 th = a.head * m;
    f2 tt = __builtin_elementwise_fma(a.head, m, -th);
    tt = __builtin_elementwise_fma(a.tail, m, tt);
    const f2 e = th + tt;
    return df32x2(e, (th - e) + tt);
}

// (a.lo - b.lo, a.hi + b.hi): sub_dblflt in the lower half and add_dblflt in the upper one, operation for operation (a
// subtraction IS the addition of the negated operand, so each packed addition below with one negated half is the scalar
// operation of the corresponding line of operator- / operator+ above).
__device__ __forceinline__ df32x2 sub_lo_add_hi(df32x2 a, df32x2 b)
{
    typedef df32x2::f2 f2;
    f2 t1 = a.head + (f2){-b.head.x, b.head.y};
    f2 t2 = t1 - a.head;
    const f2 v3 = b.head + (f2){t2.x, -t2.y};
    f2 t3 = (a.head + (t2 - t1)) + (f2){-v3.x, v3.y};
    f2 t4 = a.tail + (f2){-b.tail.x, b.tail.y};
    t2 = t4 - a.tail;
    const f2 v5 = b.tail + (f2){t2.x, -t2.y};
    const f2 t5 = (a.tail + (t2 - t4)) + (f2){-v5.x, v5.y};
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    const f2 e = t4 + t3;
    return df32x2(e, (t4 - e) + t3);
}

// Two HDRFloat<CudaDblflt> side by side, each with its own exponent: the X and the Y part of the scalar-HDR perturbation
// step (HDRFloat::custom_perturb3 and the norms around it), whose operations come in pairs of the same kind.
struct hreal2 {
    df32x2 m;
    int32_t ex, ey;
    __device__ __forceinline__ hreal2() = default;
    __device__ __forceinline__ hreal2(df32x2 m_, int32_t ex_, int32_t ey_) : m(m_), ex(ex_), ey(ey_) {}
    __device__ __forceinline__ hreal2(hreal<df32> x, hreal<df32> y) : m(x.m, y.m), ex(x.e), ey(y.e) {}
    __device__ __forceinline__ hreal<df32> x() const { return hreal<df32>{m.lo(), ex}; }
    __device__ __forceinline__ hreal<df32> y() const { return hreal<df32>{m.hi(), ey}; }
};

// (a.x - b.x, a.y + b.y) for kSubLo, else (a.x + b.x, a.y + b.y): add_mutable / subtract_mutable (hr_add / hr_sub,
// hdr_math.hpp) of both pairs in one straight line.  The reference's four-way branch on the exponent gap d = a.e - b.e is,
// for |d| < 120, "scale the operand with the smaller exponent by 2^-|d|, then add, first operand first"; which operand that
// is is a per-lane select here and the scaling one packed double-float product for both parts (mul_by_float: the
// multiplier is {2^-|d|, 0}).  From a gap of 120 on the reference returns the
// operand with the larger exponent untouched (deep zooms live there: dz is hundreds of binades below the orbit): one more
// select per word.  What the straight line does NOT cover is reported in `rare` and left to the caller's literal path: a
// zero result (the reference resets the exponent) and non-finite values.
template <bool kSubLo> __device__ __forceinline__ hreal2 hr_add2(hreal2 a, hreal2 b, bool &rare)
{
    typedef df32x2::f2 f2;
    const int32_t dx = a.ex - b.ex, dy = a.ey - b.ey;
    // every lane of the wave 120 binades or more above its second operand in both parts (a deep zoom's 2z + dz and z' + dz'):
    // the reference returns the first operand as it is, and so does this -- one vote instead of the product and the sum
    if (__builtin_amdgcn_ballot_w64(!(dx >= kExpDiffIgnored && dy >= kExpDiffIgnored)) == 0ull)
        return a;
    const bool nx = dx < 0, ny = dy < 0;
    const int32_t gx = nx ? -dx : dx, gy = ny ? -dy : dy;
    const f2 mul = {__builtin_amdgcn_ldexpf(1.0f, -gx), __builtin_amdgcn_ldexpf(1.0f, -gy)}; // multiplier_neg(-|d|), |d| < 120
    const f2 sh = {nx ? a.m.head.x : b.m.head.x, ny ? a.m.head.y : b.m.head.y};
    const f2 st = {nx ? a.m.tail.x : b.m.tail.x, ny ? a.m.tail.y : b.m.tail.y};
    const df32x2 S = mul_by_float(df32x2(sh, st), mul);
    const f2 se = S.head, sl = S.tail;
    const df32x2 A((f2){nx ? se.x : a.m.head.x, ny ? se.y : a.m.head.y}, (f2){nx ? sl.x : a.m.tail.x, ny ? sl.y : a.m.tail.y});
    const df32x2 B((f2){nx ? b.m.head.x : se.x, ny ? b.m.head.y : se.y}, (f2){nx ? b.m.tail.x : sl.x, ny ? b.m.tail.y : sl.y});
    const df32x2 W = kSubLo ? sub_lo_add_hi(A, B) : A + B;
    // a gap of 120 or more: the operand with the larger exponent, untouched (negated when it is the subtrahend)
    const bool fx = gx >= kExpDiffIgnored, fy = gy >= kExpDiffIgnored;
    const float uhx = nx ? (kSubLo ? -B.head.x : B.head.x) : A.head.x, utx = nx ? (kSubLo ? -B.tail.x : B.tail.x) : A.tail.x;
    const float uhy = ny ? B.head.y : A.head.y, uty = ny ? B.tail.y : A.tail.y;
    const df32x2 R((f2){fx ? uhx : W.head.x, fy ? uhy : W.head.y}, (f2){fx ? utx : W.tail.x, fy ? uty : W.tail.y});
    // neither zero nor infinite nor a NaN (class mask: +-normal, +-denormal)
    rare = rare || !__builtin_amdgcn_classf(R.head.x, 0x198) || !__builtin_amdgcn_classf(R.head.y, 0x198);
    return hreal2(R, nx ? b.ex : a.ex, ny ? b.ey : a.ey);
}
#endif

using hreal2x32 = hreal<df32>;
using hcplx2x32 = hcplx<df32>;
static_assert(sizeof(hreal2x32) == 12 && sizeof(hcplx2x32) == 20, "2x32 records");

} // namespace fs
