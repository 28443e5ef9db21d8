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
// (mantissa, exponent 0, Reduce), Reduce again (no-op), setMantexp, then the old exponent is added back.
template <> FS_HD void hc_reduce<df32>(hcplx<df32> &a)
{
    if (a.re == df32(0.0f) && a.im == df32(0.0f))
        return;
    hreal<df32> tr{a.re, 0}, ti{a.im, 0};
    hr_reduce(tr);
    hr_reduce(ti);
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
#endif

using hreal2x32 = hreal<df32>;
using hcplx2x32 = hcplx<df32>;
static_assert(sizeof(hreal2x32) == 12 && sizeof(hcplx2x32) == 20, "2x32 records");

} // namespace fs
