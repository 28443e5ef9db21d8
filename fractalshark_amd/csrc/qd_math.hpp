// qd_math.hpp -- four-component expansion arithmetic for the Gpu4x32 / Gpu4x64 direct kernels: q4<float> is the
// reference's GQF::gqf_real (a float4, FractalSharkLib/QuadFloat), q4<double> its GQD::gqd_real (QuadDouble) -- both
// ports of the quad-double algorithms of Hida, Li & Bailey (error-free two_sum / Dekker two_prod, renormalisation of a
// five-term expansion into four).  The two reference copies differ in three places, selected here by the scalar type:
//   * zero shortcuts in quick_two_sum / two_sum          -- QuadDouble only (inline.cuh:13-41; commented out in QuadFloat)
//   * the renormalisation's "is this term zero" branches -- QuadDouble only (gqd_basic.cuh:66-123; commented out in
//     gqf_basic.cuh:66-123, which always takes the all-non-zero arm)
//   * split(): double scales operands above 2^996 first; float multiplies by 2^12 + 1 unconditionally (inline.cuh)
// Compiled with -ffp-contract=off: every operation is one IEEE operation in source order (two_prod is Dekker's product
// on split halves, *not* an FMA, in both reference copies).  No CPU twin exists for these kernels (parity unpinned);
// the checker is oracle/gpu_ref_lp.cpp.
#pragma once

#include "hdr_math.hpp"

namespace fs {

template <class T> struct q4 {
    T x, y, z, w; // most significant first
};

template <class T> struct q4_traits;
template <> struct q4_traits<float> {
    static constexpr bool kZeroShortcuts = false;
};
template <> struct q4_traits<double> {
    static constexpr bool kZeroShortcuts = true;
};

// s = fl(a + b), err = the rounding error, assuming |a| >= |b|
template <class T> FS_HD T q_quick_two_sum(T a, T b, T &err)
{
    if (q4_traits<T>::kZeroShortcuts && b == T(0)) {
        err = T(0);
        return a + b;
    }
    const T s = a + b;
    err = b - (s - a);
    return s;
}
template <class T> FS_HD T q_two_sum(T a, T b, T &err)
{
    if (q4_traits<T>::kZeroShortcuts && (a == T(0) || b == T(0))) {
        err = T(0);
        return a + b;
    }
    const T s = a + b;
    const T bb = s - a;
    err = (a - (s - bb)) + (b - bb);
    return s;
}
FS_HD void q_split(float a, float &hi, float &lo)
{
    const float t = a * 4097.0f; // 2^12 + 1
    hi = t - (t - a);
    lo = a - hi;
}
FS_HD void q_split(double a, double &hi, double &lo)
{
    const double thresh = 6.69692879491417e+299; // 2^996
    if (a > thresh || a < -thresh) {
        a *= 3.7252902984619140625e-09; // 2^-28
        const double temp = 134217729.0 * a;
        hi = temp - (temp - a);
        lo = a - hi;
        hi *= 268435456.0; // 2^28
        lo *= 268435456.0;
    } else {
        const double temp = 134217729.0 * a; // 2^27 + 1
        hi = temp - (temp - a);
        lo = a - hi;
    }
}
template <class T> FS_HD T q_two_prod(T a, T b, T &err)
{
    T a_hi, a_lo, b_hi, b_lo;
    const T p = a * b;
    q_split(a, a_hi, a_lo);
    q_split(b, b_hi, b_lo);
    err = (a_hi * b_hi) - p + (a_hi * b_lo) + (a_lo * b_hi) + (a_lo * b_lo);
    return p;
}
template <class T> FS_HD T q_two_sqr(T a, T &err)
{
    T hi, lo;
    const T q = a * a;
    q_split(a, hi, lo);
    err = ((hi * hi - q) + T(2) * hi * lo) + lo * lo;
    return q;
}
template <class T> FS_HD void q_three_sum(T &a, T &b, T &c)
{
    T t1, t2, t3;
    t1 = q_two_sum(a, b, t2);
    a = q_two_sum(c, t1, t3);
    b = q_two_sum(t2, t3, c);
}
template <class T> FS_HD void q_three_sum2(T &a, T &b, T &c)
{
    T t1, t2, t3;
    t1 = q_two_sum(a, b, t2);
    a = q_two_sum(c, t1, t3);
    b = t2 + t3;
}

// renorm(c0..c4): five terms into four (g{qf,qd}_basic.cuh:66-123)
template <class T> FS_HD void q_renorm(T &c0, T &c1, T &c2, T &c3, T &c4)
{
    T s0, s1, s2 = T(0), s3 = T(0);
    s0 = q_quick_two_sum(c3, c4, c4);
    s0 = q_quick_two_sum(c2, s0, c3);
    s0 = q_quick_two_sum(c1, s0, c2);
    c0 = q_quick_two_sum(c0, s0, c1);
    s0 = c0;
    s1 = c1;
    s0 = q_quick_two_sum(c0, c1, s1);
    if (!q4_traits<T>::kZeroShortcuts || s1 != T(0)) {
        s1 = q_quick_two_sum(s1, c2, s2);
        if (!q4_traits<T>::kZeroShortcuts || s2 != T(0)) {
            s2 = q_quick_two_sum(s2, c3, s3);
            if (!q4_traits<T>::kZeroShortcuts || s3 != T(0))
                s3 += c4;
            else
                s2 += c4;
        } else {
            s1 = q_quick_two_sum(s1, c3, s2);
            if (s2 != T(0))
                s2 = q_quick_two_sum(s2, c4, s3);
            else
                s1 = q_quick_two_sum(s1, c4, s2);
        }
    } else {
        s0 = q_quick_two_sum(s0, c2, s1);
        if (s1 != T(0)) {
            s1 = q_quick_two_sum(s1, c3, s2);
            if (s2 != T(0))
                s2 = q_quick_two_sum(s2, c4, s3);
            else
                s1 = q_quick_two_sum(s1, c4, s2);
        } else {
            s0 = q_quick_two_sum(s0, c3, s1);
            if (s1 != T(0))
                s1 = q_quick_two_sum(s1, c4, s2);
            else
                s0 = q_quick_two_sum(s0, c4, s1);
        }
    }
    c0 = s0;
    c1 = s1;
    c2 = s2;
    c3 = s3;
}

// sloppy_add (operator+), g{qf,qd}_basic.cuh:177-233
template <class T> FS_HD q4<T> operator+(const q4<T> &a, const q4<T> &b)
{
    T s0 = a.x + b.x, s1 = a.y + b.y, s2 = a.z + b.z, s3 = a.w + b.w;
    const T v0 = s0 - a.x, v1 = s1 - a.y, v2 = s2 - a.z, v3 = s3 - a.w;
    T u0 = s0 - v0, u1 = s1 - v1, u2 = s2 - v2, u3 = s3 - v3;
    const T w0 = a.x - u0, w1 = a.y - u1, w2 = a.z - u2, w3 = a.w - u3;
    u0 = b.x - v0;
    u1 = b.y - v1;
    u2 = b.z - v2;
    u3 = b.w - v3;
    T t0 = w0 + u0, t1 = w1 + u1, t2 = w2 + u2;
    const T t3 = w3 + u3;
    s1 = q_two_sum(s1, t0, t0);
    q_three_sum(s2, t0, t1);
    q_three_sum2(s3, t0, t2);
    t0 = t0 + t1 + t3;
    q_renorm(s0, s1, s2, s3, t0);
    return q4<T>{s0, s1, s2, s3};
}
template <class T> FS_HD q4<T> operator-(const q4<T> &a) { return q4<T>{-a.x, -a.y, -a.z, -a.w}; }
template <class T> FS_HD q4<T> operator-(const q4<T> &a, const q4<T> &b) { return a + (-b); } // a + negative(b)
template <class T> FS_HD q4<T> q_mul_pwr2(const q4<T> &a, T b) { return q4<T>{a.x * b, a.y * b, a.z * b, a.w * b}; }

// quad * scalar, g{qf,qd}_basic.cuh:267-292
template <class T> FS_HD q4<T> operator*(const q4<T> &a, T b)
{
    T p0, p1, p2, p3, q0, q1, q2, s0, s1, s2, s3, s4;
    p0 = q_two_prod(a.x, b, q0);
    p1 = q_two_prod(a.y, b, q1);
    p2 = q_two_prod(a.z, b, q2);
    p3 = a.w * b;
    s0 = p0;
    s1 = q_two_sum(q0, p1, s2);
    q_three_sum(s2, q1, p2);
    q_three_sum2(q1, q2, p3);
    s3 = q1;
    s4 = q2 + p2;
    q_renorm(s0, s1, s2, s3, s4);
    return q4<T>{s0, s1, s2, s3};
}

// sloppy_mul (operator*), g{qf,qd}_basic.cuh:300-349
template <class T> FS_HD q4<T> operator*(const q4<T> &a, const q4<T> &b)
{
    T p0, p1, p2, p3, p4, p5, q0, q1, q2, q3, q4v, q5, t0, t1, s0, s1, s2;
    p0 = q_two_prod(a.x, b.x, q0);
    p1 = q_two_prod(a.x, b.y, q1);
    p2 = q_two_prod(a.y, b.x, q2);
    p3 = q_two_prod(a.x, b.z, q3);
    p4 = q_two_prod(a.y, b.y, q4v);
    p5 = q_two_prod(a.z, b.x, q5);
    q_three_sum(p1, p2, q0);
    q_three_sum(p2, q1, q2);
    q_three_sum(p3, p4, p5);
    s0 = q_two_sum(p2, p3, t0);
    s1 = q_two_sum(q1, p4, t1);
    s2 = q2 + p5;
    s1 = q_two_sum(s1, t0, t0);
    s2 += (t0 + t1);
    s1 = s1 + (a.x * b.w + a.y * b.z + a.z * b.y + a.w * b.x + q0 + q3 + q4v + q5);
    q_renorm(p0, p1, s0, s1, s2);
    return q4<T>{p0, p1, s0, s1};
}

// sqr, g{qf,qd}_basic.cuh:351-393
template <class T> FS_HD q4<T> q_sqr(const q4<T> &a)
{
    T p0, p1, p2, p3, p4, p5, q0, q1, q2, q3, s0, s1, t0, t1;
    p0 = q_two_sqr(a.x, q0);
    p1 = q_two_prod(T(2) * a.x, a.y, q1);
    p2 = q_two_prod(T(2) * a.x, a.z, q2);
    p3 = q_two_sqr(a.y, q3);
    p1 = q_two_sum(q0, p1, q0);
    q0 = q_two_sum(q0, q1, q1);
    p2 = q_two_sum(p2, p3, p3);
    s0 = q_two_sum(q0, p2, t0);
    s1 = q_two_sum(q1, p3, t1);
    s1 = q_two_sum(s1, t0, t0);
    t0 += t1;
    s1 = q_quick_two_sum(s1, t0, t0);
    p2 = q_quick_two_sum(s0, s1, t1);
    p3 = q_quick_two_sum(t1, t0, q0);
    p4 = T(2) * a.x * a.w;
    p5 = T(2) * a.y * a.z;
    p4 = q_two_sum(p4, p5, p5);
    q2 = q_two_sum(q2, q3, q3);
    t0 = q_two_sum(p4, q2, t1);
    t1 = t1 + p5 + q3;
    p3 = q_two_sum(p3, t0, p4);
    p4 = p4 + q0 + t1;
    q_renorm(p0, p1, p2, p3, p4);
    return q4<T>{p0, p1, p2, p3};
}

// operator<=(quad, quad) and operator<=(quad, scalar), g{qf,qd}_basic.cuh:513-533
template <class T> FS_HD bool operator<=(const q4<T> &a, const q4<T> &b)
{
    return a.x < b.x || (a.x == b.x && (a.y < b.y || (a.y == b.y && (a.z < b.z || (a.z == b.z && a.w <= b.w)))));
}
template <class T> FS_HD bool operator<=(const q4<T> &a, T b) { return a.x < b || (a.x == b && a.y <= T(0)); }

} // namespace fs
