// oracle/gpu_ref_qd.cpp -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// CPU restatement of the reference's quad-precision direct CUDA kernels, which have no CPU RenderAlgorithm twin:
//   mandel_4x_float    FractalSharkGpuLib/LowPrecisionKernels.cuh:5-75     (Gpu4x32, GQF::gqf_real = float4)
//   mandel_4x_double   FractalSharkGpuLib/LowPrecisionKernels.cuh:77-140   (Gpu4x64, GQD::gqd_real = double4)
// on the expansion arithmetic of FractalSharkLib/QuadFloat/{inline,gqf_basic}.cuh and QuadDouble/{inline,gqd_basic}.cuh
// (ports of Hida / Li / Bailey's quad-double library).  The reference has no CPU implementation of these kernels, no
// golden and nvcc is not in this image, so this cannot be checked against an execution of the reference.  Conventions
// (same as the HIP kernels it checks): every operation is one IEEE operation, evaluated in source order without
// contraction; __fmul_rn / __dmul_rn are plain products.  Nothing in the product path may link, import or call this file.
//
// The float and double copies of the library differ and are restated separately below (struct F4 / struct D4):
//   QuadDouble keeps the zero shortcuts of quick_two_sum / two_sum (inline.cuh:13-41) and the zero-term branches of
//   renorm (gqd_basic.cuh:66-123) and scales huge operands in split (inline.cuh:93-111); QuadFloat has all of those
//   commented out (inline.cuh:19-52,128-151; gqf_basic.cuh:66-123).
#include <cmath>
#include <cstdint>

extern "C" uint32_t orc_get_row_step(void);

namespace {

// ------------------------------------------------------------------ QuadFloat (GQF)
struct F4 {
    float v[4];
};
struct OpsF {
    using S = float;
    using Q = F4;
    static S qts(S a, S b, S *e) // quick_two_sum, QuadFloat/inline.cuh:19-35
    {
        const S s = a + b;
        *e = b - (s - a);
        return s;
    }
    static S ts(S a, S b, S *e) // two_sum, :37-52
    {
        const S s = a + b;
        const S bb = s - a;
        *e = (a - (s - bb)) + (b - bb);
        return s;
    }
    static void split(S a, S *hi, S *lo) // :128-151 (the active lines)
    {
        const S t = a * 4097.0f;
        *hi = t - (t - a);
        *lo = a - *hi;
    }
    // renorm(c0..c4), gqf_basic.cuh:66-123: unconditional chain
    static void renorm5(S &c0, S &c1, S &c2, S &c3, S &c4)
    {
        S s0, s1, s2 = 0.0f, s3 = 0.0f;
        s0 = qts(c3, c4, &c4);
        s0 = qts(c2, s0, &c3);
        s0 = qts(c1, s0, &c2);
        c0 = qts(c0, s0, &c1);
        s0 = c0;
        s1 = c1;
        s0 = qts(c0, c1, &s1);
        s1 = qts(s1, c2, &s2);
        s2 = qts(s2, c3, &s3);
        s3 += c4;
        c0 = s0;
        c1 = s1;
        c2 = s2;
        c3 = s3;
    }
};

// ------------------------------------------------------------------ QuadDouble (GQD)
struct D4 {
    double v[4];
};
struct OpsD {
    using S = double;
    using Q = D4;
    static S qts(S a, S b, S *e) // QuadDouble/inline.cuh:13-26
    {
        if (b == 0.0) {
            *e = 0.0;
            return a + b;
        }
        const S s = a + b;
        *e = b - (s - a);
        return s;
    }
    static S ts(S a, S b, S *e) // :28-41
    {
        if (a == 0.0 || b == 0.0) {
            *e = 0.0;
            return a + b;
        }
        const S s = a + b;
        const S bb = s - a;
        *e = (a - (s - bb)) + (b - bb);
        return s;
    }
    static void split(S a, S *hi, S *lo) // :93-111
    {
        S temp;
        if (a > 6.69692879491417e+299 || a < -6.69692879491417e+299) {
            a *= 3.7252902984619140625e-09;
            temp = 134217729.0 * a;
            *hi = temp - (temp - a);
            *lo = a - *hi;
            *hi *= 268435456.0;
            *lo *= 268435456.0;
        } else {
            temp = 134217729.0 * a;
            *hi = temp - (temp - a);
            *lo = a - *hi;
        }
    }
    // renorm(c0..c4), gqd_basic.cuh:66-123: skips over zero terms
    static void renorm5(S &c0, S &c1, S &c2, S &c3, S &c4)
    {
        S s0, s1, s2 = 0.0, s3 = 0.0;
        s0 = qts(c3, c4, &c4);
        s0 = qts(c2, s0, &c3);
        s0 = qts(c1, s0, &c2);
        c0 = qts(c0, s0, &c1);
        s0 = c0;
        s1 = c1;
        s0 = qts(c0, c1, &s1);
        if (s1 != 0.0) {
            s1 = qts(s1, c2, &s2);
            if (s2 != 0.0) {
                s2 = qts(s2, c3, &s3);
                if (s3 != 0.0)
                    s3 += c4;
                else
                    s2 += c4;
            } else {
                s1 = qts(s1, c3, &s2);
                if (s2 != 0.0)
                    s2 = qts(s2, c4, &s3);
                else
                    s1 = qts(s1, c4, &s2);
            }
        } else {
            s0 = qts(s0, c2, &s1);
            if (s1 != 0.0) {
                s1 = qts(s1, c3, &s2);
                if (s2 != 0.0)
                    s2 = qts(s2, c4, &s3);
                else
                    s1 = qts(s1, c4, &s2);
            } else {
                s0 = qts(s0, c3, &s1);
                if (s1 != 0.0)
                    s1 = qts(s1, c4, &s2);
                else
                    s0 = qts(s0, c4, &s1);
            }
        }
        c0 = s0;
        c1 = s1;
        c2 = s2;
        c3 = s3;
    }
};

// ------------------------------------------------------------------ shared shapes (identical text in both reference copies)
template <class O> struct QArith {
    using S = typename O::S;
    using Q = typename O::Q;

    static S two_prod(S a, S b, S *err) // inline.cuh "two_prod": Dekker product on split halves
    {
        S ah, al, bh, bl;
        const S p = a * b;
        O::split(a, &ah, &al);
        O::split(b, &bh, &bl);
        *err = (ah * bh) - p + (ah * bl) + (al * bh) + (al * bl);
        return p;
    }
    static S two_sqr(S a, S *err)
    {
        S hi, lo;
        const S q = a * a;
        O::split(a, &hi, &lo);
        *err = ((hi * hi - q) + S(2) * hi * lo) + lo * lo;
        return q;
    }
    static void three_sum(S &a, S &b, S &c) // g*_basic.cuh:136-143
    {
        S t1, t2, t3;
        t1 = O::ts(a, b, &t2);
        a = O::ts(c, t1, &t3);
        b = O::ts(t2, t3, &c);
    }
    static void three_sum2(S &a, S &b, S &c) // :145-152
    {
        S t1, t2, t3;
        t1 = O::ts(a, b, &t2);
        a = O::ts(c, t1, &t3);
        b = t2 + t3;
    }
    static Q add(const Q &a, const Q &b) // sloppy_add, :177-225
    {
        S s[4], t[4], vv[4], u[4], w[4];
        for (int i = 0; i < 4; i++)
            s[i] = a.v[i] + b.v[i];
        for (int i = 0; i < 4; i++)
            vv[i] = s[i] - a.v[i];
        for (int i = 0; i < 4; i++)
            u[i] = s[i] - vv[i];
        for (int i = 0; i < 4; i++)
            w[i] = a.v[i] - u[i];
        for (int i = 0; i < 4; i++)
            u[i] = b.v[i] - vv[i];
        for (int i = 0; i < 4; i++)
            t[i] = w[i] + u[i];
        s[1] = O::ts(s[1], t[0], &t[0]);
        three_sum(s[2], t[0], t[1]);
        three_sum2(s[3], t[0], t[2]);
        t[0] = t[0] + t[1] + t[3];
        O::renorm5(s[0], s[1], s[2], s[3], t[0]);
        return Q{{s[0], s[1], s[2], s[3]}};
    }
    static Q neg(const Q &a) { return Q{{-a.v[0], -a.v[1], -a.v[2], -a.v[3]}}; }
    static Q sub(const Q &a, const Q &b) { return add(a, neg(b)); } // :254-258
    static Q mul_pwr2(const Q &a, S b) { return Q{{a.v[0] * b, a.v[1] * b, a.v[2] * b, a.v[3] * b}}; }
    static Q mul_s(const Q &a, S b) // quad * scalar, :267-292
    {
        S p0, p1, p2, p3, q0, q1, q2, s0, s1, s2, s3, s4;
        p0 = two_prod(a.v[0], b, &q0);
        p1 = two_prod(a.v[1], b, &q1);
        p2 = two_prod(a.v[2], b, &q2);
        p3 = a.v[3] * b;
        s0 = p0;
        s1 = O::ts(q0, p1, &s2);
        three_sum(s2, q1, p2);
        three_sum2(q1, q2, p3);
        s3 = q1;
        s4 = q2 + p2;
        O::renorm5(s0, s1, s2, s3, s4);
        return Q{{s0, s1, s2, s3}};
    }
    static Q mul(const Q &a, const Q &b) // sloppy_mul, :300-344
    {
        S p0, p1, p2, p3, p4, p5, q0, q1, q2, q3, q4, q5, t0, t1, s0, s1, s2;
        p0 = two_prod(a.v[0], b.v[0], &q0);
        p1 = two_prod(a.v[0], b.v[1], &q1);
        p2 = two_prod(a.v[1], b.v[0], &q2);
        p3 = two_prod(a.v[0], b.v[2], &q3);
        p4 = two_prod(a.v[1], b.v[1], &q4);
        p5 = two_prod(a.v[2], b.v[0], &q5);
        three_sum(p1, p2, q0);
        three_sum(p2, q1, q2);
        three_sum(p3, p4, p5);
        s0 = O::ts(p2, p3, &t0);
        s1 = O::ts(q1, p4, &t1);
        s2 = q2 + p5;
        s1 = O::ts(s1, t0, &t0);
        s2 += (t0 + t1);
        const S m03 = a.v[0] * b.v[3], m12 = a.v[1] * b.v[2], m21 = a.v[2] * b.v[1], m30 = a.v[3] * b.v[0];
        s1 = s1 + (m03 + m12 + m21 + m30 + q0 + q3 + q4 + q5);
        O::renorm5(p0, p1, s0, s1, s2);
        return Q{{p0, p1, s0, s1}};
    }
    static Q sqr(const Q &a) // :351-393
    {
        S p0, p1, p2, p3, p4, p5, q0, q1, q2, q3, s0, s1, t0, t1;
        p0 = two_sqr(a.v[0], &q0);
        p1 = two_prod(S(2) * a.v[0], a.v[1], &q1);
        p2 = two_prod(S(2) * a.v[0], a.v[2], &q2);
        p3 = two_sqr(a.v[1], &q3);
        p1 = O::ts(q0, p1, &q0);
        q0 = O::ts(q0, q1, &q1);
        p2 = O::ts(p2, p3, &p3);
        s0 = O::ts(q0, p2, &t0);
        s1 = O::ts(q1, p3, &t1);
        s1 = O::ts(s1, t0, &t0);
        t0 += t1;
        s1 = O::qts(s1, t0, &t0);
        p2 = O::qts(s0, s1, &t1);
        p3 = O::qts(t1, t0, &q0);
        p4 = S(2) * a.v[0] * a.v[3];
        p5 = S(2) * a.v[1] * a.v[2];
        p4 = O::ts(p4, p5, &p5);
        q2 = O::ts(q2, q3, &q3);
        t0 = O::ts(p4, q2, &t1);
        t1 = t1 + p5 + q3;
        p3 = O::ts(p3, t0, &p4);
        p4 = p4 + q0 + t1;
        O::renorm5(p0, p1, p2, p3, p4);
        return Q{{p0, p1, p2, p3}};
    }
    static bool le(const Q &a, const Q &b) // operator<=(quad, quad), :513-519
    {
        for (int i = 0; i < 3; i++) {
            if (a.v[i] < b.v[i])
                return true;
            if (!(a.v[i] == b.v[i]))
                return false;
        }
        return a.v[3] <= b.v[3];
    }
    static bool le_s(const Q &a, S b) { return a.v[0] < b || (a.v[0] == b && a.v[1] <= S(0)); } // :530-533
};

} // namespace

extern "C" {

// coords = {cx.x..w, cy.x..w, dx.x..w, dy.x..w}; rows are written flipped (ConvertLocToIndex(X, height - Y - 1, width)).
void orc_gpu_direct_4x32(uint32_t *out, uint32_t pitch, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                         const float coords[16], uint32_t n_iterations)
{
    using A = QArith<OpsF>;
    const uint32_t step = orc_get_row_step() ? orc_get_row_step() : 1;
    F4 c[4];
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++)
            c[i].v[k] = coords[4 * i + k];
    const F4 four{{4.0f, 0.0f, 0.0f, 0.0f}};
    for (uint32_t R = y0; R < y1; R += step) {
        const int Y = (int)height - 1 - (int)R;
        for (uint32_t X = 0; X < width; X++) {
            F4 x{{0.0f, 0.0f, 0.0f, 0.0f}}, y = x;
            const F4 yq = A::add(c[1], A::mul(c[3], F4{{(float)Y, 0.0f, 0.0f, 0.0f}}));
            const F4 xq = A::add(c[0], A::mul(c[2], F4{{(float)(int)X, 0.0f, 0.0f, 0.0f}}));
            F4 zr = A::sqr(x), zi = A::sqr(y);
            uint32_t iter = 0;
            while (A::le(A::add(zr, zi), four) && iter < n_iterations) {
                y = A::mul(x, y);
                y = A::mul_pwr2(y, 2.0f);
                y = A::add(y, yq);
                x = A::add(A::sub(zr, zi), xq);
                zr = A::sqr(x);
                zi = A::sqr(y);
                iter++;
            }
            out[(size_t)R * pitch + X] = iter;
        }
    }
}

void orc_gpu_direct_4x64(uint32_t *out, uint32_t pitch, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                         const double coords[16], uint32_t n_iterations)
{
    using A = QArith<OpsD>;
    const uint32_t step = orc_get_row_step() ? orc_get_row_step() : 1;
    D4 c[4];
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++)
            c[i].v[k] = coords[4 * i + k];
    for (uint32_t R = y0; R < y1; R += step) {
        const int Y = (int)height - 1 - (int)R;
        for (uint32_t X = 0; X < width; X++) {
            D4 x{{0.0, 0.0, 0.0, 0.0}}, y = x;
            const D4 yq = A::add(c[1], A::mul_s(c[3], (double)Y));
            const D4 xq = A::add(c[0], A::mul_s(c[2], (double)(int)X));
            D4 zr = A::mul(x, x), zi = A::mul(y, y);
            uint32_t iter = 0;
            while (A::le_s(A::add(zr, zi), 4.0) && iter < n_iterations) {
                y = A::mul(x, y);
                y = A::mul_s(y, 2.0);
                y = A::add(y, yq);
                x = A::add(A::sub(zr, zi), xq);
                zr = A::mul(x, x);
                zi = A::mul(y, y);
                iter++;
            }
            out[(size_t)R * pitch + X] = iter;
        }
    }
}

} // extern "C"
