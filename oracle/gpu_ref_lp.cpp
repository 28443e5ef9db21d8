// oracle/gpu_ref_lp.cpp -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// CPU restatement of the reference's direct low-precision CUDA kernels that have no CPU RenderAlgorithm twin:
//   mandel_1x_float<.., iteration_precision>   FractalSharkGpuLib/LowPrecisionKernels.cuh:682-777   (Gpu1x32)
//   mandel_2x_float<.., iteration_precision>   :384-555, float-float via HpSharkFloatLib/dblflt.cuh:86-214 (Gpu2x32)
//   mandel_2x_double                           :171-290, double-double via HpSharkFloatLib/dbldbl.cuh:82-200 (Gpu2x64)
// The reference has no CPU implementation, no golden and nvcc is not in this image, so this cannot be checked against
// an execution of the reference.  Conventions (same as the HIP kernels it checks): every __f*_rn / __d*_rn intrinsic is
// one IEEE operation; __fmaf_rd is libm's fmaf under FE_DOWNWARD; un-annotated expressions are evaluated in source order
// without contraction.  Build: part of liboracle.so, this file with -frounding-math.
#include <atomic>
#include <cfenv>
#include <cmath>
#include <cstdint>

extern "C" uint32_t orc_get_row_step(void);

namespace {

template <class T> struct DW {
    T h, t;
};
inline float Fma(float a, float b, float c) { return fmaf(a, b, c); }
inline double Fma(double a, double b, double c) { return fma(a, b, c); }

template <class T> DW<T> TwoSum(T a, T b)
{
    DW<T> z;
    z.h = a + b;
    T t1 = z.h - a;
    T t2 = z.h - t1;
    t1 = b - t1;
    t2 = a - t2;
    z.t = t1 + t2;
    return z;
}
template <class T> DW<T> Add(DW<T> a, DW<T> b)
{
    T t1 = a.h + b.h;
    T t2 = t1 + -a.h;
    T t3 = (a.h + (t2 - t1)) + (b.h + -t2);
    T t4 = a.t + b.t;
    t2 = t4 + -a.t;
    T t5 = (a.t + (t2 - t4)) + (b.t + -t2);
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    const T e = t4 + t3;
    return DW<T>{e, (t4 - e) + t3};
}
template <class T> DW<T> Sub(DW<T> a, DW<T> b)
{
    T t1 = a.h + -b.h;
    T t2 = t1 + -a.h;
    T t3 = (a.h + (t2 - t1)) + -(b.h + t2);
    T t4 = a.t + -b.t;
    t2 = t4 + -a.t;
    T t5 = (a.t + (t2 - t4)) + -(b.t + t2);
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    const T e = t4 + t3;
    return DW<T>{e, (t4 - e) + t3};
}
template <class T> DW<T> Mul(DW<T> a, DW<T> b)
{
    DW<T> t;
    t.h = a.h * b.h;
    t.t = Fma(a.h, b.h, -t.h);
    t.t = Fma(a.t, b.t, t.t);
    t.t = Fma(a.h, b.t, t.t);
    t.t = Fma(a.t, b.h, t.t);
    const T e = t.h + t.t;
    return DW<T>{e, (t.h - e) + t.t};
}
inline DW<float> Mul2x(DW<float> a, DW<float> b)
{
    DW<float> z = Mul(a, b);
    z.t = z.t * 2.0f;
    z.h = z.h * 2.0f;
    return z;
}
inline DW<float> Sqr(DW<float> a) // sqr_dblflt
{
    DW<float> t;
    t.h = a.h * a.h;
    t.t = Fma(a.h, a.h, -t.h);
    t.t = Fma(a.t, a.t, t.t);
    const float e0 = a.h * a.t;
    t.t = Fma(2.0f, e0, t.t);
    const float e = t.h + t.t;
    return DW<float>{e, (t.h - e) + t.t};
}
inline DW<double> Sqr(DW<double> a) // sqr_dbldbl
{
    DW<double> t;
    t.h = a.h * a.h;
    t.t = Fma(a.h, a.h, -t.h);
    t.t = Fma(a.t, a.t, t.t);
    t.t = Fma(a.h, a.t, t.t);
    t.t = Fma(a.t, a.h, t.t);
    const double e = t.h + t.t;
    return DW<double>{e, (t.h - e) + t.t};
}

} // namespace

extern "C" {

// coords = {cx, cy, dx, dy}.  Rows y0 <= R < y1 of the (flipped) output are produced; out[R * pitch + X].
void orc_gpu_direct_1x32(uint32_t *out, uint32_t pitch, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                         const float coords[4], uint32_t n_iterations, int iteration_precision)
{
    const float cx = coords[0], cy = coords[1], dx = coords[2], dy = coords[3];
    const uint32_t step = orc_get_row_step() ? orc_get_row_step() : 1;
    for (uint32_t R = y0; R < y1; R += step) {
        const int Y = (int)height - 1 - (int)R;
        for (uint32_t X = 0; X < width; X++) {
            const float x0 = cx + dx * (float)(int)X;
            const float y0f = cy + dy * (float)Y;
            float x = 0.0f, y = 0.0f;
            const uint32_t n = n_iterations - (uint32_t)(iteration_precision - 1);
            uint32_t iter = 0;
            while (x * x + y * y < 4.0f && iter < n) {
                fesetround(FE_DOWNWARD);
                for (int k = 0; k < iteration_precision; k++) {
                    const float ytemp = fmaf(-y, y, x0);
                    const float xtemp = fmaf(x, x, ytemp);
                    const float xtemp2 = 2.0f * x;
                    y = fmaf(xtemp2, y, y0f);
                    x = xtemp;
                }
                fesetround(FE_TONEAREST);
                iter += (uint32_t)iteration_precision;
            }
            out[(size_t)R * pitch + X] = iter;
        }
    }
}

// coords = {cx.head, cx.tail, cy.head, cy.tail, dx.head, dx.tail, dy.head, dy.tail}
void orc_gpu_direct_2x32(uint32_t *out, uint32_t pitch, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                         const float coords[8], uint32_t n_iterations, int iteration_precision)
{
    const uint32_t step = orc_get_row_step() ? orc_get_row_step() : 1;
    const DW<float> cx2 = TwoSum(coords[0], coords[1]), cy2 = TwoSum(coords[2], coords[3]);
    const DW<float> dx2 = TwoSum(coords[4], coords[5]), dy2 = TwoSum(coords[6], coords[7]);
    for (uint32_t R = y0; R < y1; R += step) {
        const int Y = (int)height - 1 - (int)R;
        for (uint32_t X = 0; X < width; X++) {
            const DW<float> X2 = TwoSum((float)(int)X, 0.0f), Y2 = TwoSum((float)Y, 0.0f);
            const DW<float> x0 = Add(cx2, Mul(dx2, X2));
            const DW<float> y0d = Add(cy2, Mul(dy2, Y2));
            DW<float> x{0, 0}, y{0, 0}, zrsqr{0, 0}, zisqr{0, 0};
            uint32_t iter = 0;
            while (zrsqr.h + zisqr.h < 4.0f && iter < n_iterations) {
                for (int k = 0; k < iteration_precision; k++) {
                    y = Mul2x(x, y);
                    y = Add(y, y0d);
                    x = Sub(zrsqr, zisqr);
                    x = Add(x, x0);
                    zrsqr = Sqr(x);
                    zisqr = Sqr(y);
                }
                iter += (uint32_t)iteration_precision;
            }
            out[(size_t)R * pitch + X] = iter;
        }
    }
}

void orc_gpu_direct_2x64(uint32_t *out, uint32_t pitch, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                         const double coords[8], uint32_t n_iterations)
{
    const uint32_t step = orc_get_row_step() ? orc_get_row_step() : 1;
    const DW<double> cx2 = TwoSum(coords[0], coords[1]), cy2 = TwoSum(coords[2], coords[3]);
    const DW<double> dx2 = TwoSum(coords[4], coords[5]), dy2 = TwoSum(coords[6], coords[7]);
    for (uint32_t R = y0; R < y1; R += step) {
        const int Y = (int)height - 1 - (int)R;
        for (uint32_t X = 0; X < width; X++) {
            const DW<double> X2 = TwoSum((double)(int)X, 0.0), Y2 = TwoSum((double)Y, 0.0);
            const DW<double> x0 = Add(cx2, Mul(dx2, X2));
            const DW<double> y0d = Add(cy2, Mul(dy2, Y2));
            DW<double> x = TwoSum(0.0, 0.0), y = TwoSum(0.0, 0.0);
            const DW<double> two = TwoSum(2.0, 0.0);
            DW<double> zrsqr = Sqr(x), zisqr = Sqr(y);
            uint32_t iter = 0;
            while (zrsqr.h + zisqr.h < 4.0 && iter < n_iterations) {
                const DW<double> xtemp = Add(Sub(zrsqr, zisqr), x0);
                y = Add(Mul(two, Mul(x, y)), y0d);
                x = xtemp;
                zrsqr = Sqr(x);
                zisqr = Sqr(y);
                iter++;
            }
            out[(size_t)R * pitch + X] = iter;
        }
    }
}

} // extern "C"
