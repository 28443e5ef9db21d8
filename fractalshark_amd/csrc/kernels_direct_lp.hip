// kernels_direct_lp.hip -- the direct (no reference orbit) low-precision kernels of LowPrecisionKernels.cuh that have no
// CPU RenderAlgorithm twin: Gpu1x32 (mandel_1x_float, :682-...), Gpu2x32 (mandel_2x_float, :384-555, float-float) and
// Gpu2x64 (mandel_2x_double, :171-290, double-double), plus Gpu4x32 (mandel_4x_float, :5-75, quad-float) and Gpu4x64
// (mandel_4x_double, :77-140, quad-double) on the expansion arithmetic of qd_math.hpp.  Compiled with -ffp-contract=off.
//
// These restate the CUDA kernels' source semantics: every __f*_rn / __d*_rn intrinsic is one IEEE operation, __fmaf_rd is
// a fused multiply-add rounded toward -infinity (OCML's rtn variant sets the hardware rounding mode around one v_fma),
// un-annotated expressions (x0 = cx + dx * X, the bailout sum) are evaluated in source order without contraction.  Output
// rows are flipped like the reference's ConvertLocToIndex(X, height - Y - 1, width); iteration_precision is the number of
// steps between bailout tests, and mandel_1x_float shortens n_iterations by iteration_precision - 1 (:712).
// Checker: oracle/gpu_ref_lp.cpp (parity unpinned, same conventions).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"
#include "qd_math.hpp"
#include "kernels.h"
#include "kernel_common.hpp"

using fs::q4;
using fs::q_mul_pwr2;
using fs::q_sqr;

extern "C" __device__ __attribute__((const)) float __ocml_fma_rtn_f32(float, float, float);

namespace {

template <class T> struct dw {
    T h, t;
};
template <class T> __device__ __forceinline__ T fma_rn(T a, T b, T c);
template <> __device__ __forceinline__ float fma_rn<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_rn<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

// add_float_to_dblflt / add_double_to_dbldbl (Knuth two-sum), dblflt.cuh:86-97, dbldbl.cuh:82-92
template <class T> __device__ __forceinline__ dw<T> two_sum(T a, T b)
{
    dw<T> z;
    z.h = a + b;
    T t1 = z.h - a;
    T t2 = z.h - t1;
    t1 = b - t1;
    t2 = a - t2;
    z.t = t1 + t2;
    return z;
}
// add_dblflt / add_dbldbl, dblflt.cuh:116-132, dbldbl.cuh:114-130
template <class T> __device__ __forceinline__ dw<T> dw_add(dw<T> a, dw<T> b)
{
    T t1 = a.h + b.h;
    T t2 = t1 - a.h;
    T t3 = (a.h + (t2 - t1)) + (b.h - t2);
    T t4 = a.t + b.t;
    t2 = t4 - a.t;
    const T t5 = (a.t + (t2 - t4)) + (b.t - t2);
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    const T e = t4 + t3;
    return dw<T>{e, (t4 - e) + t3};
}
// sub_dblflt / sub_dbldbl
template <class T> __device__ __forceinline__ dw<T> dw_sub(dw<T> a, dw<T> b)
{
    T t1 = a.h - b.h;
    T t2 = t1 - a.h;
    T t3 = (a.h + (t2 - t1)) - (b.h + t2);
    T t4 = a.t - b.t;
    t2 = t4 - a.t;
    const T t5 = (a.t + (t2 - t4)) - (b.t + t2);
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    const T e = t4 + t3;
    return dw<T>{e, (t4 - e) + t3};
}
// mul_dblflt / mul_dbldbl
template <class T> __device__ __forceinline__ dw<T> dw_mul(dw<T> a, dw<T> b)
{
    const T th = a.h * b.h;
    T tt = fma_rn<T>(a.h, b.h, -th);
    tt = fma_rn<T>(a.t, b.t, tt);
    tt = fma_rn<T>(a.h, b.t, tt);
    tt = fma_rn<T>(a.t, b.h, tt);
    const T e = th + tt;
    return dw<T>{e, (th - e) + tt};
}
// mul_dblflt2x, dblflt.cuh:178-192: the product with both parts doubled afterwards
__device__ __forceinline__ dw<float> dw_mul2x(dw<float> a, dw<float> b)
{
    dw<float> z = dw_mul(a, b);
    z.t = z.t * 2.0f;
    z.h = z.h * 2.0f;
    return z;
}
// sqr_dblflt, dblflt.cuh:194-214 (the cross term as one product doubled inside an fma)
__device__ __forceinline__ dw<float> dw_sqr(dw<float> a)
{
    const float th = a.h * a.h;
    float tt = fma_rn<float>(a.h, a.h, -th);
    tt = fma_rn<float>(a.t, a.t, tt);
    const float e0 = a.h * a.t;
    tt = fma_rn<float>(2.0f, e0, tt);
    const float e = th + tt;
    return dw<float>{e, (th - e) + tt};
}
// sqr_dbldbl, dbldbl.cuh:187-200 (the cross term as two fmas)
__device__ __forceinline__ dw<double> dw_sqr(dw<double> a)
{
    const double th = a.h * a.h;
    double tt = fma_rn<double>(a.h, a.h, -th);
    tt = fma_rn<double>(a.t, a.t, tt);
    tt = fma_rn<double>(a.h, a.t, tt);
    tt = fma_rn<double>(a.t, a.h, tt);
    const double e = th + tt;
    return dw<double>{e, (th - e) + tt};
}

// local row L of the (possibly banded) buffer holds output row R; the kernel's Y is counted from the other end
__device__ __forceinline__ bool lp_pixel(const FsFrame &f, uint32_t &X, uint32_t &L, int &Y)
{
    X = blockIdx.x * 64u + (threadIdx.x & 63u);
    L = blockIdx.y * 4u + (threadIdx.x >> 6);
    const uint32_t R = global_row(f, L);
    Y = (int)f.height - 1 - (int)R;
    return X < f.width && L < f.local_rows && R < f.height;
}

// mandel_1x_float<IterType, iteration_precision>
template <int IP, bool kStats, class IterT = uint32_t> __global__ void __launch_bounds__(256) k_direct_1x32(FsDirectLpArgs A)
{
    uint32_t X, L;
    int Y;
    uint64_t c_pt = 0, c_px = 0;
    if (lp_pixel(A.frame, X, L, Y)) {
        c_px = 1;
        const float cx = A.c32[0], cy = A.c32[1], dx = A.c32[2], dy = A.c32[3];
        const float x0 = cx + dx * (float)(int)X;
        const float y0 = cy + dy * (float)Y;
        float x = 0.0f, y = 0.0f;
        const IterT n = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi) - (IterT)(IP - 1);
        IterT iter = 0;
        while (x * x + y * y < 4.0f && iter < n) {
#pragma unroll
            for (int k = 0; k < IP; k++) {
                const float ytemp = __ocml_fma_rtn_f32(-y, y, x0);
                const float xtemp = __ocml_fma_rtn_f32(x, x, ytemp);
                const float xtemp2 = 2.0f * x;
                y = __ocml_fma_rtn_f32(xtemp2, y, y0);
                x = xtemp;
            }
            iter += IP;
        }
        if (kStats)
            c_pt = iter;
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, 0, 0, c_pt, c_px);
}

// mandel_2x_float<IterType, iteration_precision>
template <int IP, bool kStats, class IterT = uint32_t> __global__ void __launch_bounds__(256) k_direct_2x32(FsDirectLpArgs A)
{
    uint32_t X, L;
    int Y;
    uint64_t c_pt = 0, c_px = 0;
    if (lp_pixel(A.frame, X, L, Y)) {
        c_px = 1;
        const dw<float> cx2 = two_sum(A.c32[0], A.c32[1]), cy2 = two_sum(A.c32[2], A.c32[3]);
        const dw<float> dx2 = two_sum(A.c32[4], A.c32[5]), dy2 = two_sum(A.c32[6], A.c32[7]);
        const dw<float> X2 = two_sum((float)(int)X, 0.0f), Y2 = two_sum((float)Y, 0.0f);
        const dw<float> x0 = dw_add(cx2, dw_mul(dx2, X2));
        const dw<float> y0 = dw_add(cy2, dw_mul(dy2, Y2));
        dw<float> x{0.0f, 0.0f}, y{0.0f, 0.0f}, zrsqr{0.0f, 0.0f}, zisqr{0.0f, 0.0f};
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        IterT iter = 0;
        while (zrsqr.h + zisqr.h < 4.0f && iter < n_iterations) {
#pragma unroll
            for (int k = 0; k < IP; k++) {
                y = dw_mul2x(x, y);
                y = dw_add(y, y0);
                x = dw_sub(zrsqr, zisqr);
                x = dw_add(x, x0);
                zrsqr = dw_sqr(x);
                zisqr = dw_sqr(y);
            }
            iter += IP;
        }
        if (kStats)
            c_pt = iter;
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, 0, 0, c_pt, c_px);
}

// mandel_2x_double<IterType>
template <bool kStats, class IterT = uint32_t> __global__ void __launch_bounds__(256) k_direct_2x64(FsDirectLpArgs A)
{
    uint32_t X, L;
    int Y;
    uint64_t c_pt = 0, c_px = 0;
    if (lp_pixel(A.frame, X, L, Y)) {
        c_px = 1;
        const dw<double> cx2 = two_sum(A.c64[0], A.c64[1]), cy2 = two_sum(A.c64[2], A.c64[3]);
        const dw<double> dx2 = two_sum(A.c64[4], A.c64[5]), dy2 = two_sum(A.c64[6], A.c64[7]);
        const dw<double> X2 = two_sum((double)(int)X, 0.0), Y2 = two_sum((double)Y, 0.0);
        const dw<double> x0 = dw_add(cx2, dw_mul(dx2, X2));
        const dw<double> y0 = dw_add(cy2, dw_mul(dy2, Y2));
        dw<double> x = two_sum(0.0, 0.0), y = two_sum(0.0, 0.0);
        const dw<double> two = two_sum(2.0, 0.0);
        dw<double> zrsqr = dw_sqr(x), zisqr = dw_sqr(y);
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        IterT iter = 0;
        while (zrsqr.h + zisqr.h < 4.0 && iter < n_iterations) {
            const dw<double> xtemp = dw_add(dw_sub(zrsqr, zisqr), x0);
            y = dw_add(dw_mul(two, dw_mul(x, y)), y0);
            x = xtemp;
            zrsqr = dw_sqr(x);
            zisqr = dw_sqr(y);
            iter++;
        }
        if (kStats)
            c_pt = iter;
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, 0, 0, c_pt, c_px);
}

// mandel_4x_float (Gpu4x32), LowPrecisionKernels.cuh:5-75: quad-float; the pixel index enters as make_qf(X, 0, 0, 0), the
// doubling is mul_pwr2, squares use sqr(), the bailout compares two quads
template <bool kStats, class IterT = uint32_t> __global__ void __launch_bounds__(256) k_direct_4x32(FsDirectLpArgs A)
{
    using Q = q4<float>;
    uint32_t X, L;
    int Y;
    uint64_t c_pt = 0, c_px = 0;
    if (lp_pixel(A.frame, X, L, Y)) {
        c_px = 1;
        const Q cx{A.c32[0], A.c32[1], A.c32[2], A.c32[3]}, cy{A.c32[4], A.c32[5], A.c32[6], A.c32[7]};
        const Q dx{A.c32[8], A.c32[9], A.c32[10], A.c32[11]}, dy{A.c32[12], A.c32[13], A.c32[14], A.c32[15]};
        const Q y0 = cy + dy * Q{(float)Y, 0.0f, 0.0f, 0.0f};
        const Q x0 = cx + dx * Q{(float)(int)X, 0.0f, 0.0f, 0.0f};
        const Q four{4.0f, 0.0f, 0.0f, 0.0f};
        Q x{0.0f, 0.0f, 0.0f, 0.0f}, y = x;
        Q zrsqr = q_sqr(x), zisqr = q_sqr(y);
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        IterT iter = 0;
        while (zrsqr + zisqr <= four && iter < n_iterations) {
            y = x * y;
            y = q_mul_pwr2(y, 2.0f);
            y = y + y0;
            x = zrsqr - zisqr + x0;
            zrsqr = q_sqr(x);
            zisqr = q_sqr(y);
            iter++;
        }
        if (kStats)
            c_pt = iter;
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, 0, 0, c_pt, c_px);
}

// mandel_4x_double (Gpu4x64), LowPrecisionKernels.cuh:77-140: quad-double; the pixel index and the factor two are scalars
// (quad * double), squares are full products, the bailout compares against the scalar 4.0
template <bool kStats, class IterT = uint32_t> __global__ void __launch_bounds__(256) k_direct_4x64(FsDirectLpArgs A)
{
    using Q = q4<double>;
    uint32_t X, L;
    int Y;
    uint64_t c_pt = 0, c_px = 0;
    if (lp_pixel(A.frame, X, L, Y)) {
        c_px = 1;
        const Q cx{A.c64[0], A.c64[1], A.c64[2], A.c64[3]}, cy{A.c64[4], A.c64[5], A.c64[6], A.c64[7]};
        const Q dx{A.c64[8], A.c64[9], A.c64[10], A.c64[11]}, dy{A.c64[12], A.c64[13], A.c64[14], A.c64[15]};
        const Q y0 = cy + dy * (double)Y;
        const Q x0 = cx + dx * (double)(int)X;
        Q x{0.0, 0.0, 0.0, 0.0}, y = x;
        Q zrsqr = x * x, zisqr = y * y;
        const IterT n_iterations = iter_cap<IterT>(A.n_iterations, A.n_iterations_hi);
        IterT iter = 0;
        while (zrsqr + zisqr <= 4.0 && iter < n_iterations) {
            y = x * y;
            y = y * 2.0;
            y = y + y0;
            x = zrsqr - zisqr + x0;
            zrsqr = x * x;
            zisqr = y * y;
            iter++;
        }
        if (kStats)
            c_pt = iter;
        store_iter(A.out, A.frame, L, X, iter);
    }
    if (kStats)
        add_stats(A.stats, 0, 0, c_pt, c_px);
}

} // namespace

bool fsk_direct_lp(const FsDirectLpArgs &A, int kind, int iteration_precision, bool stats, hipStream_t s)
{
    const dim3 b(256);
    const dim3 g((A.frame.width + 63) / 64, (A.frame.local_rows + 3) / 4);
    const bool wide = A.frame.wide != 0u; // iteration cap of 2^32 or above: the instantiations counting in 64 bits
#define FS_LP_IP(K, IPV)                                                                                            \
    case IPV:                                                                                                       \
        if (wide)                                                                                                   \
            hipLaunchKernelGGL((K<IPV, false, uint64_t>), g, b, 0, s, A);                                           \
        else if (stats)                                                                                             \
            hipLaunchKernelGGL((K<IPV, true>), g, b, 0, s, A);                                                      \
        else                                                                                                        \
            hipLaunchKernelGGL((K<IPV, false>), g, b, 0, s, A);                                                     \
        return true
    if (kind == 0) { // Gpu1x32: the reference instantiates 1, 4, 8, 16 (GPU_Render.cu:704-738)
        switch (iteration_precision) {
            FS_LP_IP(k_direct_1x32, 1);
            FS_LP_IP(k_direct_1x32, 4);
            FS_LP_IP(k_direct_1x32, 8);
            FS_LP_IP(k_direct_1x32, 16);
        default:
            return false; // the reference's switch does nothing for other values
        }
    } else if (kind == 1) { // Gpu2x32
        switch (iteration_precision) {
            FS_LP_IP(k_direct_2x32, 1);
            FS_LP_IP(k_direct_2x32, 4);
            FS_LP_IP(k_direct_2x32, 8);
            FS_LP_IP(k_direct_2x32, 16);
        default:
            return false;
        }
    } else if (kind == 2) {
        if (wide)
            hipLaunchKernelGGL((k_direct_2x64<false, uint64_t>), g, b, 0, s, A);
        else if (stats)
            hipLaunchKernelGGL((k_direct_2x64<true>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_direct_2x64<false>), g, b, 0, s, A);
        return true;
    } else if (kind == 3) { // Gpu4x32
        if (wide)
            hipLaunchKernelGGL((k_direct_4x32<false, uint64_t>), g, b, 0, s, A);
        else if (stats)
            hipLaunchKernelGGL((k_direct_4x32<true>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_direct_4x32<false>), g, b, 0, s, A);
        return true;
    } else { // Gpu4x64
        if (wide)
            hipLaunchKernelGGL((k_direct_4x64<false, uint64_t>), g, b, 0, s, A);
        else if (stats)
            hipLaunchKernelGGL((k_direct_4x64<true>), g, b, 0, s, A);
        else
            hipLaunchKernelGGL((k_direct_4x64<false>), g, b, 0, s, A);
        return true;
    }
#undef FS_LP_IP
}
