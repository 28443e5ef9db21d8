// oracle/gpu_ref_plain.cpp -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).
//
// CPU restatement of the reference's non-HDR LAv2 kernels, i.e. of
//   mandel_1xHDR_float_perturb_lav2<uint32_t, T, T, Mode, Disable>          FractalSharkGpuLib/LAKernel.cuh:3-315
// for T = float (RenderAlgorithm Gpu1x32PerturbedLAv2[PO|LAO], GPU_Render.cu:1025-1043), double
// (Gpu1x64PerturbedLAv2*, :1077-1100) and CudaDblflt<dblflt> (Gpu2x32PerturbedLAv2*, :1044-1076).  Fractal's AUTO mode
// picks the float kernel for zoom factors 1e4 .. 1e34 (Fractal.cpp:958-966).  Used as the parity checker for the HIP
// kernels in fractalshark_amd/csrc/kernels_plain.hip; nothing in the product path may link, import or call this file.
//
// PARITY UNPINNED: no CPU RenderAlgorithm runs LAv2 on a plain T (the CPU LAv2 function is HDR-only,
// Fractal.cpp:2485-2691), TestRenderGoldens.cpp holds no golden for these algorithms and nvcc is not in this image, so
// this file cannot be checked against an execution of the reference.  What is pinned around it:
//   * the inputs: the plain-double LA table equals the golden-pinned HDRFloat<double> table value for value
//     (tests/test_plain_oracle.py), and the plain builder is the same code instantiated for float;
//   * the double-float primitives (df_ref.hpp) are cross-checked against exact rational arithmetic;
//   * the double kernel's iteration counts are compared with the pinned HDRFloat<double> oracle on the same view.
// Arithmetic follows the *source* semantics: un-annotated float / double expressions are evaluated operation by
// operation in source order, each one correctly rounded, no contraction, denormals kept (nvcc's Debug configuration;
// the Release configuration's --use_fast_math is a compiler option, not part of the algorithm).
//
// What is restated (reference file:line):
//   FloatComplex<T>          HpSharkFloatLib/FloatComplex.h:188-211 (+, *), 245-268 (* real, + real), 327-331
//                            (norm_squared), 413-419 (chebychevNorm)
//   ATInfo (plain arms)      HpSharkFloatLib/ATInfo.h:126-188
//   GPU_LAReference / GPU_LAInfoDeep / GPU_LAstep (plain arms)
//                            FractalSharkLib/GPU_LAReference.h:238-303, GPU_LAInfoDeep.h:90-129, LAstep.h:163-185
//   bailout / compares       HpSharkFloatLib/HDRFloat.h:1536-1586 (`one < two`, `one < T(256)`)
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <thread>
#include <vector>

#include "../include/fs_layout.h"
#include "df_ref.hpp"

extern "C" uint32_t orc_get_row_step(void); // cpu_ref.cpp

namespace {

// ---- the three number types behind one set of function names
inline float Add(float a, float b) { return a + b; }
inline float Sub(float a, float b) { return a - b; }
inline float Mul(float a, float b) { return a * b; }
inline float Neg(float a) { return -a; }
inline float Abs(float a) { return std::fabs(a); } // HdrAbs, HDRFloat.h:1385-1404
inline bool Lt(float a, float b) { return a < b; }
inline bool Gt(float a, float b) { return a > b; }
inline bool Ge(float a, float b) { return a >= b; }
inline bool Le(float a, float b) { return a <= b; }
inline double Add(double a, double b) { return a + b; }
inline double Sub(double a, double b) { return a - b; }
inline double Mul(double a, double b) { return a * b; }
inline double Neg(double a) { return -a; }
inline double Abs(double a) { return std::fabs(a); }
inline bool Lt(double a, double b) { return a < b; }
inline bool Gt(double a, double b) { return a > b; }
inline bool Ge(double a, double b) { return a >= b; }
inline bool Le(double a, double b) { return a <= b; }
inline DF Add(DF a, DF b) { return DFAdd(a, b); }
inline DF Sub(DF a, DF b) { return DFSub(a, b); }
inline DF Mul(DF a, DF b) { return DFMul(a, b); }
inline DF Neg(DF a) { return DFNeg(a); }
inline DF Abs(DF a) { return DFAbs(a); }
inline bool Lt(DF a, DF b) { return DFLt(a, b); }
inline bool Gt(DF a, DF b) { return DFGt(a, b); }
inline bool Ge(DF a, DF b) { return DFGe(a, b); }
// CudaDblflt's operator<= is written `!(b > a)` (CudaDblflt.h:218-222), i.e. it answers b <= a.  Restated as written:
// the AT validity test of the 2x32 kernel (ATInfo.h:131, `cheb(dc) <= ThresholdC`) therefore passes when cheb(dc) >=
// ThresholdC.
inline bool Le(DF a, DF b) { return !DFGt(b, a); }

template <class T> struct Num;
template <> struct Num<float> {
    static float FromInt(int v) { return (float)v; }
};
template <> struct Num<double> {
    static double FromInt(int v) { return (double)v; }
};
template <> struct Num<DF> {
    static DF FromInt(int v) { return DFFromFloat((float)v); } // device: only CudaDblflt(float) is viable, CudaDblflt.h:52-68
};

template <class T> struct Cx {
    T re, im;
};
template <class T> Cx<T> CAdd(Cx<T> a, Cx<T> b) { return Cx<T>{Add(a.re, b.re), Add(a.im, b.im)}; }
template <class T> Cx<T> CMul(Cx<T> a, Cx<T> b)
{
    const T re = Sub(Mul(a.re, b.re), Mul(a.im, b.im));
    const T im = Add(Mul(a.re, b.im), Mul(a.im, b.re));
    return Cx<T>{re, im};
}
template <class T> Cx<T> CMulReal(Cx<T> a, T f) { return Cx<T>{Mul(a.re, f), Mul(a.im, f)}; }
template <class T> T CNormSq(Cx<T> a) { return Add(Mul(a.re, a.re), Mul(a.im, a.im)); }
template <class T> T CCheb(Cx<T> a)
{
    const T ar = Abs(a.re), ai = Abs(a.im);
    return Gt(ar, ai) ? ar : ai;
}

// record field readers
inline float RealOf(float v) { return v; }
inline double RealOf(double v) { return v; }
inline DF RealOf(const fs_real_p2x32 &v) { return DF{v.head, v.tail}; }
inline Cx<float> CplxOf(const fs_cplx_f32 &c) { return Cx<float>{c.re, c.im}; }
inline Cx<double> CplxOf(const fs_cplx_f64 &c) { return Cx<double>{c.re, c.im}; }
inline Cx<DF> CplxOf(const fs_cplx_p2x32 &c) { return Cx<DF>{DF{c.re_head, c.re_tail}, DF{c.im_head, c.im_tail}}; }
inline float OrbX(const fs_orbit_f32 &o) { return o.x; }
inline float OrbY(const fs_orbit_f32 &o) { return o.y; }
inline double OrbX(const fs_orbit_f64 &o) { return o.x; }
inline double OrbY(const fs_orbit_f64 &o) { return o.y; }
inline DF OrbX(const fs_orbit_p2x32 &o) { return DF{o.x_head, o.x_tail}; }
inline DF OrbY(const fs_orbit_p2x32 &o) { return DF{o.y_head, o.y_tail}; }

template <class RowFn> void run_rows(uint32_t y0, uint32_t y1, int threads, RowFn fn)
{
    std::deque<std::atomic_uint64_t> atomics;
    atomics.resize(y1);
    const uint32_t step = orc_get_row_step() ? orc_get_row_step() : 1;
    auto one_thread = [&]() {
        for (size_t y = y0; y < y1; y += step) {
            uint64_t expected = 0;
            if (atomics[y] != 0 || !atomics[y].compare_exchange_strong(expected, 1llu))
                continue;
            fn((uint32_t)y);
        }
    };
    if (threads <= 1) {
        one_thread();
        return;
    }
    std::vector<std::unique_ptr<std::thread>> pool;
    for (int t = 0; t < threads; t++)
        pool.push_back(std::make_unique<std::thread>(one_thread));
    for (auto &t : pool)
        t->join();
}

// mode: 0 = Full, 1 = PO, 2 = LAO.  stats (optional) = {AT iterations, LA steps, perturbation steps}.
template <class T, class OrbitRec, class LaRec, class AtRec, class RealRec>
void lav2_plain(uint32_t *out, uint32_t pitch, uint32_t width, uint32_t y0, uint32_t y1, const OrbitRec *orbit,
                uint32_t orbit_count, const LaRec *las, const fs_la_stage_u32 *stages, uint32_t stage_count,
                int la_valid, int use_at, const AtRec *at, const RealRec coords[4], uint32_t n_iterations, int mode,
                int threads, uint64_t *stats)
{
    const T dx = RealOf(coords[0]), dy = RealOf(coords[1]), centerX = RealOf(coords[2]), centerY = RealOf(coords[3]);
    std::atomic<uint64_t> s_at{0}, s_la{0}, s_pt{0};
    const T Two = Num<T>::FromInt(2), TwoFiftySix = Num<T>::FromInt(256);

    run_rows(y0, y1, threads, [&](uint32_t Y) {
        uint64_t c_at = 0, c_la = 0, c_pt = 0;
        for (uint32_t X = 0; X < width; X++) {
            // LAKernel.cuh:39-63
            uint32_t iter = 0, RefIteration = 0;
            const T DeltaReal = Sub(Mul(dx, Num<T>::FromInt((int)X)), centerX);
            const T DeltaImaginary = Sub(Mul(Neg(dy), Num<T>::FromInt((int)Y)), centerY);
            const T DeltaSub0X = DeltaReal, DeltaSub0Y = DeltaImaginary;
            const Cx<T> DeltaSub0{DeltaReal, DeltaImaginary};
            Cx<T> DeltaSubN{Num<T>::FromInt(0), Num<T>::FromInt(0)};

            if (mode == 0 || mode == 2) {
                // :66-71, ATInfo::isValid / PerformAT (plain arms), ATInfo.h:126-188
                if (la_valid && use_at && Le(CCheb(DeltaSub0), RealOf(at->ThresholdC))) {
                    const uint32_t ATMaxIt = n_iterations / at->StepLength;
                    const Cx<T> c = CAdd(CMul(DeltaSub0, CplxOf(at->CCoeff)), CplxOf(at->RefC));
                    Cx<T> z{Num<T>::FromInt(0), Num<T>::FromInt(0)}; // FloatComplex{}: SubType{} == 0
                    const T SqrEscapeRadius = RealOf(at->SqrEscapeRadius);
                    uint32_t i;
                    for (i = 0; i < ATMaxIt; i++) {
                        const T nsq = CNormSq(z);
                        if (Gt(nsq, SqrEscapeRadius))
                            break;
                        z = CAdd(CMul(z, z), c);
                    }
                    c_at += i;
                    DeltaSubN = CMul(z, CplxOf(at->InvZCoeff));
                    iter = i * at->StepLength;
                }
                // :73-131 (complex0's pre-loop value is dead: it is overwritten before its first use)
                uint32_t CurrentLAStage = la_valid ? stage_count : 0;
                while (CurrentLAStage > 0) {
                    CurrentLAStage--;
                    const uint32_t LAIndex = stages[CurrentLAStage].LAIndex;
                    // GPU_LAReference::isLAStageInvalid, GPU_LAReference.h:238-254 (`temp3 >= temp2`)
                    if (Ge(CCheb(DeltaSub0), RealOf(las[LAIndex].LAThresholdC)))
                        continue;
                    const uint32_t MacroItCount = stages[CurrentLAStage].MacroItCount;
                    uint32_t j = RefIteration;
                    while (iter < n_iterations) {
                        // GPU_LAReference::getLA, GPU_LAReference.h:271-303
                        const LaRec &LAj = las[LAIndex + j];
                        const uint32_t l = LAj.StepLength;
                        bool unusable = true;
                        Cx<T> newdz{};
                        if (iter + l <= n_iterations) {
                            // GPU_LAInfoDeep::Prepare, GPU_LAInfoDeep.h:90-106
                            newdz = CMul(DeltaSubN, CAdd(CMulReal(CplxOf(LAj.Ref), Two), DeltaSubN));
                            unusable = Ge(CCheb(newdz), RealOf(LAj.LAThreshold));
                        }
                        if (unusable) {
                            RefIteration = LAj.NextStageLAIndex;
                            break;
                        }
                        iter += l;
                        c_la++;
                        // Evaluate GPU_LAInfoDeep.h:120-124; getZ LAstep.h:181-185
                        DeltaSubN = CAdd(CMul(newdz, CplxOf(LAj.ZCoeff)), CMul(DeltaSub0, CplxOf(LAj.CCoeff)));
                        const Cx<T> complex0 = CAdd(CplxOf(las[LAIndex + j + 1].Ref), DeltaSubN);
                        j++;
                        if (Lt(CCheb(complex0), CCheb(DeltaSubN)) || j >= MacroItCount) {
                            DeltaSubN = complex0;
                            j = 0;
                        }
                    }
                    if (iter >= n_iterations)
                        break;
                }
            }

            if (mode == 0 || mode == 1) {
                // :133-235; perturbLoop(maxRefIteration) at :254-276 reads the block's previous results, which are zero
                // on the cleared buffer every caller passes (Fractal.cpp:2822), so only perturbLoop(n_iterations) runs.
                T dX = DeltaSubN.re, dY = DeltaSubN.im;
                T zx = OrbX(orbit[RefIteration]), zy = OrbY(orbit[RefIteration]);
                for (;;) {
                    const T dXo = dX, dYo = dY;
                    const T tempMulX2 = Mul(zx, Two), tempMulY2 = Mul(zy, Two);
                    ++RefIteration;
                    const T tempSum1 = Add(tempMulY2, dYo), tempSum2 = Add(tempMulX2, dXo);
                    dX = Add(Sub(Mul(dXo, tempSum2), Mul(dYo, tempSum1)), DeltaSub0X);
                    dY = Add(Add(Mul(dXo, tempSum1), Mul(dYo, tempSum2)), DeltaSub0Y);
                    c_pt++;
                    zx = OrbX(orbit[RefIteration]);
                    zy = OrbY(orbit[RefIteration]);
                    const T tempZX = Add(zx, dX), tempZY = Add(zy, dY);
                    const T normSquared = Add(Mul(tempZX, tempZX), Mul(tempZY, tempZY));
                    if (Lt(normSquared, TwoFiftySix) && iter < n_iterations) {
                        const T DeltaNormSquared = Add(Mul(dX, dX), Mul(dY, dY));
                        if (Lt(normSquared, DeltaNormSquared) || RefIteration >= orbit_count - 1) {
                            dX = tempZX;
                            dY = tempZY;
                            RefIteration = 0;
                            zx = OrbX(orbit[0]);
                            zy = OrbY(orbit[0]);
                        }
                        ++iter;
                    } else {
                        break;
                    }
                }
            }
            out[(size_t)Y * pitch + X] = iter;
        }
        s_at += c_at;
        s_la += c_la;
        s_pt += c_pt;
    });
    if (stats) {
        stats[0] = s_at;
        stats[1] = s_la;
        stats[2] = s_pt;
    }
}

} // namespace

extern "C" {

// kind: 0 = float (fs_orbit_f32 / fs_la_f32_u32 / fs_at_f32_u32 / float[4]), 1 = double (fs_orbit_f64 / fs_la_f64_u32 /
// fs_at_f64_u32 / double[4]), 2 = CudaDblflt (fs_orbit_p2x32 / fs_la_p2x32_u32 / fs_at_p2x32_u32 / fs_real_p2x32[4]).
void orc_gpu_lav2_plain(int kind, uint32_t *out, uint32_t pitch, uint32_t width, uint32_t y0, uint32_t y1,
                        const void *orbit, uint32_t orbit_count, const void *las, const fs_la_stage_u32 *stages,
                        uint32_t stage_count, int la_valid, int use_at, const void *at, const void *coords,
                        uint32_t n_iterations, int mode, int threads, uint64_t *stats)
{
    if (kind == 0)
        lav2_plain<float>(out, pitch, width, y0, y1, (const fs_orbit_f32 *)orbit, orbit_count, (const fs_la_f32_u32 *)las,
                          stages, stage_count, la_valid, use_at, (const fs_at_f32_u32 *)at, (const float *)coords,
                          n_iterations, mode, threads, stats);
    else if (kind == 1)
        lav2_plain<double>(out, pitch, width, y0, y1, (const fs_orbit_f64 *)orbit, orbit_count,
                           (const fs_la_f64_u32 *)las, stages, stage_count, la_valid, use_at, (const fs_at_f64_u32 *)at,
                           (const double *)coords, n_iterations, mode, threads, stats);
    else
        lav2_plain<DF>(out, pitch, width, y0, y1, (const fs_orbit_p2x32 *)orbit, orbit_count,
                       (const fs_la_p2x32_u32 *)las, stages, stage_count, la_valid, use_at, (const fs_at_p2x32_u32 *)at,
                       (const fs_real_p2x32 *)coords, n_iterations, mode, threads, stats);
}

} // extern "C"
