// oracle/gpu_ref_2x32.cpp -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).
//
// CPU restatement of the reference's 2x32 ("float-float + int32 exponent") LAv2 path, i.e. of the CUDA kernel
//   mandel_1xHDR_float_perturb_lav2<uint32_t, HDRFloat<CudaDblflt<dblflt>>, CudaDblflt<dblflt>, Mode, Disable>
//                                                                          FractalSharkGpuLib/LAKernel.cuh:3-315
// which backs RenderAlgorithm GpuHDRx2x32PerturbedLAv2[PO|LAO] (GPU_Render.cu:1152-1185).  Used as the parity
// checker for the HIP 2x32 kernel (tests/, bench.py's cpu_baseline leg).  Nothing in the product path may link,
// import or call this file.
//
// PARITY UNPINNED: the reference has *no* CPU implementation of 2x32 arithmetic (CudaDblflt's operators exist only
// under __CUDACC__, CudaDblflt.h:149-281; dblflt.cuh is all __device__), no CPU RenderAlgorithm uses the type and
// TestRenderGoldens.cpp holds no golden for it, and nvcc is not in this image.  So this file cannot be checked
// against an execution of the reference.  What it can be (and is) checked against:
//   * the host-side half -- the HDRFloat<double> orbit / LA table it starts from -- is pinned by the golden CRCs of
//     the Cpu64* algorithms (tests/test_oracle_pins.py); the double -> 2x32 conversion restated in
//     fractalshark_amd/host/refinputs.cpp is plain host C++ in the reference (dblflt.h:30-52);
//   * the double-float primitives are cross-checked against exact rational arithmetic (tests/test_2x32_oracle.py);
//   * the rendered iteration counts are compared with the pinned HDRFloat<double> oracle on the same view
//     (same algorithm, 48 vs 53 mantissa bits: equal except at chaotic pixels).
// Arithmetic follows the *source* semantics: every __fadd_rn / __fmul_rn / __fmaf_rn is one correctly rounded IEEE
// binary32 operation with denormals kept (nvcc's Debug configuration, FastMath=false,
// FractalShark.CudaDefaults.props:38).  The Release configuration adds --use_fast_math (FTZ, approximate division;
// :75), which is a compiler option, not part of the algorithm.
//
// What is restated (reference file:line):
//   dblflt primitives        HpSharkFloatLib/dblflt.cuh:86-215 (add_float_to_dblflt, add/sub/mul_dblflt), dblflt.h:19-29
//   CudaDblflt compare/abs   HpSharkFloatLib/CudaDblflt.h:197-259
//   HDRFloat<CudaDblflt>     HpSharkFloatLib/HDRFloat.h:293-363 (ctor), 458-488 (Reduce), 497-551 (getMultiplier[Neg]),
//                            808-812 (custom_perturb3), 829-840 (multiply), 877-884 (square), 974-1000 / 1039-1065
//                            (add / subtract), 1150-1184 (compares), 1385-1404 (HdrAbs)
//   HDRFloatComplex<CudaDblflt>  HpSharkFloatLib/HDRFloatComplex.h:159-171 (setMantexp), 219-247 (plus), 270-283
//                            (times), 333-347 (times real), 472-500 (Reduce, CudaDblflt branch), 516-519 (norm_squared),
//                            692-695 (chebychevNorm)
//   ATInfo                   HpSharkFloatLib/ATInfo.h:126-188
//   GPU_LAReference / GPU_LAInfoDeep / GPU_LAstep
//                            FractalSharkLib/GPU_LAReference.h:238-303, GPU_LAInfoDeep.h:90-129, LAstep.h:163-185
//
// Build: part of liboracle.so (oracle/Makefile), g++ -O3 -ffp-contract=off; fmaf() is libm's correctly rounded FMA.
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

constexpr int32_t MINEXP = INT32_MIN >> 3; // HDRFloat.h:50-58
constexpr int32_t DIFF_IGNORED = 120;      // HDRFloat.h:122

// ------------------------------------------------------------------ HDRFloat<CudaDblflt>
struct H2 {
    DF m;
    int32_t e;
};
inline H2 H2Zero() { return H2{DFZero(), MINEXP}; } // HDRFloat.h:200-204

// getMultiplier, HDRFloat.h:497-521 (float / CudaDblflt branch).  numeric_limits<CudaDblflt>::max() is the
// unspecialised primary template = CudaDblflt{}; only reachable for scale >= 128, which the path never produces.
inline DF GetMultiplier(int32_t s)
{
    if (s <= -127)
        return DFZero();
    if (s >= 128)
        return DFZero();
    return DFFromFloat(scalbnf(1.0f, s));
}
// getMultiplierNeg, HDRFloat.h:523-551
inline DF GetMultiplierNeg(int32_t s)
{
    if (s <= -127)
        return DFZero();
    return DFFromFloat(scalbnf(1.0f, s));
}

// HDRFloat(const U number) for U = int / float, T = CudaDblflt: HDRFloat.h:293-363.
inline H2 H2FromFloat(float f)
{
    if (f == 0.0f)
        return H2{DFFromFloat(0.0f), MINEXP};
    const uint32_t bits = f2u(f);
    const int32_t f_exp = (int32_t)((bits & 0x7F800000u) >> 23) - 127;
    H2 r;
    r.m.head = u2f((bits & 0x807FFFFFu) | 0x3F800000u);
    r.m.tail = 0;
    r.e = f_exp;
    return r;
}
inline H2 H2FromInt(int v) { return v == 0 ? H2{DFFromFloat(0.0f), MINEXP} : H2FromFloat((float)v); }

// Reduce(), HDRFloat.h:458-488
inline void H2Reduce(H2 &a)
{
    if (a.m.head == 0 && a.m.tail == 0)
        return;
    const uint32_t bits_y = f2u(a.m.head);
    const uint32_t bits_x = f2u(a.m.tail);
    const int32_t f_exp_y = (int32_t)((bits_y & 0x7F800000u) >> 23) - 127;
    const int32_t f_exp_x = (int32_t)((bits_x & 0x7F800000u) >> 23);
    const uint32_t val_y = (bits_y & 0x807FFFFFu) | 0x3F800000u;
    const int32_t newexp = f_exp_x - f_exp_y;
    const int32_t satexp = newexp <= 0 ? 0 : newexp;
    const uint32_t val_x = (bits_x & 0x807FFFFFu) | ((uint32_t)satexp << 23);
    a.e += f_exp_y;
    a.m.head = u2f(val_y);
    a.m.tail = u2f(val_x);
}
inline H2 H2Reduced(H2 a)
{
    H2Reduce(a);
    return a;
}
// HDRFloat(T mant): mantissa = mant, exp = 0, HdrReduce; HDRFloat.h:206-212
inline H2 H2FromMant(DF m) { return H2Reduced(H2{m, 0}); }

inline int32_t ClampExp(int32_t e) { return e < MINEXP ? MINEXP : e; }
// multiply_mutable, HDRFloat.h:829-840
inline H2 H2Mul(H2 a, H2 b) { return H2{DFMul(a.m, b.m), ClampExp(a.e + b.e)}; }
// square(), HDRFloat.h:877-884 (operator*, not CudaDblflt::square(); exponent not clamped)
inline H2 H2Square(H2 a) { return H2{DFMul(a.m, a.m), a.e * 2}; }
// negate(), HDRFloat.h:1129-1133
inline H2 H2Neg(H2 a) { return H2{DFNeg(a.m), a.e}; }
// add_mutable, HDRFloat.h:974-1000
inline H2 H2Add(H2 a, H2 v)
{
    const int32_t d = a.e - v.e;
    if (d >= DIFF_IGNORED)
        return a;
    if (d >= 0) {
        const DF mul = GetMultiplierNeg(-d);
        a.m = DFAdd(a.m, DFMul(v.m, mul));
    } else if (d > -DIFF_IGNORED) {
        const DF mul = GetMultiplierNeg(d);
        a.e = v.e;
        a.m = DFAdd(DFMul(a.m, mul), v.m);
    } else {
        a.e = v.e;
        a.m = v.m;
    }
    if (DFEq(a.m, DFZero()))
        a.e = MINEXP;
    return a;
}
// subtract_mutable, HDRFloat.h:1039-1065
inline H2 H2Sub(H2 a, H2 v)
{
    const int32_t d = a.e - v.e;
    if (d >= DIFF_IGNORED)
        return a;
    if (d >= 0) {
        const DF mul = GetMultiplierNeg(-d);
        a.m = DFSub(a.m, DFMul(v.m, mul));
    } else if (d > -DIFF_IGNORED) {
        const DF mul = GetMultiplierNeg(d);
        a.e = v.e;
        a.m = DFSub(DFMul(a.m, mul), v.m);
    } else {
        a.e = v.e;
        a.m = DFNeg(v.m);
    }
    if (DFEq(a.m, DFZero()))
        a.e = MINEXP;
    return a;
}
// compareToBothPositiveReduced, HDRFloat.h:1150-1167
inline int H2CmpPos(H2 a, H2 b)
{
    if (a.e > b.e)
        return 1;
    if (a.e < b.e)
        return -1;
    if (DFGt(a.m, b.m))
        return 1;
    if (DFLt(a.m, b.m))
        return -1;
    return 0;
}
// compareToBothPositiveReducedTemplate<256>, HDRFloat.h:1169-1184
inline int H2CmpTemplate256(H2 a)
{
    if (a.e > 1)
        return 1;
    if (a.e < 1)
        return -1;
    return DFGe(a.m, DFFromFloat(256.0f)) ? 1 : -1;
}
// HdrAbs, HDRFloat.h:1385-1404
inline H2 H2Abs(H2 a) { return H2{DFAbs(a.m), a.e}; }

// ------------------------------------------------------------------ HDRFloatComplex<CudaDblflt>
struct C2 {
    DF re;
    DF im;
    int32_t e;
};
inline C2 C2Zero() { return C2{DFZero(), DFZero(), MINEXP}; } // HDRFloatComplex.h:127-132
// setMantexp, HDRFloatComplex.h:159-171
inline C2 C2FromH(H2 re, H2 im)
{
    C2 c;
    c.e = re.e > im.e ? re.e : im.e;
    c.re = DFMul(re.m, GetMultiplier(re.e - c.e));
    c.im = DFMul(im.m, GetMultiplier(im.e - c.e));
    return c;
}
inline H2 C2Re(C2 a) { return H2{a.re, a.e}; } // getRe, HDRFloatComplex.h:637-641
inline H2 C2Im(C2 a) { return H2{a.im, a.e}; }
// plus_mutable(HDRFloatComplex), HDRFloatComplex.h:219-247
inline C2 C2Add(C2 a, C2 v)
{
    const int32_t d = a.e - v.e;
    if (d >= DIFF_IGNORED)
        return a;
    if (d >= 0) {
        const DF mul = GetMultiplier(-d);
        a.re = DFAdd(a.re, DFMul(v.re, mul));
        a.im = DFAdd(a.im, DFMul(v.im, mul));
    } else if (d > -DIFF_IGNORED) {
        const DF mul = GetMultiplier(d);
        a.e = v.e;
        a.re = DFAdd(DFMul(a.re, mul), v.re);
        a.im = DFAdd(DFMul(a.im, mul), v.im);
    } else {
        a = v;
    }
    return a;
}
// times_mutable(HDRFloatComplex), HDRFloatComplex.h:270-283
inline C2 C2Mul(C2 a, C2 f)
{
    C2 r;
    r.re = DFSub(DFMul(a.re, f.re), DFMul(a.im, f.im));
    r.im = DFAdd(DFMul(a.re, f.im), DFMul(a.im, f.re));
    r.e = ClampExp(a.e + f.e);
    return r;
}
// times_mutable(HDRFloat), HDRFloatComplex.h:333-347
inline C2 C2MulReal(C2 a, H2 f) { return C2{DFMul(a.re, f.m), DFMul(a.im, f.m), ClampExp(a.e + f.e)}; }
// Reduce(), CudaDblflt branch, HDRFloatComplex.h:472-500: two HDRFloat(T mant) (each reduces) + Reduce again
// (idempotent), setMantexp, exp += old exp.
inline void C2Reduce(C2 &a)
{
    if (DFEq(a.re, DFZero()) && DFEq(a.im, DFZero()))
        return;
    H2 tr = H2FromMant(a.re);
    H2 ti = H2FromMant(a.im);
    H2Reduce(tr);
    H2Reduce(ti);
    const int32_t old = a.e;
    a = C2FromH(tr, ti);
    a.e += old;
}
// norm_squared(), HDRFloatComplex.h:516-519
inline H2 C2NormSq(C2 a) { return H2{DFAdd(DFMul(a.re, a.re), DFMul(a.im, a.im)), a.e << 1}; }
// chebychevNorm(), HDRFloatComplex.h:692-695 (maxBothPositiveReduced: a > b ? a : b)
inline H2 C2Cheb(C2 a)
{
    const H2 x = H2Abs(C2Re(a)), y = H2Abs(C2Im(a));
    return H2CmpPos(x, y) > 0 ? x : y;
}

inline H2 RealOf(const fs_real_2x32 &r) { return H2{DF{r.head, r.tail}, r.e}; }
inline C2 CplxOf(const fs_cplx_2x32 &c) { return C2{DF{c.re_head, c.re_tail}, DF{c.im_head, c.im_tail}, c.e}; }

template <class RowFn> void run_rows(uint32_t y0, uint32_t y1, int threads, RowFn fn)
{
    std::deque<std::atomic_uint64_t> atomics;
    atomics.resize(y1);
    const uint32_t step = orc_get_row_step() ? orc_get_row_step() : 1;
    auto one_thread = [&]() {
        for (size_t y = y0; y < y1; y += step) {
            if (atomics[y] != 0)
                continue;
            uint64_t expected = 0;
            if (atomics[y].compare_exchange_strong(expected, 1llu) == false)
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

} // namespace

extern "C" {

// Primitive hooks for tests/test_2x32_oracle.py (exact-arithmetic cross-check of the double-float operations).
void orc_df_add(const float a[2], const float b[2], float out[2])
{
    const DF r = DFAdd(DF{a[0], a[1]}, DF{b[0], b[1]});
    out[0] = r.head;
    out[1] = r.tail;
}
void orc_df_sub(const float a[2], const float b[2], float out[2])
{
    const DF r = DFSub(DF{a[0], a[1]}, DF{b[0], b[1]});
    out[0] = r.head;
    out[1] = r.tail;
}
void orc_df_mul(const float a[2], const float b[2], float out[2])
{
    const DF r = DFMul(DF{a[0], a[1]}, DF{b[0], b[1]});
    out[0] = r.head;
    out[1] = r.tail;
}
void orc_h2_reduce(fs_real_2x32 *v)
{
    H2 h = RealOf(*v);
    H2Reduce(h);
    *v = fs_real_2x32{h.m.head, h.m.tail, h.e};
}

void orc_h2_add(const fs_real_2x32 *a, const fs_real_2x32 *b, int subtract, fs_real_2x32 *out)
{
    const H2 r = subtract ? H2Sub(RealOf(*a), RealOf(*b)) : H2Add(RealOf(*a), RealOf(*b));
    *out = fs_real_2x32{r.m.head, r.m.tail, r.e};
}
void orc_c2_reduce(fs_cplx_2x32 *v)
{
    C2 c = CplxOf(*v);
    C2Reduce(c);
    *v = fs_cplx_2x32{c.re.head, c.re.tail, c.im.head, c.im.tail, c.e};
}

// mode: 0 = Full, 1 = PO, 2 = LAO (RenderAlgorithm.h:12-17).  stats (optional, [3]): AT iterations, LA steps,
// perturbation steps summed over the rendered rows.
void orc_gpu_lav2_2x32(uint32_t *out, uint32_t pitch, uint32_t width, uint32_t y0, uint32_t y1,
                       const fs_orbit_2x32 *orbit, uint32_t orbit_count, const fs_la_2x32_u32 *las,
                       const fs_la_stage_u32 *stages, uint32_t stage_count, int la_valid, int use_at,
                       const fs_at_2x32_u32 *at, const fs_real_2x32 coords[4], uint32_t n_iterations, int mode,
                       int threads, uint64_t *stats)
{
    const H2 dx = RealOf(coords[0]), dy = RealOf(coords[1]), centerX = RealOf(coords[2]), centerY = RealOf(coords[3]);
    std::atomic<uint64_t> s_at{0}, s_la{0}, s_pt{0};
    auto OrbX = [&](uint32_t i) { return H2{DF{orbit[i].x_head, orbit[i].x_tail}, orbit[i].ex}; };
    auto OrbY = [&](uint32_t i) { return H2{DF{orbit[i].y_head, orbit[i].y_tail}, orbit[i].ey}; };

    run_rows(y0, y1, threads, [&](uint32_t Y) {
        uint64_t c_at = 0, c_la = 0, c_pt = 0;
        for (uint32_t X = 0; X < width; X++) {
            // LAKernel.cuh:39-63
            uint32_t iter = 0, RefIteration = 0;
            const H2 DeltaReal = H2Sub(H2Mul(dx, H2FromInt((int)X)), centerX);
            const H2 DeltaImaginary = H2Sub(H2Mul(H2Neg(dy), H2FromInt((int)Y)), centerY);
            const H2 DeltaSub0X = DeltaReal, DeltaSub0Y = DeltaImaginary;
            const C2 DeltaSub0 = C2FromH(DeltaReal, DeltaImaginary);
            C2 DeltaSubN = C2FromH(H2FromInt(0), H2FromInt(0));

            if (mode == 0 || mode == 2) {
                // :66-71, ATInfo::isValid / PerformAT, ATInfo.h:126-188
                if (la_valid && use_at && H2CmpPos(C2Cheb(DeltaSub0), RealOf(at->ThresholdC)) <= 0) {
                    const uint32_t ATMaxIt = n_iterations / at->StepLength;
                    C2 c = C2Add(C2Mul(DeltaSub0, CplxOf(at->CCoeff)), CplxOf(at->RefC));
                    C2Reduce(c);
                    C2 z = C2Zero();
                    const H2 SqrEscapeRadius = RealOf(at->SqrEscapeRadius);
                    uint32_t i;
                    for (i = 0; i < ATMaxIt; i++) {
                        H2 nsq = C2NormSq(z);
                        H2Reduce(nsq);
                        if (H2CmpPos(nsq, SqrEscapeRadius) > 0)
                            break;
                        z = C2Add(C2Mul(z, z), c);
                    }
                    c_at += i;
                    C2 dz = C2Mul(z, CplxOf(at->InvZCoeff));
                    C2Reduce(dz);
                    iter = i * at->StepLength;
                    DeltaSubN = dz;
                }
                // :73-131 (complex0's pre-loop value is dead: it is overwritten before its first use)
                uint32_t CurrentLAStage = la_valid ? stage_count : 0;
                while (CurrentLAStage > 0) {
                    CurrentLAStage--;
                    const uint32_t LAIndex = stages[CurrentLAStage].LAIndex;
                    // GPU_LAReference::isLAStageInvalid, GPU_LAReference.h:238-254
                    if (H2CmpPos(C2Cheb(DeltaSub0), RealOf(las[LAIndex].LAThresholdC)) >= 0)
                        continue;
                    const uint32_t MacroItCount = stages[CurrentLAStage].MacroItCount;
                    uint32_t j = RefIteration;
                    while (iter < n_iterations) {
                        // GPU_LAReference::getLA, GPU_LAReference.h:271-303
                        const fs_la_2x32_u32 &LAj = las[LAIndex + j];
                        const uint32_t l = LAj.StepLength;
                        bool unusable = true;
                        C2 newdz = C2Zero();
                        if (iter + l <= n_iterations) {
                            // GPU_LAInfoDeep::Prepare, GPU_LAInfoDeep.h:90-106
                            newdz = C2Mul(DeltaSubN, C2Add(C2MulReal(CplxOf(LAj.Ref), H2FromInt(2)), DeltaSubN));
                            C2Reduce(newdz);
                            unusable = H2CmpPos(C2Cheb(newdz), RealOf(LAj.LAThreshold)) >= 0;
                        }
                        if (unusable) {
                            RefIteration = LAj.NextStageLAIndex;
                            break;
                        }
                        c_la++;
                        iter += l;
                        // GPU_LAInfoDeep::Evaluate :120-124, GPU_LAstep::getZ LAstep.h:181-185
                        DeltaSubN = C2Add(C2Mul(newdz, CplxOf(LAj.ZCoeff)), C2Mul(DeltaSub0, CplxOf(LAj.CCoeff)));
                        const C2 complex0 = C2Add(CplxOf(las[LAIndex + j + 1].Ref), DeltaSubN);
                        j++;
                        const H2 complex0Norm = H2Reduced(C2Cheb(complex0));
                        const H2 DeltaSubNNorm = H2Reduced(C2Cheb(DeltaSubN));
                        if (H2CmpPos(complex0Norm, DeltaSubNNorm) < 0 || j >= MacroItCount) {
                            DeltaSubN = complex0;
                            j = 0;
                        }
                    }
                    if (iter >= n_iterations)
                        break;
                }
            }

            if (mode == 0 || mode == 1) {
                // :133-235; the first perturbLoop(maxRefIteration) call (:254-276) never runs on a cleared buffer.
                H2 DeltaSubNX = C2Re(DeltaSubN), DeltaSubNY = C2Im(DeltaSubN);
                uint32_t idx = RefIteration;
                H2 zx = OrbX(idx), zy = OrbY(idx);
                for (;;) {
                    const H2 DeltaSubNXOrig = DeltaSubNX, DeltaSubNYOrig = DeltaSubNY;
                    const H2 two = H2FromInt(2);
                    const H2 tempMulX2 = H2Mul(zx, two);
                    const H2 tempMulY2 = H2Mul(zy, two);
                    ++RefIteration;
                    const H2 tempSum1 = H2Add(tempMulY2, DeltaSubNYOrig);
                    const H2 tempSum2 = H2Add(tempMulX2, DeltaSubNXOrig);
                    // custom_perturb3, HDRFloat.h:797-812 (its tempSum1 parameter receives tempSum2 and vice versa)
                    DeltaSubNX = H2Add(H2Sub(H2Mul(DeltaSubNXOrig, tempSum2), H2Mul(DeltaSubNYOrig, tempSum1)), DeltaSub0X);
                    H2Reduce(DeltaSubNX);
                    DeltaSubNY = H2Add(H2Add(H2Mul(DeltaSubNXOrig, tempSum1), H2Mul(DeltaSubNYOrig, tempSum2)), DeltaSub0Y);
                    H2Reduce(DeltaSubNY);
                    c_pt++;

                    idx++;
                    zx = OrbX(idx);
                    zy = OrbY(idx);
                    const H2 tempZX = H2Add(zx, DeltaSubNX);
                    const H2 tempZY = H2Add(zy, DeltaSubNY);
                    const H2 normSquared = H2Reduced(H2Add(H2Square(tempZX), H2Square(tempZY)));
                    if (H2CmpTemplate256(normSquared) < 0 && iter < n_iterations) {
                        const H2 DeltaNormSquared = H2Reduced(H2Add(H2Square(DeltaSubNX), H2Square(DeltaSubNY)));
                        if (H2CmpPos(normSquared, DeltaNormSquared) < 0 || RefIteration >= orbit_count - 1) {
                            DeltaSubNX = tempZX;
                            DeltaSubNY = tempZY;
                            RefIteration = 0;
                            idx = 0;
                            zx = OrbX(0);
                            zy = OrbY(0);
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


// PerturbExtras::SimpleCompression orbits as the reference's *RC* kernels see them, entry by entry:
// GPUPerturbSingleResults::GetCompressedComplexSeq (Perturb.cuh:300-326) walked from index 0 -- the next waypoint when
// its CompressionIndex is reached, otherwise z <- z*z + OrbitLow in the kernel's number type (Type{2} * zx_old * zy).
void orc_decompress_p2x32(const fs_orbit_p2x32_rc *wp, uint64_t n_wp, uint64_t n_uncompressed, const fs_real_p2x32 low[2],
                          fs_orbit_p2x32 *out)
{
    const DF cx{low[0].head, low[0].tail}, cy{low[1].head, low[1].tail};
    const DF two = DFFromFloat(2.0f);
    DF zx = DFZero(), zy = DFZero();
    uint64_t next = 0;
    for (uint64_t i = 0; i < n_uncompressed; i++) {
        if (next < n_wp && (wp[next].index_and_rebase & 0x7FFFFFFFFFFFFFFFull) == i) {
            zx = DF{wp[next].x_head, wp[next].x_tail};
            zy = DF{wp[next].y_head, wp[next].y_tail};
            next++;
        } else {
            const DF zx_old = zx;
            zx = DFAdd(DFSub(DFMul(zx, zx), DFMul(zy, zy)), cx);
            zy = DFAdd(DFMul(DFMul(two, zx_old), zy), cy);
        }
        out[i] = fs_orbit_p2x32{zx.head, zx.tail, zy.head, zy.tail};
    }
}

void orc_decompress_hdr2x32(const fs_orbit_2x32_rc *wp, uint64_t n_wp, uint64_t n_uncompressed, const fs_real_2x32 low[2],
                            fs_orbit_2x32 *out)
{
    const H2 cx = RealOf(low[0]), cy = RealOf(low[1]);
    const H2 two = H2FromInt(2);
    H2 zx = H2Zero(), zy = H2Zero();
    uint64_t next = 0;
    for (uint64_t i = 0; i < n_uncompressed; i++) {
        if (next < n_wp && (wp[next].index_and_rebase & 0x7FFFFFFFFFFFFFFFull) == i) {
            zx = H2{DF{wp[next].x_head, wp[next].x_tail}, wp[next].ex};
            zy = H2{DF{wp[next].y_head, wp[next].y_tail}, wp[next].ey};
            next++;
        } else {
            const H2 zx_old = zx;
            zx = H2Add(H2Sub(H2Mul(zx, zx), H2Mul(zy, zy)), cx);
            H2Reduce(zx);
            zy = H2Add(H2Mul(H2Mul(two, zx_old), zy), cy);
            H2Reduce(zy);
        }
        out[i] = fs_orbit_2x32{zx.m.head, zx.m.tail, zx.e, zy.e, zy.m.head, zy.m.tail};
    }
}

} // extern "C"
