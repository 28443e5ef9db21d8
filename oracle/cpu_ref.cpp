// oracle/cpu_ref.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's per-pixel RenderAlgorithm functions, used as the parity checker for
// the HIP kernels (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing in the product
// path (fractalshark_amd/, libfsmi355.so) may link, import or call this file.
//
// What it restates (each function cites its source):
//   orc_direct_f64        Fractal::CalcCpuHDR<uint32_t,double,double>                 Fractal.cpp:2096-2206
//   orc_bla_hdr32         Fractal::CalcCpuPerturbationFractalBLA<uint32_t,HDRFloat<float>,float>
//                                                                                      Fractal.cpp:2208-2483
//                         (+ BLAS::LookupBackwards BLAS.cpp:256-310, BLA::getValue BLA.cuh:21-38);
//                         with n_levels == 0 the BLA lookup is suppressed = the perturbation-only
//                         single-step branch :2342-2466, the parity target of config C2
//   orc_lav2_hdr32        Fractal::CalcCpuPerturbationFractalLAV2<uint32_t,float,Disable>
//                                                                                      Fractal.cpp:2485-2691
//                         (+ LAReference::getLA/isLAStageInvalid LAReference.cpp:1076-1134,
//                          LAInfoDeep::Prepare/Evaluate LAInfoDeep.h:395-420, LAstep::getZ LAstep.h:116-120,
//                          ATInfo::isValid/getC/getDZ/PerformAT ATInfo.h:126-188)
// Numeric types: a self-contained restatement of HDRFloat<float> (HpSharkFloatLib/HDRFloat.h) and
// HDRFloatComplex<float> (HDRFloatComplex.h), written independently of fractalshark_amd/csrc/hdr_math.hpp
// (this one calls libm's scalbnf exactly like the reference; the product builds powers of two from bits).
//
// Threading mirrors the reference: N std::threads, each claims whole rows through a per-row atomic
// compare-exchange (Fractal.cpp:2122-2146,2240-2264,2523-2543).
//
// Parity pin: reproduces the reference's golden CRC-64s of FractalSharkTest/TestRenderGoldens.cpp:84-97
// for view0/Cpu64, view5/Cpu32PerturbedBLAHDR, view5/Cpu32PerturbedBLAV2HDR when chained with
// oracle/png_pin.cpp (tests/test_golden_crc.py).
//
// Build: g++ -O3 -ffp-contract=off (no -march, no fast-math: reference build_linux.sh:20-23).
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <thread>
#include <vector>

#include "../include/fs_layout.h"

namespace {

constexpr int32_t MINEXP = INT32_MIN >> 3; // HDRFloat.h:50-58
constexpr int32_t DIFF_IGNORED = 120;      // HDRFloat.h:122

struct H {
    float m;
    int32_t e;
};
struct HC {
    float re, im;
    int32_t e;
};

inline uint32_t f2u(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
inline float u2f(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// HDRFloat.h:497-521
inline float getMultiplier(int32_t s)
{
    if (s <= -127)
        return 0.0f;
    if (s >= 128)
        return 3.402823466e+38f;
    return scalbnf(1.0f, s);
}
// HDRFloat.h:523-551
inline float getMultiplierNeg(int32_t s)
{
    if (s <= -127)
        return 0.0f;
    return scalbnf(1.0f, s);
}

// HDRFloat.h:438-457
inline void Reduce(H &a)
{
    if (a.m == 0)
        return;
    const uint32_t bits = f2u(a.m);
    const int32_t f_exp = (int32_t)((bits & 0x7F800000u) >> 23) - 127;
    a.m = u2f((bits & 0x807FFFFFu) | 0x3F800000u);
    a.e += f_exp;
}
inline H Reduced(H a)
{
    Reduce(a);
    return a;
}
// HDRFloat.h:200-204
inline H HZero() { return H{0.0f, MINEXP}; }
// HDRFloat.h:206-212 HDRFloat(T mant)
inline H HFromMant(float v)
{
    H r{v, 0};
    Reduce(r);
    return r;
}
// HDRFloat.h:295-363 HDRFloat(U number), U=int
inline H HFromInt(int n)
{
    if (n == 0)
        return HZero();
    const uint32_t bits = f2u((float)n);
    return H{u2f((bits & 0x807FFFFFu) | 0x3F800000u), (int32_t)((bits & 0x7F800000u) >> 23) - 127};
}
inline int32_t clampE(int32_t e) { return e < MINEXP ? MINEXP : e; }
// HDRFloat.h:829-840
inline H Mul(H a, H b) { return H{a.m * b.m, clampE(a.e + b.e)}; }
// operator*(HDRFloat, const T&) with the literal 2: HDRFloat(2.0f) = {1.0f, 1}
inline H MulBy2(H a) { return Mul(a, HFromMant(2.0f)); }
inline H MulByFloat(H a, float f) { return Mul(a, HFromMant(f)); }
// HDRFloat.h:974-1000
inline H Add(H a, H b)
{
    const int32_t expDiff = a.e - b.e;
    if (expDiff >= DIFF_IGNORED) {
        return a;
    } else if (expDiff >= 0) {
        const float mul = getMultiplierNeg(-expDiff);
        a.m = a.m + b.m * mul;
    } else if (expDiff > -DIFF_IGNORED) {
        const float mul = getMultiplierNeg(expDiff);
        a.e = b.e;
        a.m = a.m * mul + b.m;
    } else {
        a.e = b.e;
        a.m = b.m;
    }
    if (a.m == 0.0f)
        a.e = MINEXP;
    return a;
}
// HDRFloat.h:1039-1065
inline H Sub(H a, H b)
{
    const int32_t expDiff = a.e - b.e;
    if (expDiff >= DIFF_IGNORED) {
        return a;
    } else if (expDiff >= 0) {
        const float mul = getMultiplierNeg(-expDiff);
        a.m = a.m - b.m * mul;
    } else if (expDiff > -DIFF_IGNORED) {
        const float mul = getMultiplierNeg(expDiff);
        a.e = b.e;
        a.m = a.m * mul - b.m;
    } else {
        a.e = b.e;
        a.m = -b.m;
    }
    if (a.m == 0.0f)
        a.e = MINEXP;
    return a;
}
inline H Neg(H a) { return H{-a.m, a.e}; }
// HDRFloat.h:1150-1167
inline int CmpPosReduced(H a, H b)
{
    if (a.e > b.e)
        return 1;
    else if (a.e < b.e)
        return -1;
    else {
        if (a.m > b.m)
            return 1;
        else if (a.m < b.m)
            return -1;
        return 0;
    }
}

// HDRFloatComplex.h:166-173
inline HC CFromH(H re, H im)
{
    HC c;
    c.e = re.e > im.e ? re.e : im.e;
    c.re = re.m * getMultiplier(re.e - c.e);
    c.im = im.m * getMultiplier(im.e - c.e);
    return c;
}
// HDRFloatComplex.h:160-163
inline HC CFromFloats(float re, float im) { return CFromH(HFromMant(re), HFromMant(im)); }
inline HC CZero() { return HC{0.0f, 0.0f, MINEXP}; }
inline H CRe(HC c) { return H{c.re, c.e}; }
inline H CIm(HC c) { return H{c.im, c.e}; }
// HDRFloatComplex.h:219-247
inline HC CAdd(HC a, HC v)
{
    const int32_t expDiff = a.e - v.e;
    if (expDiff >= DIFF_IGNORED) {
        return a;
    } else if (expDiff >= 0) {
        const float mul = getMultiplier(-expDiff);
        a.re = a.re + v.re * mul;
        a.im = a.im + v.im * mul;
    } else if (expDiff > -DIFF_IGNORED) {
        const float mul = getMultiplier(expDiff);
        a.e = v.e;
        a.re = a.re * mul + v.re;
        a.im = a.im * mul + v.im;
    } else {
        a.e = v.e;
        a.re = v.re;
        a.im = v.im;
    }
    return a;
}
// HDRFloatComplex.h:267-283
inline HC CMul(HC a, HC f)
{
    const float re = (a.re * f.re) - (a.im * f.im);
    const float im = (a.re * f.im) + (a.im * f.re);
    return HC{re, im, clampE(a.e + f.e)};
}
// HDRFloatComplex.h:334-348
inline HC CMulH(HC a, H f) { return HC{a.re * f.m, a.im * f.m, clampE(a.e + f.e)}; }
// HDRFloatComplex.h:473-510
inline void CReduce(HC &a)
{
    if (a.re == 0.0f && a.im == 0.0f)
        return;
    const int32_t f_expReal = (int32_t)((f2u(a.re) & 0x7F800000u) >> 23);
    const int32_t f_expImag = (int32_t)((f2u(a.im) & 0x7F800000u) >> 23);
    const int32_t expDiff = (f_expReal > f_expImag ? f_expReal : f_expImag) + (-127);
    const int32_t expCombined = a.e + expDiff;
    const float mul = getMultiplier(-expDiff);
    a.re *= mul;
    a.im *= mul;
    a.e = expCombined;
}
// HDRFloatComplex.h:544-548
inline H CNormSq(HC a) { return H{a.re * a.re + a.im * a.im, a.e << 1}; }
// HDRFloatComplex.h:691-695 with HdrAbs (HDRFloat.h:1385-1404) and maxBothPositiveReduced
inline H CCheb(HC a)
{
    const H x{fabsf(a.re), a.e};
    const H y{fabsf(a.im), a.e};
    return CmpPosReduced(x, y) > 0 ? x : y;
}

inline H ld(const fs_real_hdr32 &r) { return H{r.m, r.e}; }
inline HC ld(const fs_cplx_hdr32 &c) { return HC{c.re, c.im, c.e}; }

// PerturbationResults::GetComplex<float>, PerturbationResults.h:174-185: complex built from the stored
// (un-reduced) x and y.
inline HC OrbitAt(const fs_orbit_hdr32 *orb, uint64_t i) { return CFromH(H{orb[i].mx, orb[i].ex}, H{orb[i].my, orb[i].ey}); }

// Row-claiming thread pool, Fractal.cpp:2523-2543.
// Rows y0, y0+g_row_step, ... < y1 (g_row_step = 1 is the reference; a larger step selects an evenly spread sample
// of rows of the same frame for the bounded cpu_baseline timing, orc_set_row_step()).
static uint32_t g_row_step = 1;
template <class RowFn> void run_rows(uint32_t y0, uint32_t y1, int threads, RowFn fn)
{
    std::deque<std::atomic_uint64_t> atomics;
    atomics.resize(y1);
    const uint32_t step = g_row_step ? g_row_step : 1;
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

// Pixel -> delta c, Fractal.cpp:2272-2281 == 2553-2562.
inline void pixel_delta(const H dx, const H dy, const H centerX, const H centerY, size_t x, size_t y, H &dRe, H &dIm)
{
    H deltaReal = MulByFloat(dx, (float)x);
    Reduce(deltaReal);
    deltaReal = Sub(deltaReal, centerX);
    H deltaImaginary = MulByFloat(Neg(dy), (float)y);
    Reduce(deltaImaginary);
    deltaImaginary = Sub(deltaImaginary, centerY);
    Reduce(deltaReal);
    Reduce(deltaImaginary);
    dRe = deltaReal;
    dIm = deltaImaginary;
}

struct BlaTable {
    const fs_bla_hdr32 *const *levels; // indexed by level; entries below firstLevel are null
    const uint64_t *sizes;
    int32_t n_levels; // m_B.size(); 0 = lookup suppressed
    int32_t lm2;
    static constexpr int32_t firstLevel = 2; // BLAS.h:22
};

// BLAS::LookupBackwards, BLAS.cpp:256-310
const fs_bla_hdr32 *LookupBackwards(const BlaTable &B, size_t m, H z2)
{
    if (B.n_levels == 0)
        return nullptr;
    if (m == 0)
        return nullptr;
    const int32_t k = (int32_t)m - 1;
    if ((k & 1) == 1)
        return nullptr;
    int32_t zeros;
    uint32_t ix;
    if (k == 0) {
        if (CmpPosReduced(z2, ld(B.levels[B.firstLevel][0].r2)) >= 0)
            return nullptr;
        zeros = 32;
        ix = 0;
    } else {
        const float v = (float)(k & -k);
        const uint32_t bits = f2u(v);
        zeros = (int32_t)(bits >> 23) - 0x7f;
        ix = (uint32_t)k >> zeros;
    }
    const int32_t startLevel = (zeros <= B.lm2) ? zeros : B.lm2;
    for (int32_t level = startLevel; level >= B.firstLevel; --level) {
        const fs_bla_hdr32 *t = &B.levels[level][ix];
        if (CmpPosReduced(z2, ld(t->r2)) < 0)
            return t;
        ix = ix << 1;
    }
    return nullptr;
}

} // namespace

extern "C" {

void orc_set_row_step(uint32_t step) { g_row_step = step ? step : 1; }

// Fractal::CalcCpuHDR<uint32_t,double,double>, Fractal.cpp:2096-2206.  coords = {dx, dy, minX, maxY}.
void orc_direct_f64(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const double coords[4],
                    uint32_t n_iterations, uint32_t *out, uint32_t stride, int threads)
{
    (void)height;
    const double dx = coords[0], dy = coords[1], minX = coords[2], maxY = coords[3];
    const double Four = 4, Two = 2;
    run_rows(y0, y1, threads, [&](uint32_t y) {
        double cx = minX;
        const double cy = maxY - dy * (double)((float)y);
        double zx, zy, zx2, zy2, sum;
        unsigned int i;
        for (size_t x = 0; x < width; x++) {
            zx = cx;
            zy = cy;
            for (i = 0; i < n_iterations; i++) {
                zx2 = zx * zx;
                zy2 = zy * zy;
                sum = zx2 + zy2;
                if (sum > Four)
                    break;
                zy = Two * zx * zy;
                zx = zx2 - zy2;
                zx += cx;
                zy += cy;
            }
            cx += dx;
            out[(size_t)y * stride + x] = i;
        }
    });
}

// Fractal::CalcCpuPerturbationFractalBLA<uint32_t,HDRFloat<float>,float>, Fractal.cpp:2208-2483.
// coords = {dx, dy, centerX, centerY} (already reduced).
void orc_bla_hdr32(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_orbit_hdr32 *orbit,
                   uint64_t orbit_count, const fs_real_hdr32 coords[4], uint32_t n_iterations,
                   const fs_bla_hdr32 *const *bla_levels, const uint64_t *bla_level_sizes, int32_t bla_n_levels,
                   int32_t bla_lm2, uint32_t *out, uint32_t stride, int threads)
{
    (void)height;
    const H dx = ld(coords[0]), dy = ld(coords[1]), centerX = ld(coords[2]), centerY = ld(coords[3]);
    const BlaTable blas{bla_levels, bla_level_sizes, bla_n_levels, bla_lm2};
    const uint32_t count = (uint32_t)orbit_count;
    const H TwoFiftySix = HFromInt(256);
    run_rows(y0, y1, threads, [&](uint32_t y) {
        for (size_t x = 0; x < width; x++) {
            uint32_t iter = 0;
            uint32_t RefIteration = 0;
            H DeltaSub0X, DeltaSub0Y;
            pixel_delta(dx, dy, centerX, centerY, x, y, DeltaSub0X, DeltaSub0Y);
            H DeltaSubNX = HFromInt(0);
            H DeltaSubNY = HFromInt(0);
            H DeltaNormSquared = HFromInt(0);

            while (iter < n_iterations) {
                const fs_bla_hdr32 *b = nullptr;
                while ((b = LookupBackwards(blas, RefIteration, DeltaNormSquared)) != nullptr) {
                    const int l = b->l;
                    if (RefIteration + l >= count)
                        break; // "Out of bounds! :("
                    if (iter + l >= n_iterations)
                        break;
                    iter += l;
                    // BLA::getValue, BLA.cuh:21-38
                    {
                        const H Ax = ld(b->Ax), Ay = ld(b->Ay), Bx = ld(b->Bx), By = ld(b->By);
                        // left-associative, as written: ((Ax*zx - Ay*zy) + Bx*cx) - By*cy
                        const H zxn = Sub(Add(Sub(Mul(Ax, DeltaSubNX), Mul(Ay, DeltaSubNY)), Mul(Bx, DeltaSub0X)),
                                          Mul(By, DeltaSub0Y));
                        const H zyn = Add(Add(Add(Mul(Ax, DeltaSubNY), Mul(Ay, DeltaSubNX)), Mul(Bx, DeltaSub0Y)),
                                          Mul(By, DeltaSub0X));
                        DeltaSubNX = zxn;
                        DeltaSubNY = zyn;
                    }
                    RefIteration += l;
                    const HC tempZComplex = OrbitAt(orbit, RefIteration);
                    const H tempZX = Add(CRe(tempZComplex), DeltaSubNX);
                    const H tempZY = Add(CIm(tempZComplex), DeltaSubNY);
                    H normSquared = Add(Mul(tempZX, tempZX), Mul(tempZY, tempZY));
                    DeltaNormSquared = Add(Mul(DeltaSubNX, DeltaSubNX), Mul(DeltaSubNY, DeltaSubNY));
                    Reduce(normSquared);
                    Reduce(DeltaNormSquared);
                    if (CmpPosReduced(normSquared, TwoFiftySix) > 0)
                        break;
                    if (CmpPosReduced(normSquared, DeltaNormSquared) < 0 || RefIteration >= count - 1) {
                        DeltaSubNX = tempZX;
                        DeltaSubNY = tempZY;
                        DeltaNormSquared = normSquared;
                        RefIteration = 0;
                    }
                }
                if (iter >= n_iterations)
                    break;

                const H DeltaSubNXOrig = DeltaSubNX;
                const H DeltaSubNYOrig = DeltaSubNY;
                const HC tempZComplex = OrbitAt(orbit, RefIteration);
                const H TermB1 = Mul(DeltaSubNXOrig, Add(MulBy2(CRe(tempZComplex)), DeltaSubNXOrig));
                const H TermB2 = Mul(DeltaSubNYOrig, Add(MulBy2(CIm(tempZComplex)), DeltaSubNYOrig));
                DeltaSubNX = Sub(TermB1, TermB2);
                DeltaSubNX = Add(DeltaSubNX, DeltaSub0X);
                Reduce(DeltaSubNX);

                const H Term3 = Add(MulBy2(CIm(tempZComplex)), DeltaSubNYOrig);
                const H Term4 = Add(MulBy2(CRe(tempZComplex)), DeltaSubNXOrig);
                DeltaSubNY = Add(Mul(DeltaSubNXOrig, Term3), Mul(DeltaSubNYOrig, Term4));
                DeltaSubNY = Add(DeltaSubNY, DeltaSub0Y);
                Reduce(DeltaSubNY);

                ++RefIteration;
                if (RefIteration >= count)
                    break; // "Out of bounds 2! :("

                const HC tempZComplex2 = OrbitAt(orbit, RefIteration);
                const H tempZX = Add(CRe(tempZComplex2), DeltaSubNX);
                const H tempZY = Add(CIm(tempZComplex2), DeltaSubNY);
                const H nT1 = Mul(tempZX, tempZX);
                const H nT2 = Mul(tempZY, tempZY);
                H normSquared = Add(nT1, nT2);
                Reduce(normSquared);
                DeltaNormSquared = Add(Mul(DeltaSubNX, DeltaSubNX), Mul(DeltaSubNY, DeltaSubNY));
                Reduce(DeltaNormSquared);
                if (CmpPosReduced(normSquared, TwoFiftySix) > 0)
                    break;
                if (CmpPosReduced(normSquared, DeltaNormSquared) < 0 || RefIteration >= count - 1) {
                    DeltaSubNX = tempZX;
                    DeltaSubNY = tempZY;
                    DeltaNormSquared = normSquared;
                    RefIteration = 0;
                }
                ++iter;
            }
            out[(size_t)y * stride + x] = iter;
        }
    });
}

// Fractal::CalcCpuPerturbationFractalLAV2<uint32_t,float,PerturbExtras::Disable>, Fractal.cpp:2485-2691.
// stage_test: 0 = literal CPU LAReference::isLAStageInvalid (cheb(dc) <  LAThresholdC, LAReference.cpp:1076-1081)
//             1 = the direction the GPU twin uses      (cheb(dc) >= LAThresholdC, GPU_LAReference.h:240-254)
// mode: 0 = Full, 1 = PO (skip AT + LA stages), 2 = LAO (skip the perturbation loop) -- LAv2Mode of the GPU
//       path (RenderAlgorithm.h:12-17); the CPU function is always Full.
void orc_lav2_hdr32(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_orbit_hdr32 *orbit,
                    uint64_t orbit_count, uint64_t period_maybe_zero, const fs_la_hdr32_u32 *las, uint32_t n_las,
                    const fs_la_stage_u32 *stages, uint32_t stage_count, int la_valid, int use_at,
                    const fs_at_hdr32_u32 *at, const fs_real_hdr32 coords[4], uint32_t n_iterations,
                    int stage_test, int mode, uint32_t *out, uint32_t stride, int threads, uint64_t *stats)
{
    // stats (optional, 4 x uint64): [0] AT iterations, [1] LA steps taken, [2] perturbation steps, [3] pixels
    (void)height;
    (void)n_las;
    std::atomic<uint64_t> st_at{0}, st_la{0}, st_pt{0}, st_px{0};
    const H dx = ld(coords[0]), dy = ld(coords[1]), centerX = ld(coords[2]), centerY = ld(coords[3]);
    const H TwoFiftySix = HFromInt(256);
    const H Two = HFromInt(2);
    run_rows(y0, y1, threads, [&](uint32_t y) {
        uint64_t c_at = 0, c_la = 0, c_pt = 0;
        for (size_t x = 0; x < width; x++) {
            uint32_t BLA2SkippedIterations = 0;
            H deltaReal, deltaImaginary;
            pixel_delta(dx, dy, centerX, centerY, x, y, deltaReal, deltaImaginary);
            const HC DeltaSub0 = CFromH(deltaReal, deltaImaginary);
            HC DeltaSubN = CFromFloats(0.0f, 0.0f); // {0, 0}: zero with exponent 0, SURVEY 0.11

            if (mode != 1 && la_valid && use_at &&
                CmpPosReduced(CCheb(DeltaSub0), ld(at->ThresholdC)) <= 0) { // ATInfo::isValid
                // ATInfo::PerformAT, ATInfo.h:155-188
                const uint32_t ATMaxIt = n_iterations / at->StepLength;
                HC c = CAdd(CMul(DeltaSub0, ld(at->CCoeff)), ld(at->RefC)); // getC
                CReduce(c);
                HC z = CZero();
                uint32_t i;
                for (i = 0; i < ATMaxIt; i++) {
                    H nsq = CNormSq(z);
                    Reduce(nsq);
                    if (CmpPosReduced(nsq, ld(at->SqrEscapeRadius)) > 0)
                        break;
                    z = CAdd(CMul(z, z), c);
                }
                HC dz = CMul(z, ld(at->InvZCoeff)); // getDZ
                CReduce(dz);
                DeltaSubN = dz;
                BLA2SkippedIterations = i * at->StepLength;
                c_at += i;
            }

            uint32_t iterations = 0;
            uint32_t RefIteration = 0;
            const uint32_t MaxRefIteration = (uint32_t)orbit_count - 1;
            iterations = BLA2SkippedIterations;
            HC complex0 = CFromH(deltaReal, deltaImaginary);

            if (iterations != 0 && RefIteration < MaxRefIteration) {
                complex0 = CAdd(OrbitAt(orbit, RefIteration), DeltaSubN);
            } else if (iterations != 0 && period_maybe_zero != 0) {
                RefIteration = RefIteration % (uint32_t)period_maybe_zero;
                complex0 = CAdd(OrbitAt(orbit, RefIteration), DeltaSubN);
            }

            uint32_t CurrentLAStage = (la_valid && mode != 1) ? stage_count : 0;
            while (CurrentLAStage > 0) {
                CurrentLAStage--;
                const uint32_t LAIndex = stages[CurrentLAStage].LAIndex;
                {
                    const int c = CmpPosReduced(CCheb(DeltaSub0), ld(las[LAIndex].LAThresholdC));
                    const bool invalid = stage_test == 0 ? (c < 0) : (c >= 0);
                    if (invalid)
                        continue;
                }
                const uint32_t MacroItCount = stages[CurrentLAStage].MacroItCount;
                uint32_t j = RefIteration;
                while (iterations < n_iterations) {
                    // LAReference::getLA, LAReference.cpp:1097-1134
                    const uint32_t LAIndexj = LAIndex + j;
                    const fs_la_hdr32_u32 &LAj = las[LAIndexj];
                    const uint32_t l = LAj.StepLength;
                    const bool usable = iterations + l <= n_iterations;
                    bool unusable = true;
                    HC newDz = CZero();
                    if (usable) {
                        // LAInfoDeep::Prepare, LAInfoDeep.h:395-414
                        newDz = CMul(DeltaSubN, CAdd(CMulH(ld(LAj.Ref), Two), DeltaSubN));
                        CReduce(newDz);
                        unusable = CmpPosReduced(CCheb(newDz), ld(LAj.LAThreshold)) >= 0;
                    }
                    if (unusable) {
                        RefIteration = LAj.NextStageLAIndex;
                        break;
                    }
                    iterations += l;
                    c_la++;
                    // LAInfoDeep::Evaluate, LAInfoDeep.h:416-420
                    DeltaSubN = CAdd(CMul(newDz, ld(LAj.ZCoeff)), CMul(DeltaSub0, ld(LAj.CCoeff)));
                    // LAstep::getZ, LAstep.h:116-120
                    complex0 = CAdd(ld(las[LAIndexj + 1].Ref), DeltaSubN);
                    j++;
                    H lhs = CCheb(complex0);
                    Reduce(lhs);
                    H rhs = CCheb(DeltaSubN);
                    Reduce(rhs);
                    if (CmpPosReduced(lhs, rhs) < 0 || j >= MacroItCount) {
                        DeltaSubN = complex0;
                        j = 0;
                    }
                }
                if (iterations >= n_iterations)
                    break;
            }

            if (mode != 2) {
                for (; iterations < n_iterations; iterations++) {
                    HC curIter = OrbitAt(orbit, RefIteration);
                    curIter = CMulH(curIter, Two);
                    curIter = CAdd(curIter, DeltaSubN);
                    DeltaSubN = CMul(DeltaSubN, curIter);
                    DeltaSubN = CAdd(DeltaSubN, DeltaSub0);
                    CReduce(DeltaSubN);
                    c_pt++;
                    RefIteration++;
                    complex0 = CAdd(OrbitAt(orbit, RefIteration), DeltaSubN);
                    CReduce(complex0);
                    H normSquared = CNormSq(complex0);
                    Reduce(normSquared);
                    H DeltaNormSquared = CNormSq(DeltaSubN);
                    Reduce(DeltaNormSquared);
                    if (CmpPosReduced(normSquared, TwoFiftySix) > 0)
                        break;
                    if (CmpPosReduced(normSquared, DeltaNormSquared) < 0 || RefIteration >= MaxRefIteration) {
                        DeltaSubN = complex0;
                        RefIteration = 0;
                    }
                }
            }
            out[(size_t)y * stride + x] = iterations;
        }
        st_at += c_at;
        st_la += c_la;
        st_pt += c_pt;
        st_px += width;
    });
    if (stats) {
        stats[0] = st_at;
        stats[1] = st_la;
        stats[2] = st_pt;
        stats[3] = st_px;
    }
}

} // extern "C"
