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
// Every perturbation function is a template over F in {float, double} = HDRFloat<float> / HDRFloat<double>
// (Cpu32* / Cpu64* algorithms); orc_direct_hdr{32,64} restate CalcCpuHDR<uint32_t,HDRFloat<F>,F> (CpuHDR32/64).
// Numeric types: a self-contained restatement of HDRFloat<F> (HpSharkFloatLib/HDRFloat.h) and
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

template <class F> struct Bits;
template <> struct Bits<float> {
    using U = uint32_t;
    static constexpr U EXPMASK = 0x7F800000u, KEEP = 0x807FFFFFu, ONE = 0x3F800000u;
    static constexpr int SHIFT = 23, BIAS = 127, MULMAX = 128;
    static float maxval() { return 3.402823466e+38f; }
    static float scalb(int s) { return scalbnf(1.0f, s); }
};
template <> struct Bits<double> {
    using U = uint64_t;
    static constexpr U EXPMASK = 0x7FF0000000000000ull, KEEP = 0x800FFFFFFFFFFFFFull, ONE = 0x3FF0000000000000ull;
    static constexpr int SHIFT = 52, BIAS = 1023, MULMAX = 1024;
    static double maxval() { return 1.7976931348623157e+308; }
    static double scalb(int s) { return scalbn(1.0, s); }
};

template <class F> struct HT {
    F m;
    int32_t e;
};
template <class F> struct HCT {
    F re, im;
    int32_t e;
};

template <class F> inline typename Bits<F>::U f2u(F f)
{
    typename Bits<F>::U u;
    memcpy(&u, &f, sizeof(u));
    return u;
}
template <class F> inline F u2f(typename Bits<F>::U u)
{
    F f;
    memcpy(&f, &u, sizeof(f));
    return f;
}
template <class F> inline int32_t expField(F v) { return (int32_t)((f2u<F>(v) & Bits<F>::EXPMASK) >> Bits<F>::SHIFT); }

// HDRFloat.h:497-521
template <class F> inline F getMultiplier(int32_t s)
{
    if (s <= -Bits<F>::BIAS)
        return F(0);
    if (s >= Bits<F>::MULMAX)
        return Bits<F>::maxval();
    return Bits<F>::scalb(s);
}
// HDRFloat.h:523-551
template <class F> inline F getMultiplierNeg(int32_t s)
{
    if (s <= -Bits<F>::BIAS)
        return F(0);
    return Bits<F>::scalb(s);
}

// HDRFloat.h:414-457
template <class F> inline void Reduce(HT<F> &a)
{
    if (a.m == 0)
        return;
    const auto bits = f2u<F>(a.m);
    const int32_t f_exp = (int32_t)((bits & Bits<F>::EXPMASK) >> Bits<F>::SHIFT) - Bits<F>::BIAS;
    a.m = u2f<F>((bits & Bits<F>::KEEP) | Bits<F>::ONE);
    a.e += f_exp;
}
template <class F> inline HT<F> Reduced(HT<F> a)
{
    Reduce(a);
    return a;
}
// HDRFloat.h:200-204
template <class F> inline HT<F> HZero() { return HT<F>{F(0), MINEXP}; }
// HDRFloat.h:206-212 HDRFloat(T mant)
template <class F> inline HT<F> HFromMant(F v)
{
    HT<F> r{v, 0};
    Reduce(r);
    return r;
}
// HDRFloat.h:295-363 HDRFloat(U number), U in {int, float, double}
template <class F> inline HT<F> HFromNumber(F n)
{
    if (n == F(0))
        return HZero<F>();
    const auto bits = f2u<F>(n);
    return HT<F>{u2f<F>((bits & Bits<F>::KEEP) | Bits<F>::ONE),
                 (int32_t)((bits & Bits<F>::EXPMASK) >> Bits<F>::SHIFT) - Bits<F>::BIAS};
}
template <class F> inline HT<F> HFromInt(int n) { return HFromNumber<F>((F)n); }
inline int32_t clampE(int32_t e) { return e < MINEXP ? MINEXP : e; }
// HDRFloat.h:829-840
template <class F> inline HT<F> Mul(HT<F> a, HT<F> b) { return HT<F>{a.m * b.m, clampE(a.e + b.e)}; }
// operator*(HDRFloat, const T&): the scalar goes through HDRFloat(T mant)
template <class F> inline HT<F> MulBy2(HT<F> a) { return Mul(a, HFromMant<F>(F(2))); }
template <class F> inline HT<F> MulByScalar(HT<F> a, F f) { return Mul(a, HFromMant<F>(f)); }
// HDRFloat.h:974-1000
template <class F> inline HT<F> Add(HT<F> a, HT<F> b)
{
    const int32_t expDiff = a.e - b.e;
    if (expDiff >= DIFF_IGNORED) {
        return a;
    } else if (expDiff >= 0) {
        const F mul = getMultiplierNeg<F>(-expDiff);
        a.m = a.m + b.m * mul;
    } else if (expDiff > -DIFF_IGNORED) {
        const F mul = getMultiplierNeg<F>(expDiff);
        a.e = b.e;
        a.m = a.m * mul + b.m;
    } else {
        a.e = b.e;
        a.m = b.m;
    }
    if (a.m == F(0))
        a.e = MINEXP;
    return a;
}
// HDRFloat.h:1039-1065
template <class F> inline HT<F> Sub(HT<F> a, HT<F> b)
{
    const int32_t expDiff = a.e - b.e;
    if (expDiff >= DIFF_IGNORED) {
        return a;
    } else if (expDiff >= 0) {
        const F mul = getMultiplierNeg<F>(-expDiff);
        a.m = a.m - b.m * mul;
    } else if (expDiff > -DIFF_IGNORED) {
        const F mul = getMultiplierNeg<F>(expDiff);
        a.e = b.e;
        a.m = a.m * mul - b.m;
    } else {
        a.e = b.e;
        a.m = -b.m;
    }
    if (a.m == F(0))
        a.e = MINEXP;
    return a;
}
template <class F> inline HT<F> Neg(HT<F> a) { return HT<F>{-a.m, a.e}; }
// HDRFloat.h:1150-1167
template <class F> inline int CmpPosReduced(HT<F> a, HT<F> b)
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
template <class F> inline HCT<F> CFromH(HT<F> re, HT<F> im)
{
    HCT<F> c;
    c.e = re.e > im.e ? re.e : im.e;
    c.re = re.m * getMultiplier<F>(re.e - c.e);
    c.im = im.m * getMultiplier<F>(im.e - c.e);
    return c;
}
// HDRFloatComplex.h:160-163
template <class F> inline HCT<F> CFromScalars(F re, F im) { return CFromH(HFromMant<F>(re), HFromMant<F>(im)); }
template <class F> inline HCT<F> CZero() { return HCT<F>{F(0), F(0), MINEXP}; }
template <class F> inline HT<F> CRe(HCT<F> c) { return HT<F>{c.re, c.e}; }
template <class F> inline HT<F> CIm(HCT<F> c) { return HT<F>{c.im, c.e}; }
// HDRFloatComplex.h:219-247
template <class F> inline HCT<F> CAdd(HCT<F> a, HCT<F> v)
{
    const int32_t expDiff = a.e - v.e;
    if (expDiff >= DIFF_IGNORED) {
        return a;
    } else if (expDiff >= 0) {
        const F mul = getMultiplier<F>(-expDiff);
        a.re = a.re + v.re * mul;
        a.im = a.im + v.im * mul;
    } else if (expDiff > -DIFF_IGNORED) {
        const F mul = getMultiplier<F>(expDiff);
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
template <class F> inline HCT<F> CMul(HCT<F> a, HCT<F> f)
{
    const F re = (a.re * f.re) - (a.im * f.im);
    const F im = (a.re * f.im) + (a.im * f.re);
    return HCT<F>{re, im, clampE(a.e + f.e)};
}
// HDRFloatComplex.h:334-348
template <class F> inline HCT<F> CMulH(HCT<F> a, HT<F> f) { return HCT<F>{a.re * f.m, a.im * f.m, clampE(a.e + f.e)}; }
// HDRFloatComplex.h:473-510
template <class F> inline void CReduce(HCT<F> &a)
{
    if (a.re == F(0) && a.im == F(0))
        return;
    const int32_t f_expReal = expField<F>(a.re);
    const int32_t f_expImag = expField<F>(a.im);
    const int32_t expDiff = (f_expReal > f_expImag ? f_expReal : f_expImag) + (-Bits<F>::BIAS);
    const int32_t expCombined = a.e + expDiff;
    const F mul = getMultiplier<F>(-expDiff);
    a.re *= mul;
    a.im *= mul;
    a.e = expCombined;
}
// HDRFloatComplex.h:544-548
template <class F> inline HT<F> CNormSq(HCT<F> a) { return HT<F>{a.re * a.re + a.im * a.im, a.e << 1}; }
// HDRFloatComplex.h:691-695 with HdrAbs (HDRFloat.h:1385-1404) and maxBothPositiveReduced
template <class F> inline HT<F> CCheb(HCT<F> a)
{
    const HT<F> x{(F)fabs(a.re), a.e};
    const HT<F> y{(F)fabs(a.im), a.e};
    return CmpPosReduced(x, y) > 0 ? x : y;
}

// Record access for the two ABI layouts (include/fs_layout.h).
template <class F> struct Rec;
template <> struct Rec<float> {
    using Real = fs_real_hdr32;
    using Cplx = fs_cplx_hdr32;
    using Orbit = fs_orbit_hdr32;
    using LA = fs_la_hdr32_u32;
    using AT = fs_at_hdr32_u32;
    using BLA = fs_bla_hdr32;
};
template <> struct Rec<double> {
    using Real = fs_real_hdr64;
    using Cplx = fs_cplx_hdr64;
    using Orbit = fs_orbit_hdr64;
    using LA = fs_la_hdr64_u32;
    using AT = fs_at_hdr64_u32;
    using BLA = fs_bla_hdr64;
};
inline HT<float> ld(const fs_real_hdr32 &r) { return HT<float>{r.m, r.e}; }
inline HCT<float> ld(const fs_cplx_hdr32 &c) { return HCT<float>{c.re, c.im, c.e}; }
inline HT<double> ld(const fs_real_hdr64 &r) { return HT<double>{r.m, r.e}; }
inline HCT<double> ld(const fs_cplx_hdr64 &c) { return HCT<double>{c.re, c.im, c.e}; }

// PerturbationResults::GetComplex<SubType>, PerturbationResults.h:174-185: complex built from the stored
// (un-reduced) x and y.
inline HCT<float> OrbitAt(const fs_orbit_hdr32 *orb, uint64_t i)
{
    return CFromH(HT<float>{orb[i].mx, orb[i].ex}, HT<float>{orb[i].my, orb[i].ey});
}
inline HCT<double> OrbitAt(const fs_orbit_hdr64 *orb, uint64_t i)
{
    return CFromH(HT<double>{orb[i].mx, orb[i].ex}, HT<double>{orb[i].my, orb[i].ey});
}

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
template <class F>
inline void pixel_delta(const HT<F> dx, const HT<F> dy, const HT<F> centerX, const HT<F> centerY, size_t x, size_t y,
                        HT<F> &dRe, HT<F> &dIm)
{
    HT<F> deltaReal = MulByScalar<F>(dx, (F)x);
    Reduce(deltaReal);
    deltaReal = Sub(deltaReal, centerX);
    HT<F> deltaImaginary = MulByScalar<F>(Neg(dy), (F)y);
    Reduce(deltaImaginary);
    deltaImaginary = Sub(deltaImaginary, centerY);
    Reduce(deltaReal);
    Reduce(deltaImaginary);
    dRe = deltaReal;
    dIm = deltaImaginary;
}

template <class F> struct BlaTable {
    const typename Rec<F>::BLA *const *levels; // indexed by level; entries below firstLevel are null
    const uint64_t *sizes;
    int32_t n_levels; // m_B.size(); 0 = lookup suppressed
    int32_t lm2;
    static constexpr int32_t firstLevel = 2; // BLAS.h:22
};

// BLAS::LookupBackwards, BLAS.cpp:256-310
template <class F> const typename Rec<F>::BLA *LookupBackwards(const BlaTable<F> &B, size_t m, HT<F> z2)
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
        const uint32_t bits = f2u<float>(v);
        zeros = (int32_t)(bits >> 23) - 0x7f;
        ix = (uint32_t)k >> zeros;
    }
    const int32_t startLevel = (zeros <= B.lm2) ? zeros : B.lm2;
    for (int32_t level = startLevel; level >= B.firstLevel; --level) {
        const typename Rec<F>::BLA *t = &B.levels[level][ix];
        if (CmpPosReduced(z2, ld(t->r2)) < 0)
            return t;
        ix = ix << 1;
    }
    return nullptr;
}

// Fractal::CalcCpuHDR<uint32_t,HDRFloat<F>,F>, Fractal.cpp:2096-2206 (CpuHDR32 / CpuHDR64).
// coords = {dx, dy, minX, maxY} as HDRFloat built from mpf (mantissa in [0.5,1), NOT reduced).
template <class F>
void direct_hdr_impl(uint32_t width, uint32_t y0, uint32_t y1, const typename Rec<F>::Real coords[4],
                     uint32_t n_iterations, uint32_t *out, uint32_t stride, int threads)
{
    using H = HT<F>;
    const H dx = ld(coords[0]), dy = ld(coords[1]), minX = ld(coords[2]), maxY = ld(coords[3]);
    const H Four = HFromInt<F>(4);
    const H Two = HFromInt<F>(2);
    run_rows(y0, y1, threads, [&](uint32_t y) {
        H cx = minX;
        // T{static_cast<float>(y)}: HDRFloat<float> takes the non-template HDRFloat(T mant) ctor ({0,0} for y = 0),
        // HDRFloat<double> the templated one with U = float ({0, MINEXP} for y = 0).
        const H yh = sizeof(F) == 4 ? HFromMant<F>((F)(float)y) : HFromNumber<F>((F)(float)y);
        const H cy = Sub(maxY, Mul(dy, yh));
        H zx, zy, zx2, zy2, sum;
        unsigned int i;
        for (size_t x = 0; x < width; x++) {
            zx = cx;
            zy = cy;
            for (i = 0; i < n_iterations; i++) {
                zx2 = Mul(zx, zx);
                zy2 = Mul(zy, zy);
                sum = Add(zx2, zy2);
                Reduce(sum);
                if (CmpPosReduced(sum, Four) > 0)
                    break;
                zy = Mul(Mul(Two, zx), zy);
                zx = Sub(zx2, zy2);
                zx = Add(zx, cx);
                zy = Add(zy, cy);
                Reduce(zx);
                Reduce(zy);
            }
            cx = Add(cx, dx);
            out[(size_t)y * stride + x] = i;
        }
    });
}

// Fractal::CalcCpuPerturbationFractalBLA<uint32_t,HDRFloat<F>,F>, Fractal.cpp:2208-2483.
// coords = {dx, dy, centerX, centerY} (already reduced).
template <class F>
void bla_impl(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const typename Rec<F>::Orbit *orbit,
              uint64_t orbit_count, const typename Rec<F>::Real coords[4], uint32_t n_iterations,
              const typename Rec<F>::BLA *const *bla_levels, const uint64_t *bla_level_sizes, int32_t bla_n_levels,
              int32_t bla_lm2, uint32_t *out, uint32_t stride, int threads)
{
    using H = HT<F>;
    using HC = HCT<F>;
    (void)height;
    const H dx = ld(coords[0]), dy = ld(coords[1]), centerX = ld(coords[2]), centerY = ld(coords[3]);
    const BlaTable<F> blas{bla_levels, bla_level_sizes, bla_n_levels, bla_lm2};
    const uint32_t count = (uint32_t)orbit_count;
    const H TwoFiftySix = HFromInt<F>(256);
    run_rows(y0, y1, threads, [&](uint32_t y) {
        for (size_t x = 0; x < width; x++) {
            uint32_t iter = 0;
            uint32_t RefIteration = 0;
            H DeltaSub0X, DeltaSub0Y;
            pixel_delta<F>(dx, dy, centerX, centerY, x, y, DeltaSub0X, DeltaSub0Y);
            H DeltaSubNX = HFromInt<F>(0);
            H DeltaSubNY = HFromInt<F>(0);
            H DeltaNormSquared = HFromInt<F>(0);

            while (iter < n_iterations) {
                const typename Rec<F>::BLA *b = nullptr;
                while ((b = LookupBackwards<F>(blas, RefIteration, DeltaNormSquared)) != nullptr) {
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

// Fractal::CalcCpuPerturbationFractalLAV2<uint32_t,F,PerturbExtras::Disable>, Fractal.cpp:2485-2691.
// stage_test: 0 = literal CPU LAReference::isLAStageInvalid (cheb(dc) <  LAThresholdC, LAReference.cpp:1076-1081)
//             1 = the direction the GPU twin uses      (cheb(dc) >= LAThresholdC, GPU_LAReference.h:240-254)
// mode: 0 = Full, 1 = PO (skip AT + LA stages), 2 = LAO (skip the perturbation loop) -- LAv2Mode of the GPU
//       path (RenderAlgorithm.h:12-17); the CPU function is always Full.
// IterT = the reference's IterType for the counters (uint32_t, or uint64_t for iteration caps of 2^32 and above); table
// fields stay 32-bit like the product's device records.
template <class F, class IterT = uint32_t>
void lav2_impl(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const typename Rec<F>::Orbit *orbit,
               uint64_t orbit_count, uint64_t period_maybe_zero, const typename Rec<F>::LA *las, uint32_t n_las,
               const fs_la_stage_u32 *stages, uint32_t stage_count, int la_valid, int use_at,
               const typename Rec<F>::AT *at, const typename Rec<F>::Real coords[4], IterT n_iterations,
               int stage_test, int mode, IterT *out, uint32_t stride, int threads, uint64_t *stats)
{
    using H = HT<F>;
    using HC = HCT<F>;
    // stats (optional, 4 x uint64): [0] AT iterations, [1] LA steps taken, [2] perturbation steps, [3] pixels
    (void)height;
    (void)n_las;
    std::atomic<uint64_t> st_at{0}, st_la{0}, st_pt{0}, st_px{0};
    const H dx = ld(coords[0]), dy = ld(coords[1]), centerX = ld(coords[2]), centerY = ld(coords[3]);
    const H TwoFiftySix = HFromInt<F>(256);
    const H Two = HFromInt<F>(2);
    run_rows(y0, y1, threads, [&](uint32_t y) {
        uint64_t c_at = 0, c_la = 0, c_pt = 0;
        for (size_t x = 0; x < width; x++) {
            IterT BLA2SkippedIterations = 0;
            H deltaReal, deltaImaginary;
            pixel_delta<F>(dx, dy, centerX, centerY, x, y, deltaReal, deltaImaginary);
            const HC DeltaSub0 = CFromH(deltaReal, deltaImaginary);
            HC DeltaSubN = CFromScalars<F>(F(0), F(0)); // {0, 0}: zero with exponent 0, SURVEY 0.11

            if (mode != 1 && la_valid && use_at &&
                CmpPosReduced(CCheb(DeltaSub0), ld(at->ThresholdC)) <= 0) { // ATInfo::isValid
                // ATInfo::PerformAT, ATInfo.h:155-188
                const IterT ATMaxIt = n_iterations / at->StepLength;
                HC c = CAdd(CMul(DeltaSub0, ld(at->CCoeff)), ld(at->RefC)); // getC
                CReduce(c);
                HC z = CZero<F>();
                IterT i;
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

            IterT iterations = 0;
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
                    const typename Rec<F>::LA &LAj = las[LAIndexj];
                    const uint32_t l = LAj.StepLength;
                    const bool usable = iterations + l <= n_iterations;
                    bool unusable = true;
                    HC newDz = CZero<F>();
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

} // namespace

extern "C" {

void orc_set_row_step(uint32_t step) { g_row_step = step ? step : 1; }
uint32_t orc_get_row_step(void) { return g_row_step; } // shared with gpu_ref_2x32.cpp

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


void orc_direct_hdr32(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_real_hdr32 coords[4],
                      uint32_t n_iterations, uint32_t *out, uint32_t stride, int threads)
{
    (void)height;
    direct_hdr_impl<float>(width, y0, y1, coords, n_iterations, out, stride, threads);
}
void orc_direct_hdr64(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_real_hdr64 coords[4],
                      uint32_t n_iterations, uint32_t *out, uint32_t stride, int threads)
{
    (void)height;
    direct_hdr_impl<double>(width, y0, y1, coords, n_iterations, out, stride, threads);
}

// Fractal::CalcCpuPerturbationFractalBLA<uint32_t,double,double> (Cpu64PerturbedBLA), Fractal.cpp:2208-2483 with
// T = double: every HdrReduce is a no-op, comparisons are plain, GetComplex returns FloatComplex<double>.
// coords = {dx, dy, centerX, centerY} as doubles.
void orc_bla_f64(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_orbit_f64 *orbit,
                 uint64_t orbit_count, const double coords[4], uint32_t n_iterations,
                 const fs_bla_f64 *const *B, const uint64_t *sizes, int32_t n_levels, int32_t lm2, uint32_t *out,
                 uint32_t stride, int threads)
{
    (void)height;
    (void)sizes;
    const double dx = coords[0], dy = coords[1], centerX = coords[2], centerY = coords[3];
    const uint32_t count = (uint32_t)orbit_count;
    auto Lookup = [&](size_t m, double z2) -> const fs_bla_f64 * { // BLAS::LookupBackwards, BLAS.cpp:256-310
        if (n_levels == 0 || m == 0)
            return nullptr;
        const int32_t k = (int32_t)m - 1;
        if ((k & 1) == 1)
            return nullptr;
        int32_t zeros;
        uint32_t ix;
        if (k == 0) {
            if (z2 >= B[2][0].r2)
                return nullptr;
            zeros = 32;
            ix = 0;
        } else {
            const float v = (float)(k & -k);
            uint32_t bits;
            memcpy(&bits, &v, 4);
            zeros = (int32_t)(bits >> 23) - 0x7f;
            ix = (uint32_t)k >> zeros;
        }
        const int32_t startLevel = zeros <= lm2 ? zeros : lm2;
        for (int32_t level = startLevel; level >= 2; --level) {
            const fs_bla_f64 *t = &B[level][ix];
            if (z2 < t->r2)
                return t;
            ix = ix << 1;
        }
        return nullptr;
    };
    run_rows(y0, y1, threads, [&](uint32_t y) {
        for (size_t x = 0; x < width; x++) {
            uint32_t iter = 0, RefIteration = 0;
            double deltaReal = dx * (double)x;
            deltaReal -= centerX;
            double deltaImaginary = -dy * (double)y;
            deltaImaginary -= centerY;
            const double DeltaSub0X = deltaReal, DeltaSub0Y = deltaImaginary;
            double DeltaSubNX = 0, DeltaSubNY = 0, DeltaNormSquared = 0;
            while (iter < n_iterations) {
                const fs_bla_f64 *b = nullptr;
                while ((b = Lookup(RefIteration, DeltaNormSquared)) != nullptr) {
                    const int l = b->l;
                    if (RefIteration + l >= count)
                        break;
                    if (iter + l >= n_iterations)
                        break;
                    iter += l;
                    {
                        const double nx = b->Ax * DeltaSubNX - b->Ay * DeltaSubNY + b->Bx * DeltaSub0X - b->By * DeltaSub0Y;
                        const double ny = b->Ax * DeltaSubNY + b->Ay * DeltaSubNX + b->Bx * DeltaSub0Y + b->By * DeltaSub0X;
                        DeltaSubNX = nx;
                        DeltaSubNY = ny;
                    }
                    RefIteration += l;
                    const double tempZX = orbit[RefIteration].x + DeltaSubNX;
                    const double tempZY = orbit[RefIteration].y + DeltaSubNY;
                    const double normSquared = tempZX * tempZX + tempZY * tempZY;
                    DeltaNormSquared = DeltaSubNX * DeltaSubNX + DeltaSubNY * DeltaSubNY;
                    if (normSquared > 256.0)
                        break;
                    if (normSquared < DeltaNormSquared || RefIteration >= count - 1) {
                        DeltaSubNX = tempZX;
                        DeltaSubNY = tempZY;
                        DeltaNormSquared = normSquared;
                        RefIteration = 0;
                    }
                }
                if (iter >= n_iterations)
                    break;
                const double OX = DeltaSubNX, OY = DeltaSubNY;
                const double Zre = orbit[RefIteration].x, Zim = orbit[RefIteration].y;
                const double TermB1 = OX * (Zre * 2 + OX);
                const double TermB2 = OY * (Zim * 2 + OY);
                DeltaSubNX = TermB1 - TermB2;
                DeltaSubNX += DeltaSub0X;
                const double Term3 = Zim * 2 + OY;
                const double Term4 = Zre * 2 + OX;
                DeltaSubNY = OX * Term3 + OY * Term4;
                DeltaSubNY += DeltaSub0Y;
                ++RefIteration;
                if (RefIteration >= count)
                    break;
                const double tempZX = orbit[RefIteration].x + DeltaSubNX;
                const double tempZY = orbit[RefIteration].y + DeltaSubNY;
                const double nT1 = tempZX * tempZX;
                const double nT2 = tempZY * tempZY;
                const double normSquared = nT1 + nT2;
                DeltaNormSquared = DeltaSubNX * DeltaSubNX + DeltaSubNY * DeltaSubNY;
                if (normSquared > 256.0)
                    break;
                if (normSquared < DeltaNormSquared || RefIteration >= count - 1) {
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

void orc_bla_hdr32(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_orbit_hdr32 *orbit,
                   uint64_t orbit_count, const fs_real_hdr32 coords[4], uint32_t n_iterations,
                   const fs_bla_hdr32 *const *bla_levels, const uint64_t *bla_level_sizes, int32_t bla_n_levels,
                   int32_t bla_lm2, uint32_t *out, uint32_t stride, int threads)
{
    bla_impl<float>(width, height, y0, y1, orbit, orbit_count, coords, n_iterations, bla_levels, bla_level_sizes,
                    bla_n_levels, bla_lm2, out, stride, threads);
}
void orc_bla_hdr64(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_orbit_hdr64 *orbit,
                   uint64_t orbit_count, const fs_real_hdr64 coords[4], uint32_t n_iterations,
                   const fs_bla_hdr64 *const *bla_levels, const uint64_t *bla_level_sizes, int32_t bla_n_levels,
                   int32_t bla_lm2, uint32_t *out, uint32_t stride, int threads)
{
    bla_impl<double>(width, height, y0, y1, orbit, orbit_count, coords, n_iterations, bla_levels, bla_level_sizes,
                     bla_n_levels, bla_lm2, out, stride, threads);
}

// stage_test / mode / stats: see lav2_impl.
void orc_lav2_hdr32(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_orbit_hdr32 *orbit,
                    uint64_t orbit_count, uint64_t period_maybe_zero, const fs_la_hdr32_u32 *las, uint32_t n_las,
                    const fs_la_stage_u32 *stages, uint32_t stage_count, int la_valid, int use_at,
                    const fs_at_hdr32_u32 *at, const fs_real_hdr32 coords[4], uint32_t n_iterations,
                    int stage_test, int mode, uint32_t *out, uint32_t stride, int threads, uint64_t *stats)
{
    lav2_impl<float>(width, height, y0, y1, orbit, orbit_count, period_maybe_zero, las, n_las, stages, stage_count,
                     la_valid, use_at, at, coords, n_iterations, stage_test, mode, out, stride, threads, stats);
}
void orc_lav2_hdr64(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_orbit_hdr64 *orbit,
                    uint64_t orbit_count, uint64_t period_maybe_zero, const fs_la_hdr64_u32 *las, uint32_t n_las,
                    const fs_la_stage_u32 *stages, uint32_t stage_count, int la_valid, int use_at,
                    const fs_at_hdr64_u32 *at, const fs_real_hdr64 coords[4], uint32_t n_iterations,
                    int stage_test, int mode, uint32_t *out, uint32_t stride, int threads, uint64_t *stats)
{
    lav2_impl<double>(width, height, y0, y1, orbit, orbit_count, period_maybe_zero, las, n_las, stages, stage_count,
                      la_valid, use_at, at, coords, n_iterations, stage_test, mode, out, stride, threads, stats);
}

// IterType = uint64_t (iteration caps of 2^32 and above): the same functions with 64-bit counters and a uint64_t buffer
void orc_lav2_hdr32_u64(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_orbit_hdr32 *orbit,
                        uint64_t orbit_count, uint64_t period_maybe_zero, const fs_la_hdr32_u32 *las, uint32_t n_las,
                        const fs_la_stage_u32 *stages, uint32_t stage_count, int la_valid, int use_at,
                        const fs_at_hdr32_u32 *at, const fs_real_hdr32 coords[4], uint64_t n_iterations, int stage_test,
                        int mode, uint64_t *out, uint32_t stride, int threads, uint64_t *stats)
{
    lav2_impl<float, uint64_t>(width, height, y0, y1, orbit, orbit_count, period_maybe_zero, las, n_las, stages,
                               stage_count, la_valid, use_at, at, coords, n_iterations, stage_test, mode, out, stride,
                               threads, stats);
}
void orc_lav2_hdr64_u64(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, const fs_orbit_hdr64 *orbit,
                        uint64_t orbit_count, uint64_t period_maybe_zero, const fs_la_hdr64_u32 *las, uint32_t n_las,
                        const fs_la_stage_u32 *stages, uint32_t stage_count, int la_valid, int use_at,
                        const fs_at_hdr64_u32 *at, const fs_real_hdr64 coords[4], uint64_t n_iterations, int stage_test,
                        int mode, uint64_t *out, uint32_t stride, int threads, uint64_t *stats)
{
    lav2_impl<double, uint64_t>(width, height, y0, y1, orbit, orbit_count, period_maybe_zero, las, n_las, stages,
                                stage_count, la_valid, use_at, at, coords, n_iterations, stage_test, mode, out, stride,
                                threads, stats);
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// Scaled perturbation, T = HDRFloat<float>: restated CUDA kernel
//   mandel_1x_float_perturb_scaled<uint32_t, HDRFloat<float>>      FractalSharkGpuLib/ScaledKernels.cuh:3-239
// behind RenderAlgorithm GpuHDRx32PerturbedScaled (GPU_Render.cu:1302-1376).  PARITY UNPINNED: the reference has no CPU
// twin of this algorithm and no golden for it, and its binary32 expressions are subject to nvcc's FMA contraction,
// which the source does not determine.  This restatement evaluates every expression in source order with one IEEE
// operation per operator (no contraction), like the HIP kernel it checks.  w2threshold = exp(log(1e30f)/2) is passed in
// (the caller evaluates it once in double precision) instead of relying on a particular libm.
//   HDRFloat pieces used here only: divide_mutable HDRFloat.h:624-636, HdrSqrt :1358-1383, operator T() :557-568,
//   compareToBothPositiveReducedTemplate<256> :1169-1184.
namespace {
inline HT<float> HDiv(HT<float> a, HT<float> b) { return HT<float>{a.m / b.m, clampE(a.e - b.e)}; }
inline HT<float> HSqrt(HT<float> a)
{
    const bool odd = (a.e & 1) != 0;
    return HT<float>{std::sqrt(odd ? 2.0f * a.m : a.m), odd ? (a.e - 1) / 2 : a.e / 2};
}
inline float HToFloat(HT<float> a) { return a.m * getMultiplier<float>(a.e); }
} // namespace

extern "C" void orc_gpu_scaled_hdr32(uint32_t *out, uint32_t pitch, uint32_t width, uint32_t y0, uint32_t y1,
                                     const fs_orbit_hdr32_bad *orbT, const fs_orbit_f32_bad *orbF, uint32_t count,
                                     const fs_real_hdr32 coords[4], uint32_t n_iterations, float w2threshold, int threads,
                                     uint64_t *stats)
{
    using H = HT<float>;
    const H dx = ld(coords[0]), dy = ld(coords[1]), centerX = ld(coords[2]), centerY = ld(coords[3]);
    std::atomic<uint64_t> s_rescale{0}, s_full{0}, s_float{0};
    const uint32_t MaxRefIteration = count - 1;
    auto TX = [&](uint32_t i) { return H{orbT[i].mx, orbT[i].ex}; };
    auto TY = [&](uint32_t i) { return H{orbT[i].my, orbT[i].ey}; };
    const H Two = HFromMant<float>(2.0f);
    run_rows(y0, y1, threads, [&](uint32_t Y) {
        uint64_t c_rescale = 0, c_full = 0, c_float = 0;
        for (uint32_t X = 0; X < width; X++) {
            uint32_t iter = 0, RefIteration = 0;
            H DeltaReal = Sub(MulByScalar<float>(dx, (float)(int)X), centerX);
            Reduce(DeltaReal);
            H DeltaImaginary = Sub(MulByScalar<float>(Neg(dy), (float)(int)Y), centerY);
            Reduce(DeltaImaginary);
            H S = HSqrt(Add(Mul(DeltaReal, DeltaReal), Mul(DeltaImaginary, DeltaImaginary)));
            Reduce(S);
            float DeltaSub0DX = HToFloat(HDiv(DeltaReal, S));
            float DeltaSub0DY = HToFloat(HDiv(DeltaImaginary, S));
            float DeltaSubNWX = 0, DeltaSubNWY = 0;
            float s = HToFloat(S);
            float twos = 2 * s;
            auto rescale = [&](H NewX, H NewY) {
                S = HSqrt(Add(Mul(NewX, NewX), Mul(NewY, NewY)));
                Reduce(S);
                s = HToFloat(S);
                twos = 2 * s;
                DeltaSub0DX = HToFloat(HDiv(DeltaReal, S));
                DeltaSub0DY = HToFloat(HDiv(DeltaImaginary, S));
                DeltaSubNWX = HToFloat(HDiv(NewX, S));
                DeltaSubNWY = HToFloat(HDiv(NewY, S));
            };
            while (iter < n_iterations) {
                if (orbF[RefIteration].bad == 0) {
                    const float fx = orbF[RefIteration].x, fy = orbF[RefIteration].y;
                    const float wx = DeltaSubNWX, wy = DeltaSubNWY;
                    DeltaSubNWX = wx * fx * 2 - wy * fy * 2 + s * wx * wx - s * wy * wy + DeltaSub0DX;
                    DeltaSubNWY = wx * (fy * 2 + twos * wy) + wy * fx * 2 + DeltaSub0DY;
                    c_float++;
                    ++RefIteration;
                    const float tempZX = orbF[RefIteration].x + DeltaSubNWX * s;
                    const float tempZY = orbF[RefIteration].y + DeltaSubNWY * s;
                    const float zn_size = tempZX * tempZX + tempZY * tempZY;
                    const float w2 = DeltaSubNWX * DeltaSubNWX + DeltaSubNWY * DeltaSubNWY;
                    const float normDeltaSubN = w2 * s * s;
                    const bool zn_size_OK = zn_size < 256.0f;
                    const bool test1a = zn_size < normDeltaSubN;
                    const bool test1b = RefIteration == MaxRefIteration;
                    const bool test1ab = test1a || (test1b && zn_size_OK);
                    const bool testw2 = (w2 >= w2threshold) && zn_size_OK;
                    const bool none = !test1ab && !testw2 && zn_size_OK;
                    if (none) {
                        ++iter;
                        continue;
                    } else if (test1ab) {
                        const H ZX = Add(TX(RefIteration), Mul(HFromMant<float>(DeltaSubNWX), S));
                        const H ZY = Add(TY(RefIteration), Mul(HFromMant<float>(DeltaSubNWY), S));
                        RefIteration = 0;
                        rescale(ZX, ZY);
                        c_rescale++;
                        ++iter;
                        continue;
                    } else if (testw2) {
                        rescale(Mul(HFromMant<float>(DeltaSubNWX), S), Mul(HFromMant<float>(DeltaSubNWY), S));
                        c_rescale++;
                        ++iter;
                        continue;
                    } else {
                        break;
                    }
                } else {
                    // full iteration in T
                    const H wx = HFromMant<float>(DeltaSubNWX), wy = HFromMant<float>(DeltaSubNWY);
                    const H cxr = TX(RefIteration), cyr = TY(RefIteration);
                    H nX = Mul(Mul(wx, cxr), Two);
                    nX = Sub(nX, Mul(Mul(wy, cyr), Two));
                    nX = Add(nX, Mul(Mul(S, wx), wx));
                    nX = Sub(nX, Mul(Mul(S, wy), wy));
                    nX = Add(nX, HDiv(DeltaReal, S));
                    Reduce(nX);
                    H nY = Mul(wx, Add(Mul(cyr, Two), Mul(Mul(HFromNumber<float>(2.0f), S), wy)));
                    nY = Add(nY, Mul(Mul(wy, cxr), Two));
                    nY = Add(nY, HDiv(DeltaImaginary, S));
                    Reduce(nY);
                    c_full++;
                    ++RefIteration;
                    const H tempZX = Add(TX(RefIteration), Mul(nX, S));
                    const H tempZY = Add(TY(RefIteration), Mul(nY, S));
                    H zn_size = Add(Mul(tempZX, tempZX), Mul(tempZY, tempZY));
                    Reduce(zn_size);
                    // !HdrCompareToBothPositiveReducedLT<T,256>(zn_size)
                    const int c256 = zn_size.e > 1 ? 1 : (zn_size.e < 1 ? -1 : (zn_size.m >= 256.0f ? 1 : -1));
                    if (!(c256 < 0))
                        break;
                    const H TwoS = Mul(S, S);
                    H normDeltaSubN = Add(Mul(Mul(nX, nX), TwoS), Mul(Mul(nY, nY), TwoS));
                    Reduce(normDeltaSubN);
                    H NewX, NewY;
                    if (CmpPosReduced(zn_size, normDeltaSubN) < 0 || RefIteration == MaxRefIteration) {
                        NewX = Add(TX(RefIteration), Mul(nX, S));
                        NewY = Add(TY(RefIteration), Mul(nY, S));
                        RefIteration = 0;
                    } else {
                        NewX = Mul(nX, S);
                        NewY = Mul(nY, S);
                    }
                    rescale(NewX, NewY);
                }
                ++iter;
            }
            out[(size_t)Y * pitch + X] = iter;
        }
        s_rescale += c_rescale;
        s_full += c_full;
        s_float += c_float;
    });
    if (stats) {
        stats[0] = s_rescale;
        stats[1] = s_full;
        stats[2] = s_float;
    }
}

// The same kernel for T = double (Gpu1x32PerturbedScaled): plain double arithmetic.  PARITY UNPINNED, same conventions.
extern "C" void orc_gpu_scaled_f64(uint32_t *out, uint32_t pitch, uint32_t width, uint32_t y0, uint32_t y1,
                                   const fs_orbit_f64_bad *orbT, const fs_orbit_f32_bad *orbF, uint32_t count,
                                   const double coords[4], uint32_t n_iterations, float w2threshold, int threads,
                                   uint64_t *stats)
{
    const double dx = coords[0], dy = coords[1], centerX = coords[2], centerY = coords[3];
    std::atomic<uint64_t> s_rescale{0}, s_full{0}, s_float{0};
    const uint32_t MaxRefIteration = count - 1;
    run_rows(y0, y1, threads, [&](uint32_t Y) {
        uint64_t c_rescale = 0, c_full = 0, c_float = 0;
        for (uint32_t X = 0; X < width; X++) {
            uint32_t iter = 0, RefIteration = 0;
            const double DeltaReal = dx * (double)(int)X - centerX;
            const double DeltaImaginary = -dy * (double)(int)Y - centerY;
            double S = std::sqrt(DeltaReal * DeltaReal + DeltaImaginary * DeltaImaginary);
            float DeltaSub0DX = (float)(DeltaReal / S), DeltaSub0DY = (float)(DeltaImaginary / S);
            float DeltaSubNWX = 0, DeltaSubNWY = 0;
            float s = (float)S, twos = 2 * s;
            auto rescale = [&](double NewX, double NewY) {
                S = std::sqrt(NewX * NewX + NewY * NewY);
                s = (float)S;
                twos = 2 * s;
                DeltaSub0DX = (float)(DeltaReal / S);
                DeltaSub0DY = (float)(DeltaImaginary / S);
                DeltaSubNWX = (float)(NewX / S);
                DeltaSubNWY = (float)(NewY / S);
            };
            while (iter < n_iterations) {
                if (orbF[RefIteration].bad == 0) {
                    const float fx = orbF[RefIteration].x, fy = orbF[RefIteration].y;
                    const float wx = DeltaSubNWX, wy = DeltaSubNWY;
                    DeltaSubNWX = wx * fx * 2 - wy * fy * 2 + s * wx * wx - s * wy * wy + DeltaSub0DX;
                    DeltaSubNWY = wx * (fy * 2 + twos * wy) + wy * fx * 2 + DeltaSub0DY;
                    c_float++;
                    ++RefIteration;
                    const float tempZX = orbF[RefIteration].x + DeltaSubNWX * s;
                    const float tempZY = orbF[RefIteration].y + DeltaSubNWY * s;
                    const float zn_size = tempZX * tempZX + tempZY * tempZY;
                    const float w2 = DeltaSubNWX * DeltaSubNWX + DeltaSubNWY * DeltaSubNWY;
                    const float normDeltaSubN = w2 * s * s;
                    const bool zn_size_OK = zn_size < 256.0f;
                    const bool test1ab = (zn_size < normDeltaSubN) || ((RefIteration == MaxRefIteration) && zn_size_OK);
                    const bool testw2 = (w2 >= w2threshold) && zn_size_OK;
                    const bool none = !test1ab && !testw2 && zn_size_OK;
                    if (none) {
                        ++iter;
                        continue;
                    } else if (test1ab) {
                        const double ZX = orbT[RefIteration].x + (double)DeltaSubNWX * S;
                        const double ZY = orbT[RefIteration].y + (double)DeltaSubNWY * S;
                        RefIteration = 0;
                        rescale(ZX, ZY);
                        c_rescale++;
                        ++iter;
                        continue;
                    } else if (testw2) {
                        rescale((double)DeltaSubNWX * S, (double)DeltaSubNWY * S);
                        c_rescale++;
                        ++iter;
                        continue;
                    } else {
                        break;
                    }
                } else {
                    const double wx = (double)DeltaSubNWX, wy = (double)DeltaSubNWY;
                    const double cxr = orbT[RefIteration].x, cyr = orbT[RefIteration].y;
                    double nX = wx * cxr * 2;
                    nX -= wy * cyr * 2;
                    nX += S * wx * wx;
                    nX -= S * wy * wy;
                    nX += DeltaReal / S;
                    double nY = wx * (cyr * 2 + 2.0 * S * wy);
                    nY += wy * cxr * 2;
                    nY += DeltaImaginary / S;
                    c_full++;
                    ++RefIteration;
                    const double tempZX = orbT[RefIteration].x + nX * S;
                    const double tempZY = orbT[RefIteration].y + nY * S;
                    const double zn_size = tempZX * tempZX + tempZY * tempZY;
                    if (!(zn_size < 256.0))
                        break;
                    const double TwoS = S * S;
                    const double normDeltaSubN = nX * nX * TwoS + nY * nY * TwoS;
                    double NewX, NewY;
                    if (zn_size < normDeltaSubN || RefIteration == MaxRefIteration) {
                        NewX = orbT[RefIteration].x + nX * S;
                        NewY = orbT[RefIteration].y + nY * S;
                        RefIteration = 0;
                    } else {
                        NewX = nX * S;
                        NewY = nY * S;
                    }
                    rescale(NewX, NewY);
                }
                ++iter;
            }
            out[(size_t)Y * pitch + X] = iter;
        }
        s_rescale += c_rescale;
        s_full += c_full;
        s_float += c_float;
    });
    if (stats) {
        stats[0] = s_rescale;
        stats[1] = s_full;
        stats[2] = s_float;
    }
}
