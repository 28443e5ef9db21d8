// refinputs.cpp -- host-side builders for the *inputs* of the per-pixel pass (C ABI: include/fs_inputs.h).
//
// The MI355X renderer consumes what FractalShark's host code already produces: a view (bounding box ->
// dx/dy/centre), a reference orbit (PerturbationResults layout) and an LAv2 table (LAReference layout).
// On a FractalShark host these come from Fractal.cpp / RefOrbitCalc.cpp / LAReference.cpp; this file is
// the stand-alone equivalent used by bench.py, the tests and the oracle harness, so that the whole
// pipeline can run where FractalShark itself is not built.  It is upstream of the hot path (SURVEY.md
// section 8(f) rows 1-3), host-only, single-threaded, and follows the reference call-for-call on GMP's
// mpf_t so the extracted float/double values are the reference's:
//
//   view      FractalViewPresets.cpp GetViewPreset, PointZoomBBConverter.cpp:26-53,100-117,271-330,
//             PrecisionCalculator.cpp:58-108, Fractal.cpp:264-307,591-629
//   coords    Fractal.cpp:2118-2119 (direct), :2230-2238 / :2513-2521 (perturbation)
//   orbit     RefOrbitCalc.cpp:244-249,423-647 (ST / STPeriodicity), PerturbationResults.cpp:812-884
//   LA table  LAReference.cpp:28-210 (stage 0, single-threaded), :215-770 (stage 0, multi-threaded
//             hand-off, replayed sequentially), :774-966 (higher stages), :971-1013, :1050-1074 (AT);
//             LAInfoDeep.h:108-391,456-506; LAParameters.h:66-75
#include <gmp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../csrc/hdr_math.hpp"
#include "../csrc/df32_math.hpp"
#include "../csrc/la_math.hpp"
#include "../csrc/bla_math.hpp"
#include "../../include/fs_inputs.h"

using namespace fs;
using namespace fs::la;

namespace {

// ------------------------------------------------------------------ mpf wrapper (HighPrecision.h)
// Mirrors HighPrecisionT: values carry their own precision; binary operators produce a result with the
// precision of the left operand (HighPrecision.h:262-300); default-constructed values use
// mpf_get_default_prec().
struct Mp {
    mpf_t v;
    Mp() { mpf_init(v); }
    explicit Mp(uint64_t prec_bits, int) { mpf_init2(v, prec_bits); }
    Mp(const Mp &o)
    {
        mpf_init2(v, mpf_get_prec(o.v));
        mpf_set(v, o.v);
    }
    Mp &operator=(const Mp &o)
    {
        if (this != &o) {
            mpf_set_prec(v, mpf_get_prec(o.v));
            mpf_set(v, o.v);
        }
        return *this;
    }
    ~Mp() { mpf_clear(v); }
    static Mp from_str(const char *s)
    {
        Mp r;
        mpf_set_str(r.v, s, 10);
        return r;
    }
    static Mp from_ui(uint64_t x)
    {
        Mp r;
        mpf_set_ui(r.v, x);
        return r;
    }
    uint64_t prec() const { return (uint32_t)mpf_get_prec(v); }
    void set_prec(uint64_t p) { mpf_set_prec(v, p); }
};

Mp operator+(const Mp &a, const Mp &b)
{
    Mp r(a.prec(), 0);
    mpf_add(r.v, a.v, b.v);
    return r;
}
Mp operator-(const Mp &a, const Mp &b)
{
    Mp r(a.prec(), 0);
    mpf_sub(r.v, a.v, b.v);
    return r;
}
Mp operator/(const Mp &a, const Mp &b)
{
    Mp r(a.prec(), 0);
    mpf_div(r.v, a.v, b.v);
    return r;
}
Mp mp_abs(const Mp &a)
{
    Mp r(a.prec(), 0);
    mpf_abs(r.v, a.v);
    return r;
}

// HDRFloat(const mpf_t) -- HDRFloat.h:366-389: mantissa in [0.5,1) as returned by mpf_get_d_2exp, cast
// to F; NOT reduced.  Zero -> default.
template <class F> hreal<F> hr_from_mpf(const mpf_t x)
{
    if (mpf_cmp_ui(x, 0) == 0)
        return hr_zero<F>();
    long e;
    const double d = mpf_get_d_2exp(&e, x);
    return hreal<F>{(F)d, (int32_t)e};
}

} // namespace

// ------------------------------------------------------------------ view
struct fsh_view {
    Mp minX, minY, maxX, maxY, ptX, ptY, zoom;
    uint64_t prec_bits = 0;
    uint32_t width = 0, height = 0;
};

namespace {

// PointZoomBBConverter::SquareAspectRatio, PointZoomBBConverter.cpp:271-330.
void square_aspect(fsh_view &vw, size_t scrnWidth, size_t scrnHeight)
{
    if (scrnWidth == 0 || scrnHeight == 0)
        return;
    const uint64_t prec = vw.ptX.prec();
    Mp ratio(prec, 0), mwidth(prec, 0), height(prec, 0), tmp(prec, 0);
    {
        Mp w = Mp::from_ui(scrnWidth);
        Mp h = Mp::from_ui(scrnHeight);
        mpf_div(ratio.v, w.v, h.v);
    }
    mpf_sub(mwidth.v, vw.maxX.v, vw.minX.v);
    mpf_div(mwidth.v, mwidth.v, ratio.v);
    mpf_sub(height.v, vw.maxY.v, vw.minY.v);

    if (mpf_cmp(height.v, mwidth.v) > 0) {
        mpf_sub(tmp.v, height.v, mwidth.v);
        mpf_mul(tmp.v, ratio.v, tmp.v);
        mpf_div_ui(tmp.v, tmp.v, 2);
        mpf_sub(vw.minX.v, vw.minX.v, tmp.v);
        mpf_add(vw.maxX.v, vw.maxX.v, tmp.v);
    } else if (mpf_cmp(height.v, mwidth.v) < 0) {
        mpf_sub(tmp.v, mwidth.v, height.v);
        mpf_div_ui(tmp.v, tmp.v, 2);
        mpf_sub(vw.minY.v, vw.minY.v, tmp.v);
        mpf_add(vw.maxY.v, vw.maxY.v, tmp.v);
    }
    mpf_add(vw.ptX.v, vw.minX.v, vw.maxX.v);
    mpf_div_ui(vw.ptX.v, vw.ptX.v, 2);
    mpf_add(vw.ptY.v, vw.minY.v, vw.maxY.v);
    mpf_div_ui(vw.ptY.v, vw.ptY.v, 2);

    Mp deltaY(prec, 0);
    mpf_sub(deltaY.v, vw.maxY.v, vw.minY.v);
    if (mpf_cmp_ui(deltaY.v, 0) == 0) {
        vw.zoom = Mp::from_ui(1);
    } else {
        Mp two = Mp::from_ui(2);
        mpf_div(vw.zoom.v, two.v, deltaY.v);
        mpf_mul_ui(vw.zoom.v, vw.zoom.v, 2);
    }
}

} // namespace

namespace {
fsh_view *finish_view(std::unique_ptr<fsh_view> vw, bool from_box);
}

extern "C" fsh_view *fsh_view_create(const char *minX, const char *minY, const char *maxX, const char *maxY,
                                     uint32_t width, uint32_t height)
{
    // GetViewPreset: coordinates are parsed at 1,000,000 bits (FractalViewPresets.cpp:11-12).
    mpf_set_default_prec(1000000);
    auto vw = std::make_unique<fsh_view>();
    vw->width = width;
    vw->height = height;
    vw->minX = Mp::from_str(minX);
    vw->minY = Mp::from_str(minY);
    vw->maxX = Mp::from_str(maxX);
    vw->maxY = Mp::from_str(maxY);
    return finish_view(std::move(vw), true);
}

namespace {

// The rest of what the reference does to a new view: the box form of PointZoomBBConverter (from_box; the point + zoom form
// has set ptX / ptY / zoom already), Fractal::SetPrecision, SquareAspectRatio.
fsh_view *finish_view(std::unique_ptr<fsh_view> vw, bool from_box)
{
    const uint32_t width = vw->width, height = vw->height;
    // PointZoomBBConverter(minX,minY,maxX,maxY), PointZoomBBConverter.cpp:26-53
    if (from_box) {
        Mp two = Mp::from_ui(2);
        vw->ptX = (vw->minX + vw->maxX) / two;
        vw->ptY = (vw->minY + vw->maxY) / two;
        Mp deltaY = vw->maxY - vw->minY;
        if (mpf_cmp_ui(deltaY.v, 0) == 0) {
            vw->zoom = Mp::from_ui(1);
        } else {
            Mp zf = Mp::from_ui(2) / deltaY;
            Mp zf2(zf.prec(), 0);
            mpf_mul(zf2.v, zf.v, two.v);
            vw->zoom = zf2;
        }
    }
    // Fractal::SetPrecision -> PrecisionCalculator::GetPrecision (PrecisionCalculator.cpp:21-107);
    // RequiresReuse() is false for the ST algorithms this file implements.
    {
        Mp dX = mp_abs(vw->maxX - vw->minX);
        Mp dY = mp_abs(vw->maxY - vw->minY);
        const hreal<double> tx = hr_from_mpf<double>(dX.v);
        const hreal<double> ty = hr_from_mpf<double>(dY.v);
        uint64_t larger = (uint64_t)std::max(std::abs(tx.e), std::abs(ty.e));
        larger += 120; // AuthoritativeMinExtraPrecisionInBits, HighPrecision.h
        vw->prec_bits = larger;
    }
    // PointZoomBBConverter::SetPrecision, PointZoomBBConverter.cpp:100-117
    mpf_set_default_prec(vw->prec_bits);
    vw->minX.set_prec(vw->prec_bits);
    vw->minY.set_prec(vw->prec_bits);
    vw->maxX.set_prec(vw->prec_bits);
    vw->maxY.set_prec(vw->prec_bits);
    vw->ptX.set_prec(vw->prec_bits);
    vw->ptY.set_prec(vw->prec_bits);
    vw->zoom.set_prec(vw->prec_bits);
    square_aspect(*vw, width, height);
    return vw.release();
}

} // namespace

// ------------------------------------------------------------------ Imagina ".im" files, location form
// The reference saves / loads a view as an Imagina location file (RefOrbitCalc::SaveOrbitResults(filename),
// RefOrbitCalc.cpp:3117-3166; LoadOrbitConstInternal, :3425-3520; ImaginaOrbit.h:10-24):
//   IMFileHeader { u64 Magic = 0x000A0D56504D49FF ("\xFFIMPV\r\n\0"; "Sharks:)" = 0x536861726b733a29 for the reference's own
//                  variant), u64 Reserved = 0, u64 LocationOffset = 32, u64 ReferenceOffset = 0 (no orbit stored) }
//   at LocationOffset: HRReal halfH  = HDRFloat<double, Left, int64_t>{ double mantissa; int64 exp } = half the view's height
//                      u64 iterationLimit
//                      mpf orbitX, mpf orbitY  (the view's centre) in the MPIR raw stream form of MpirSerialization.cpp:157-187:
//                          long _mp_exp (limbs), then the limbs as one integer: int32 big-endian signed byte count +
//                          magnitude bytes, most significant first
// Files that also carry a reference orbit (ReferenceOffset != 0: the reference's MaxCompression intermediate form) are
// recognised and loaded as locations; the stored orbit is ignored (it is not needed to render: the orbit is recomputed).
namespace {

constexpr uint64_t kImMagic = 0x000A0D56504D49FFull, kSharksMagic = 0x536861726b733a29ull;
// precision a location file may ask for (bits): the view must still be expressible with int32 exponents, and GMP aborts
// (it does not throw) when asked for more than it can allocate
constexpr uint64_t kMaxImPrecisionBits = 1ull << 26;
// orbit entries a file may announce (32 bytes each while it is expanded on the host: 16 GiB)
constexpr uint64_t kMaxImOrbitEntries = 1ull << 29;

// exp_bytes = sizeof(long) of the build that wrote the file: 4 for the reference's Windows (MSVC) build and Imagina itself,
// 8 for the reference built on Linux.  Limbs are 64-bit in both (MPIR x64 / GMP), so only the field width differs.
// The integer stream under the mpf one: MpirSerialization::mpz_out_raw_stream / mpz_inp_raw_stream (MPIR's mpz_out_raw):
// a 4-byte big-endian signed header sign x byte-count, then the magnitude, most significant byte first.  Pinned by the
// reference's own byte vectors (FractalSharkTest/TestMpirSerialization.cpp:236-308, :393-428; tests/test_mpir_wire_format.py).
size_t im_write_mpz(FILE *f, mpz_srcptr Z)
{
    size_t byte_count = 0;
    const int sign = mpz_sgn(Z);
    if (sign != 0)
        byte_count = (mpz_sizeinbase(Z, 2) + 7) / 8;
    const int32_t header = sign == 0 ? 0 : (sign < 0 ? -1 : 1) * (int32_t)byte_count;
    const uint32_t u = (uint32_t)header;
    const unsigned char hdr[4] = {(unsigned char)(u >> 24), (unsigned char)(u >> 16), (unsigned char)(u >> 8), (unsigned char)u};
    fwrite(hdr, 1, 4, f);
    if (byte_count) {
        std::vector<unsigned char> buf(byte_count);
        size_t n = 0;
        mpz_export(buf.data(), &n, 1, 1, 1, 0, Z);
        fwrite(buf.data(), 1, byte_count, f);
    }
    return 4 + byte_count;
}

// Z must be initialised.  *nonzero: whether the header announced a magnitude.
bool im_read_mpz(FILE *f, mpz_ptr Z, bool *nonzero)
{
    unsigned char hdr[4];
    if (fread(hdr, 1, 4, f) != 4)
        return false;
    const int32_t raw = (int32_t)(((uint32_t)hdr[0] << 24) | ((uint32_t)hdr[1] << 16) | ((uint32_t)hdr[2] << 8) | hdr[3]);
    mpz_set_ui(Z, 0);
    if (raw != 0) {
        const size_t n = (size_t)std::abs((int64_t)raw);
        if (n > (1u << 24)) // 128 Mbit of mantissa: not a location anyone saved; the wrong exponent width reads such counts
            return false;
        std::vector<unsigned char> buf(n);
        if (fread(buf.data(), 1, n, f) != n)
            return false;
        mpz_import(Z, n, 1, 1, 1, 0, buf.data());
        if (raw < 0)
            mpz_neg(Z, Z);
    }
    if (nonzero)
        *nonzero = raw != 0;
    return true;
}

void im_write_mpf(FILE *f, mpf_srcptr X, int exp_bytes)
{
    const int64_t expt = X->_mp_exp;
    if (exp_bytes == 4) {
        const int32_t e32 = (int32_t)expt;
        fwrite(&e32, 4, 1, f);
    } else {
        fwrite(&expt, 8, 1, f);
    }
    mpz_t Z; // non-owning view of X's limbs, like the reference
    const int nz = X->_mp_size;
    Z->_mp_alloc = std::abs(nz);
    Z->_mp_size = nz;
    Z->_mp_d = X->_mp_d;
    im_write_mpz(f, Z);
}

bool im_read_mpf(FILE *f, mpf_ptr X, int exp_bytes)
{
    int64_t expt = 0;
    if (exp_bytes == 4) {
        int32_t e32;
        if (fread(&e32, 4, 1, f) != 1)
            return false;
        expt = e32;
    } else if (fread(&expt, 8, 1, f) != 1) {
        return false;
    }
    mpz_t Z;
    mpz_init(Z);
    bool nonzero = false;
    if (!im_read_mpz(f, Z, &nonzero)) {
        mpz_clear(Z);
        return false;
    }
    mpf_set_z(X, Z);
    if (nonzero)
        X->_mp_exp = (mp_exp_t)expt;
    mpz_clear(Z);
    return true;
}

struct ImHalfH { // Imagina::HRReal
    double mantissa;
    int64_t exp;
};

} // namespace

// The integer stream on its own, memory to memory (tests: the reference's wire-format vectors).  fsh_mpz_raw_write: the
// bytes of `value` (a number in `base`) into out[0 .. cap), returns their count (0: bad number or no room).
// fsh_mpz_raw_read: one integer from in[0 .. n), its decimal text into out_decimal; returns the bytes consumed (0: error).
// The LAParameters every table in this file is built with (la_math.hpp LAParams = the reference's defaults,
// LAParameters.h:66-75): {detectionMethod, LAThresholdScaleExp, LAThresholdCScaleExp, Stage0PeriodDetectionThreshold2Exp,
// PeriodDetectionThreshold2Exp, Stage0PeriodDetectionThresholdExp, PeriodDetectionThresholdExp}.
extern "C" void fsh_la_default_params(int32_t out[7])
{
    const LAParams p{};
    out[0] = p.detectionMethod, out[1] = p.laThresholdScaleExp, out[2] = p.laThresholdCScaleExp;
    out[3] = p.stage0PeriodDetectionThreshold2Exp, out[4] = p.periodDetectionThreshold2Exp;
    out[5] = p.stage0PeriodDetectionThresholdExp, out[6] = p.periodDetectionThresholdExp;
}

extern "C" size_t fsh_mpz_raw_write(const char *value, int base, unsigned char *out, size_t cap)
{
    mpz_t z;
    if (mpz_init_set_str(z, value, base) != 0) {
        mpz_clear(z);
        return 0;
    }
    char *mem = nullptr;
    size_t len = 0;
    FILE *f = open_memstream(&mem, &len);
    size_t written = 0;
    if (f) {
        written = im_write_mpz(f, z);
        fclose(f);
        if (written != len || len > cap)
            written = 0;
        else
            memcpy(out, mem, len);
        free(mem);
    }
    mpz_clear(z);
    return written;
}
extern "C" size_t fsh_mpz_raw_read(const unsigned char *in, size_t n, char *out_decimal, size_t cap)
{
    if (n == 0)
        return 0;
    FILE *f = fmemopen(const_cast<unsigned char *>(in), n, "rb");
    if (!f)
        return 0;
    mpz_t z;
    mpz_init(z);
    size_t used = 0;
    if (im_read_mpz(f, z, nullptr) && mpz_sizeinbase(z, 10) + 2 <= cap) {
        mpz_get_str(out_decimal, 10, z);
        used = (size_t)ftell(f);
    }
    mpz_clear(z);
    fclose(f);
    return used;
}

extern "C" int fsh_view_save_im(const fsh_view *v, uint64_t iteration_limit, const char *path, int exp_bytes)
{
    if (exp_bytes != 4 && exp_bytes != 8)
        return -1;
    FILE *f = fopen(path, "wb");
    if (!f)
        return -1;
    const uint64_t header[4] = {kImMagic, 0, 32, 0};
    fwrite(header, 8, 4, f);
    // radiusY = double{maxY - minY} / 2.0; halfH = HRReal{radiusY}: the templated HDRFloat(number) constructor
    // (HDRFloat.h:295-363): zero -> {0, MIN_BIG_EXPONENT}, else mantissa in [1, 2) and the exponent beside it
    mpf_set_default_prec(v->prec_bits);
    Mp dY = v->maxY - v->minY;
    const double radiusY = mpf_get_d(dY.v) / 2.0;
    ImHalfH hh;
    if (radiusY == 0.0) {
        hh.mantissa = 0.0;
        hh.exp = INT16_MIN >> 3; // GenericHdrBase::MIN_BIG_EXPONENT for a TExp that is neither int32_t nor float
    } else {
        // the same bit surgery (HDRFloat.h:307-316), so that subnormal radii come out the way the reference writes them
        uint64_t bits;
        memcpy(&bits, &radiusY, 8);
        const uint64_t val = (bits & 0x800FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
        memcpy(&hh.mantissa, &val, 8);
        hh.exp = (int64_t)((bits & 0x7FF0000000000000ull) >> 52) - 1023;
    }
    fwrite(&hh, sizeof(hh), 1, f);
    fwrite(&iteration_limit, 8, 1, f);
    im_write_mpf(f, v->ptX.v, exp_bytes);
    im_write_mpf(f, v->ptY.v, exp_bytes);
    const bool ok = !ferror(f);
    fclose(f);
    return ok ? 0 : -1;
}

extern "C" fsh_view *fsh_view_load_im(const char *path, uint32_t width, uint32_t height, uint64_t *iteration_limit,
                                     int *has_orbit, int *exp_bytes_out)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return nullptr;
    uint64_t header[4];
    ImHalfH hh;
    uint64_t limit = 0;
    fsh_view *out = nullptr;
    if (fread(header, 8, 4, f) == 4 && (header[0] == kImMagic || header[0] == kSharksMagic) &&
        fseek(f, (long)header[2], SEEK_SET) == 0 && fread(&hh, sizeof(hh), 1, f) == 1 && fread(&limit, 8, 1, f) == 1) {
        // precision = -min(0, halfH.exp) + AuthoritativeMinExtraPrecisionInBits (RefOrbitCalc.cpp:3458-3460)
        if (hh.exp < -(int64_t)kMaxImPrecisionBits) {
            fclose(f);
            return nullptr;
        }
        const uint64_t precision = (uint64_t)(-std::min<int64_t>(0, hh.exp)) + 120u;
        mpf_set_default_prec(precision);
        Mp X(precision, 0), Y(precision, 0), H(precision, 0);
        // which `long` wrote the file: the two values must end exactly where the location section ends (the reference
        // orbit's offset, or the end of the file)
        const long at = ftell(f);
        fseek(f, 0, SEEK_END);
        const long section_end = header[3] > header[2] ? (long)header[3] : ftell(f);
        int width_ok = 0;
        for (int exp_bytes : {4, 8}) {
            fseek(f, at, SEEK_SET);
            if (im_read_mpf(f, X.v, exp_bytes) && im_read_mpf(f, Y.v, exp_bytes) && ftell(f) == section_end) {
                width_ok = exp_bytes;
                break;
            }
        }
        // halfH == 0 is what the writer stores for a view too deep for a double radius; there is no box to load
        // (the reference divides by it)
        if (width_ok && hh.mantissa != 0.0) {
            // halfH.GetHighPrecision (HDRFloat.h:396-412); the box is pt -+ Factor / zoomFactor = pt -+ halfH
            mpf_set_d(H.v, hh.mantissa);
            if (hh.exp >= 0)
                mpf_mul_2exp(H.v, H.v, (mp_bitcnt_t)hh.exp);
            else
                mpf_div_2exp(H.v, H.v, (mp_bitcnt_t)(-hh.exp));
            // zoomFactor = 2 / halfH (:3462-3465), then PointZoomBBConverter(ptX, ptY, zoomFactor): the box is
            // pt -+ Factor / zoomFactor (PointZoomBBConverter.cpp:9-24), every operation rounded at `precision`
            const Mp two = Mp::from_ui(2);
            const Mp zoom = two / H;
            const Mp r = two / zoom;
            auto vw = std::make_unique<fsh_view>();
            vw->width = width;
            vw->height = height;
            vw->ptX = X;
            vw->ptY = Y;
            vw->zoom = zoom;
            vw->minX = X - r;
            vw->minY = Y - r;
            vw->maxX = X + r;
            vw->maxY = Y + r;
            out = finish_view(std::move(vw), false);
            if (iteration_limit)
                *iteration_limit = limit;
            if (has_orbit)
                *has_orbit = header[3] != 0;
            if (exp_bytes_out)
                *exp_bytes_out = width_ok;
        }
    }
    fclose(f);
    return out;
}

extern "C" void fsh_view_destroy(fsh_view *v) { delete v; }
extern "C" uint64_t fsh_view_precision_bits(const fsh_view *v) { return v->prec_bits; }

extern "C" int fsh_view_bbox_str(const fsh_view *v, int which, char *buf, size_t buflen)
{
    const Mp *p = which == 0 ? &v->minX : which == 1 ? &v->minY : which == 2 ? &v->maxX : &v->maxY;
    return gmp_snprintf(buf, buflen, "%.Fe", p->v);
}

// Cpu64 / direct kernels: dx, dy, minX, maxY as doubles -- Fractal.cpp:2118-2119,2148-2151
// (T(HighPrecision) for T=double is mpf_get_d, HighPrecision.h:497-501).
extern "C" void fsh_view_coords_direct_f64(const fsh_view *v, uint32_t w_aa, uint32_t h_aa, double out[4])
{
    mpf_set_default_prec(v->prec_bits);
    Mp dx = (v->maxX - v->minX) / Mp::from_ui(w_aa);
    Mp dy = (v->maxY - v->minY) / Mp::from_ui(h_aa);
    out[0] = mpf_get_d(dx.v);
    out[1] = mpf_get_d(dy.v);
    out[2] = mpf_get_d(v->minX.v);
    out[3] = mpf_get_d(v->maxY.v);
}

// Gpu1x32 / Gpu2x32 / Gpu2x64 (Fractal::CalcGpuFractal<IterType, float | MattDblflt | MattDbldbl>, Fractal.cpp:1896-1915):
// FillGpuCoords (:1833-1844) = {MinX, MinY, dx, dy} through FillCoord: float = (float)mpf_get_d (:1813-1818, Convert,
// HighPrecision.h:516-519); MattDblflt = head (float)src, tail (float)(src - HighPrecision{head}) (:1805-1811);
// MattDbldbl the same in double (:1774-1780).  kind: 0 = float[4], 1 = float[8], 2 = double[8], order cx, cy, dx, dy.
extern "C" void fsh_view_coords_direct_lp(const fsh_view *v, uint32_t w_aa, uint32_t h_aa, int kind, void *out)
{
    mpf_set_default_prec(v->prec_bits);
    const Mp dx = (v->maxX - v->minX) / Mp::from_ui(w_aa);
    const Mp dy = (v->maxY - v->minY) / Mp::from_ui(h_aa);
    const Mp *src[4] = {&v->minX, &v->minY, &dx, &dy};
    for (int i = 0; i < 4; i++) {
        if (kind == 0) {
            ((float *)out)[i] = (float)mpf_get_d(src[i]->v);
        } else if (kind == 1) {
            const float head = (float)mpf_get_d(src[i]->v);
            Mp h;
            mpf_set_d(h.v, (double)head);
            const Mp rest = *src[i] - h;
            ((float *)out)[2 * i] = head;
            ((float *)out)[2 * i + 1] = (float)mpf_get_d(rest.v);
        } else if (kind == 2) {
            const double head = mpf_get_d(src[i]->v);
            Mp h;
            mpf_set_d(h.v, head);
            const Mp rest = *src[i] - h;
            ((double *)out)[2 * i] = head;
            ((double *)out)[2 * i + 1] = mpf_get_d(rest.v);
        } else {
            // FillCoord(MattQFltflt / MattQDbldbl), Fractal.cpp:1751-1771: each component converts what is left after
            // subtracting the previous ones (left to right, in the precision of the source)
            Mp rest = *src[i];
            for (int k = 0; k < 4; k++) {
                const double part = kind == 3 ? (double)(float)mpf_get_d(rest.v) : mpf_get_d(rest.v);
                if (kind == 3)
                    ((float *)out)[4 * i + k] = (float)part;
                else
                    ((double *)out)[4 * i + k] = part;
                Mp h;
                mpf_set_d(h.v, part);
                rest = rest - h;
            }
        }
    }
}

// CpuHDR32 / CpuHDR64 (CalcCpuHDR<.., HDRFloat<F>, F>): dx, dy, minX, maxY as HDRFloat built from mpf
// (Fractal.cpp:2118-2119,2148-2151): mantissa in [0.5,1), NOT reduced.
template <class F> static void direct_hdr_coords(const fsh_view &v, uint32_t w_aa, uint32_t h_aa, hreal<F> out[4])
{
    mpf_set_default_prec(v.prec_bits);
    Mp dx = (v.maxX - v.minX) / Mp::from_ui(w_aa);
    Mp dy = (v.maxY - v.minY) / Mp::from_ui(h_aa);
    out[0] = hr_from_mpf<F>(dx.v);
    out[1] = hr_from_mpf<F>(dy.v);
    out[2] = hr_from_mpf<F>(v.minX.v);
    out[3] = hr_from_mpf<F>(v.maxY.v);
}
extern "C" void fsh_view_coords_direct_hdr32(const fsh_view *v, uint32_t w_aa, uint32_t h_aa, fs_real_hdr32 out[4])
{
    hreal<float> t[4];
    direct_hdr_coords<float>(*v, w_aa, h_aa, t);
    for (int i = 0; i < 4; i++)
        out[i] = fs_real_hdr32{t[i].m, t[i].e};
}
extern "C" void fsh_view_coords_direct_hdr64(const fsh_view *v, uint32_t w_aa, uint32_t h_aa, fs_real_hdr64 out[4])
{
    hreal<double> t[4];
    direct_hdr_coords<double>(*v, w_aa, h_aa, t);
    for (int i = 0; i < 4; i++)
        out[i] = fs_real_hdr64{t[i].m, t[i].e, 0};
}

// ------------------------------------------------------------------ reference orbit
template <class F> struct OrbitT {
    std::vector<hreal<F>> x, y; // entry 0 is the explicit zero entry
    uint64_t period = 0;
    hreal<F> maxRadius{};
    hreal<F> orbitXLow{}, orbitYLow{};
    Mp cx, cy; // reference point (m_OrbitX / m_OrbitY)
    uint64_t prec_bits = 0;
    // PerturbExtras::Bad flag per entry (RefOrbitCalc.cpp:550-562,625-627); computed for every orbit, only the scaled
    // kernels read it
    std::vector<uint8_t> bad;
    // packed copies in the ABI layout, built lazily
    std::vector<fs_orbit_hdr32> packed32;
    std::vector<fs_orbit_hdr64> packed64;
    std::vector<fs_orbit_hdr32_bad> packed32_bad;
    std::vector<fs_orbit_f32_bad> packed_f32_bad;
    // PerturbExtras::SimpleCompression: waypoints (m_FullOrbit of the compressed PerturbationResults); x / y above
    // then hold the orbit as RuntimeDecompressor reproduces it.
    bool compressed = false;
    std::vector<uint64_t> wp_index;
    std::vector<hreal<F>> wp_x, wp_y;
    std::vector<fs_orbit_hdr32_rc> packed_rc32;
    std::vector<fs_orbit_hdr64_rc> packed_rc64;
};

struct fsh_orbit {
    int is64 = 0;
    OrbitT<float> f;
    OrbitT<double> d;
};

namespace {

// AddPerturbationReferencePointST<..., Periodicity, ...>, RefOrbitCalc.cpp:423-647, for T = HDRFloat<F>,
// PerturbExtras::Disable, ReuseMode::DontSaveForReuse.
// RuntimeDecompressor::GetCompressedComplex's runOneIter, PerturbationResultsHelpers.h:51-58 (== the compressor's own
// advance, PerturbationResults.cpp:2374-2378).
template <class F> void rc_one_iter(hreal<F> &zx, hreal<F> &zy, hreal<F> cxLow, hreal<F> cyLow)
{
    const hreal<F> zx_old = zx;
    zx = hr_add(hr_sub(hr_mul(zx, zx), hr_mul(zy, zy)), cxLow);
    hr_reduce(zx);
    zy = hr_add(hr_mul(hr_mul(hr_from_number<F>(F(2)), zx_old), zy), cyLow);
    hr_reduce(zy);
}

// compression_exp < 0: PerturbExtras::Disable.  Otherwise PerturbExtras::SimpleCompression with
// CompressionError = T(10^compression_exp) (RefOrbitCompressor, PerturbationResults.cpp:2334-2381; default exponent 20,
// Fractal.h:138-141).
template <class F>
void build_orbit(const fsh_view &vw, uint64_t max_iter, bool periodicity, OrbitT<F> &ob, int compression_exp = -1)
{
    mpf_set_default_prec(vw.prec_bits);
    ob.prec_bits = vw.prec_bits;
    // RefOrbitCalc.cpp:244-249: guess = bbox midpoint
    {
        Mp two = Mp::from_ui(2);
        ob.cx = (vw.maxX + vw.minX) / two;
        ob.cy = (vw.maxY + vw.minY) / two;
    }
    // PerturbationResults::InitResults, PerturbationResults.cpp:812-884
    {
        Mp delta = vw.maxY - vw.minY;
        const hreal<F> radiusY = hr_div(hr_from_mpf<F>(delta.v), hr_from_mant<F>(F(2)));
        ob.maxRadius = hr_reduced(radiusY);
        ob.orbitXLow = hr_from_mpf<F>(ob.cx.v);
        ob.orbitYLow = hr_from_mpf<F>(ob.cy.v);
    }
    ob.x.clear();
    ob.y.clear();
    ob.x.push_back(hr_zero<F>());
    ob.y.push_back(hr_zero<F>());
    ob.bad.assign(1, 0);
    const hreal<F> small_float = hr_from_mant<F>((F)1.1754944e-38); // T((SubType)1.1754944e-38), :472
    const F glitch = (F)0.0000001;                                  // :474
    ob.period = 0;
    ob.compressed = compression_exp >= 0;
    ob.wp_index.assign(1, 0);
    ob.wp_x.assign(1, hr_zero<F>());
    ob.wp_y.assign(1, hr_zero<F>());
    uint64_t count = 1; // m_UncompressedItersInOrbit
    hreal<F> rc_zx = ob.orbitXLow, rc_zy = ob.orbitYLow;
    const hreal<F> rc_err = hr_from_number<F>((F)std::pow(10.0, compression_exp));

    mpf_t cx, cy, zx, zy, zx2, t1, t2;
    mpf_init(cx);
    mpf_set(cx, ob.cx.v);
    mpf_init(cy);
    mpf_set(cy, ob.cy.v);
    mpf_init(zx);
    mpf_init(zy);
    mpf_init(zx2);
    mpf_init(t1);
    mpf_init(t2);

    hreal<F> dzdcX = hr_from_number<F>(F(1));
    hreal<F> dzdcY = hr_from_number<F>(F(0));
    hreal<F> cx_cast, cy_cast;
    {
        long e;
        double m = mpf_get_d_2exp(&e, cx);
        cx_cast = hr_raw<F>((int32_t)e, (F)m);
        m = mpf_get_d_2exp(&e, cy);
        cy_cast = hr_raw<F>((int32_t)e, (F)m);
    }
    const hreal<F> HighOne = hr_from_number<F>(F(1));
    const hreal<F> HighTwo = hr_from_number<F>(F(2));
    const hreal<F> TwoFiftySix = hr_from_number<F>(F(256));

    mpf_set(zx, cx);
    mpf_set(zy, cy);

    for (uint64_t i = 0; i < max_iter; i++) {
        mpf_mul_2exp(zx2, zx, 1);
        hreal<F> double_zx = hr_from_mpf<F>(zx);
        hreal<F> double_zy = hr_from_mpf<F>(zy);
        if (!ob.compressed) {
            ob.x.push_back(double_zx);
            ob.y.push_back(double_zy);
        } else {
            // MaybeAddCompressedIteration({double_zx, double_zy, i + 1})
            const hreal<F> errX = hr_sub(rc_zx, double_zx);
            const hreal<F> errY = hr_sub(rc_zy, double_zy);
            const hreal<F> norm_z = hr_reduced(hr_add(hr_mul(double_zx, double_zx), hr_mul(double_zy, double_zy)));
            const hreal<F> err = hr_reduced(hr_mul(hr_add(hr_mul(errX, errX), hr_mul(errY, errY)), rc_err));
            if (hr_cmp_pos(err, norm_z) >= 0) {
                ob.wp_index.push_back(i + 1);
                ob.wp_x.push_back(double_zx);
                ob.wp_y.push_back(double_zy);
                rc_zx = double_zx;
                rc_zy = double_zy;
            }
            rc_one_iter(rc_zx, rc_zy, ob.orbitXLow, ob.orbitYLow);
        }
        count++;
        {
            // PerturbExtras::Bad, RefOrbitCalc.cpp:550-562: entries whose parts or norm underflow a binary32
            const hreal<F> sq_x = hr_mul(double_zx, double_zx);
            const hreal<F> sq_y = hr_mul(double_zy, double_zy);
            const hreal<F> norm = hr_reduced(hr_mul(hr_add(sq_x, sq_y), hr_from_mant<F>(glitch)));
            // (T)mpf_get_d(z): templated HDRFloat(const U) with U = double -- the zero test is on the double, the
            // normalisation on its cast to SubType (HDRFloat.h:295-326)
            auto from_double = [](double d) {
                if (d == 0.0)
                    return hr_zero<F>();
                const F v = (F)d;
                const auto bits = to_bits<F>(v);
                const int32_t fe = (int32_t)((bits & fbits<F>::kExpMask) >> fbits<F>::kShift) - fbits<F>::kBias;
                return hreal<F>{from_bits<F>((bits & fbits<F>::kKeepMask) | fbits<F>::kOneExp), fe};
            };
            const hreal<F> zx_reduced = hr_reduced(hr_abs(from_double(mpf_get_d(zx))));
            const hreal<F> zy_reduced = hr_reduced(hr_abs(from_double(mpf_get_d(zy))));
            const bool underflow = hr_cmp_pos(zx_reduced, small_float) <= 0 || hr_cmp_pos(zy_reduced, small_float) <= 0 ||
                                   hr_cmp_pos(norm, small_float) <= 0;
            ob.bad.push_back(underflow ? 1 : 0);
        }

        if (periodicity) {
            hr_reduce(dzdcX);
            const hreal<F> dzdcX1 = hr_abs(dzdcX);
            hr_reduce(dzdcY);
            const hreal<F> dzdcY1 = hr_abs(dzdcY);
            hr_reduce(double_zx);
            const hreal<F> zxCopy1 = hr_abs(double_zx);
            hr_reduce(double_zy);
            const hreal<F> zyCopy1 = hr_abs(double_zy);
            const hreal<F> n2 = hr_max_pos(zxCopy1, zyCopy1); // HdrMaxPositiveReduced
            const hreal<F> r0 = hr_max_pos(dzdcX1, dzdcY1);
            hreal<F> n3 = hr_mul(hr_mul(ob.maxRadius, r0), HighTwo);
            hr_reduce(n3);
            if (hr_cmp_pos(n2, n3) < 0) {
                ob.period = count; // GetCountOrbitEntries()
                break;
            } else {
                const hreal<F> dzdcXOrig = dzdcX;
                dzdcX = hr_add(hr_mul(HighTwo, hr_sub(hr_mul(double_zx, dzdcX), hr_mul(double_zy, dzdcY))), HighOne);
                dzdcY = hr_mul(HighTwo, hr_add(hr_mul(double_zx, dzdcY), hr_mul(double_zy, dzdcXOrig)));
            }
        }

        mpf_mul(t1, zx, zx);
        mpf_mul(t2, zy, zy);
        mpf_sub(zx, t1, t2);
        mpf_add(zx, zx, cx);
        mpf_mul(zy, zx2, zy);
        mpf_add(zy, zy, cy);

        // RefOrbitCalc.cpp:616-622: tests z_i + c (not z_{i+1}), unreduced, against 256.
        const hreal<F> tempZX = hr_add(double_zx, cx_cast);
        const hreal<F> tempZY = hr_add(double_zy, cy_cast);
        const hreal<F> zn = hr_add(hr_mul(tempZX, tempZX), hr_mul(tempZY, tempZY));
        if (hr_cmp_pos(zn, TwoFiftySix) > 0)
            break;
    }
    mpf_clear(cx);
    mpf_clear(cy);
    mpf_clear(zx);
    mpf_clear(zy);
    mpf_clear(zx2);
    mpf_clear(t1);
    mpf_clear(t2);
    ob.bad.back() = 0; // results->SetBad(false), RefOrbitCalc.cpp:625-627

    if (ob.compressed) {
        // What every consumer of a compressed orbit sees through RuntimeDecompressor::GetCompressedComplex
        // (PerturbationResultsHelpers.h:46-161): the waypoint at or below the index, advanced with runOneIter.  The
        // cache in that function only avoids recomputation; the value at an index is a pure function of the waypoints.
        ob.x.assign(count, hr_zero<F>());
        ob.y.assign(count, hr_zero<F>());
        for (size_t k = 0; k < ob.wp_index.size(); k++) {
            const uint64_t i0 = ob.wp_index[k];
            const uint64_t i1 = k + 1 < ob.wp_index.size() ? ob.wp_index[k + 1] : count;
            hreal<F> zx_ = ob.wp_x[k], zy_ = ob.wp_y[k];
            for (uint64_t i = i0; i < i1; i++) {
                ob.x[i] = zx_;
                ob.y[i] = zy_;
                rc_one_iter(zx_, zy_, ob.orbitXLow, ob.orbitYLow);
            }
        }
    }
}

} // namespace

// ------------------------------------------------------------------ Imagina ".im" files that carry a reference orbit
// RefOrbitCalc::SaveOrbitResults(results, filename) (RefOrbitCalc.cpp:3039-3115) writes, after the location section
// (header, halfH = the orbit's MaxRadius, iteration limit = MaxIterations - 1, centre), at ReferenceOffset:
//   ReferenceHeader { bool ExtendedRange }                                                     1 B
//   ReferenceTrivialContent { HRReal AbsolutePrecision {2, -precisionInBits}, RelativePrecision {}, ValidRadius = MaxRadius }   48 B
//   LAReferenceTrivialContent { complex<double> Refc; size_t RefIt = count - 1, MaxIt = MaxIterations - 2; bool x4
//                               (IsPeriodic = period != 0); ImaginaATInfo AT {}; size_t LAStageCount = 0 }                     192 B
//   size_t n; n x { HRReal x, HRReal y, CompressionIndexField (63-bit orbit index, 1 rebase bit) }                            40 B each
//   size_t r; r x uint64 rebase indices
// (PerturbationResults::SaveOrbitBin, PerturbationResults.cpp:2013-2082; sizes and offsets checked against the reference's
// headers in tests/test_im_orbit.py).  The waypoints are the orbit under "max compression" (PerturbationResults::
// CompressMax, :1347-1640 -- the waypoint scheme of Imagina: the orbit is re-derived by perturbing it against its own
// beginning, and a waypoint is stored where that drifts by more than sqrt(10^-CompressionErrorExp) relative, Chebyshev norm);
// the reader rebuilds the full orbit from them (LoadOrbitBin :2104-2211, DecompressMax :1660-1840, with the backwards
// Newton correction of every stretch between two waypoints).  Restated here for T = HDRFloat<float> ("Sharks:)" magic) and
// HDRFloat<double> (Imagina's magic), ExtendedRange = true.
namespace {

template <class T> void rc_one_iter_plain(T &zx, T &zy, T cxLow, T cyLow); // (defined with the plain orbit builder below)

// (F is a number family of la_math.hpp: float / double = HDRFloat<F>, plain<float> / plain<double> = the type itself --
// the reference's CompressMax / DecompressMax are one template over T, with HdrReduce a no-op, HdrAbs = fabs and
// HdrMaxReduced = `(one > two) ? one : two` for a plain T, HDRFloat.h:1385-1500)
template <class T> preal<T> hr_add(preal<T> a, preal<T> b) { return preal<T>{a.m + b.m}; }
template <class T> preal<T> hr_sub(preal<T> a, preal<T> b) { return preal<T>{a.m - b.m}; }
template <class T> preal<T> hr_abs(preal<T> a) { return preal<T>{std::fabs(a.m)}; }
template <class T> int hr_cmp(preal<T> a, preal<T> b) { return a.m > b.m ? 1 : (a.m < b.m ? -1 : 0); }
template <class T> void rc_one_iter(preal<T> &zx, preal<T> &zy, preal<T> cxLow, preal<T> cyLow)
{
    rc_one_iter_plain(zx.m, zy.m, cxLow.m, cyLow.m);
}

template <class F> struct MaxWaypoints {
    std::vector<real_t<F>> x, y;
    std::vector<uint64_t> index;
    std::vector<uint8_t> rebase;
    std::vector<uint64_t> rebases;
};

template <class R> R mc_norm(R x, R y)
{
    // HdrMaxReduced(HdrAbs(x), HdrAbs(y)) (HDRFloat.h:1477-1500: `one.compareTo(two) > 0 ? one : two`), HdrReduce
    const R ax = hr_abs(x), ay = hr_abs(y);
    return hr_reduced(hr_cmp(ax, ay) > 0 ? ax : ay);
}
template <class R> R mc_norm_times(R x, R y, R t)
{
    const R ax = hr_abs(x), ay = hr_abs(y);
    return hr_reduced(hr_mul(hr_cmp(ax, ay) > 0 ? ax : ay, t));
}
// dz' = (2 Z + dz) dz in the reference's operand order (PerturbationResults.cpp:1489-1493, 1610-1614, 1842-1848)
template <class F> void mc_dz_step(real_t<F> &dzX, real_t<F> &dzY, real_t<F> Zx, real_t<F> Zy)
{
    const real_t<F> Two = mk<F>::number(2.0);
    const real_t<F> old = dzX;
    dzX = hr_sub(hr_add(hr_sub(hr_mul(hr_mul(Two, Zx), dzX), hr_mul(hr_mul(Two, Zy), dzY)), hr_mul(dzX, dzX)), hr_mul(dzY, dzY));
    hr_reduce(dzX);
    dzY = hr_add(hr_add(hr_mul(hr_mul(Two, Zx), dzY), hr_mul(hr_mul(Two, Zy), old)), hr_mul(hr_mul(Two, old), dzY));
    hr_reduce(dzY);
}

// PerturbationResults::CompressMax, PerturbationResults.cpp:1347-1640 (includeDummy = false)
template <class F, class Orb>
MaxWaypoints<F> compress_max(const Orb &ob, real_t<F> orbitXLow, real_t<F> orbitYLow, int compression_exp)
{
    using hr = real_t<F>;
    using S = scalar_t<F>;
    MaxWaypoints<F> w;
    const uint64_t count = ob.x.size();
    const hr threshold2 = mk<F>::number((S)std::sqrt(std::pow(10.0, compression_exp)));
    const hr constant1 = hr_reduced(mk<F>::number((S)0x1.0p-4));
    const hr constant2 = hr_reduced(mk<F>::number((S)0x1.000001p0));
    hr zx = orbitXLow, zy = orbitYLow;
    auto push = [&](hr x, hr y, uint64_t i, bool rebase) {
        w.x.push_back(x), w.y.push_back(y), w.index.push_back(i), w.rebase.push_back(rebase ? 1 : 0);
    };
    uint64_t i = 1;
    for (; i < count; i++) {
        const hr outX = ob.x[i], outY = ob.y[i];
        const hr norm_z = mc_norm(outX, outY);
        if (hr_cmp_pos(norm_z, constant1) < 0) {
            zx = outX, zy = outY;
            push(outX, outY, i, true);
            break;
        } else {
            const hr err = mc_norm_times(hr_sub(zx, outX), hr_sub(zy, outY), threshold2);
            if (hr_cmp_pos(err, norm_z) >= 0) {
                zx = outX, zy = outY;
                push(outX, outY, i, false);
            }
        }
        rc_one_iter(zx, zy, orbitXLow, orbitYLow);
    }
    hr dzX = zx, dzY = zy;
    uint64_t prev_waypoint = i;
    mc_dz_step<F>(dzX, dzY, ob.x[0], ob.y[0]);
    i++;
    uint64_t j = 1;
    for (; i < count; i++, j++) {
        const hr outXi = ob.x[i], outYi = ob.y[i];
        hr outXj = ob.x[j], outYj = ob.y[j];
        zx = hr_add(dzX, outXj);
        zy = hr_add(dzY, outYj);
        const hr norm_z_orig = mc_norm(zx, zy);
        const hr norm_dz_orig = mc_norm_times(dzX, dzY, constant2);
        const hr err = mc_norm_times(hr_sub(zx, outXi), hr_sub(zy, outYi), threshold2);
        const bool condition1 = j >= prev_waypoint;
        const bool condition2 = hr_cmp_pos(err, norm_z_orig) >= 0;
        if (condition1 || condition2) {
            prev_waypoint = i;
            zx = outXi, zy = outYi;
            dzX = hr_sub(zx, outXj);
            dzY = hr_sub(zy, outYj);
            const hr norm_z = mc_norm(zx, zy), norm_dz = mc_norm(dzX, dzY);
            if (hr_cmp_pos(norm_z, norm_dz) < 0 || (i - j) * 4 < i) {
                dzX = zx, dzY = zy;
                j = 0;
                push(dzX, dzY, i, true);
            } else {
                push(dzX, dzY, i, false);
            }
        } else if (hr_cmp_pos(norm_z_orig, norm_dz_orig) < 0) {
            dzX = zx, dzY = zy;
            j = 0;
            if (!w.rebases.empty() && !w.index.empty() && w.rebases.back() > w.index.back())
                w.rebases.back() = i;
            else
                w.rebases.push_back(i);
        }
        outXj = ob.x[j], outYj = ob.y[j]; // j may have changed
        mc_dz_step<F>(dzX, dzY, outXj, outYj);
    }
    return w;
}

// PerturbationResults::DecompressMax<Disable>, PerturbationResults.cpp:1660-1850.  The waypoint and rebase lists carry the
// reader's terminators (index ~0) at their ends (LoadOrbitBin :2208-2210).
template <class F>
void decompress_max(const MaxWaypoints<F> &w, real_t<F> cxLow, real_t<F> cyLow, uint64_t target, std::vector<real_t<F>> &ox,
                    std::vector<real_t<F>> &oy)
{
    using hr = real_t<F>;
    ox.clear(), oy.clear();
    ox.reserve(target), oy.reserve(target);
    const hr Two = mk<F>::number(2.0);
    // CorrectOrbit, :1712-1760: the stretch [begin, end) is pulled onto the waypoint by a backwards Newton step per entry
    auto correct = [&](uint64_t begin, uint64_t end, hr diffX, hr diffY) {
        hr dzdcX = mk<F>::number(1.0), dzdcY = mk<F>::number(0.0);
        hr_reduce(diffX);
        hr_reduce(diffY);
        for (uint64_t i = end; i > begin;) {
            i--;
            const hr old = dzdcX;
            // dzdcX * Z.x * 2 - dzdcY * Z.y * 2 ; `* 2` is HDRFloat * int -> HDRFloat(int 2) = {1.0, 1}
            dzdcX = hr_sub(hr_mul(hr_mul(dzdcX, ox[i]), Two), hr_mul(hr_mul(dzdcY, oy[i]), Two));
            hr_reduce(dzdcX);
            dzdcY = hr_add(hr_mul(hr_mul(old, oy[i]), Two), hr_mul(hr_mul(dzdcY, ox[i]), Two));
            hr_reduce(dzdcY);
            const hr den = hr_add(hr_mul(dzdcX, dzdcX), hr_mul(dzdcY, dzdcY));
            hr resultReal = hr_div(hr_add(hr_mul(diffX, dzdcX), hr_mul(diffY, dzdcY)), den);
            hr_reduce(resultReal);
            hr resultImag = hr_div(hr_sub(hr_mul(diffY, dzdcX), hr_mul(diffX, dzdcY)), den);
            hr_reduce(resultImag);
            ox[i] = hr_reduced(hr_add(ox[i], resultReal));
            oy[i] = hr_reduced(hr_add(oy[i], resultImag));
        }
    };
    hr zx = mk<F>::zero(), zy = mk<F>::zero(); // T zx{}, zy{}
    uint64_t wp = 0, rb = 0;
    uint64_t next_index = w.index[0];
    uint64_t next_rebase = w.rebases[0];
    uint64_t i = 0, uncorrected_begin = 1;
    for (; i < target; i++) {
        if (i == next_index) {
            correct(uncorrected_begin, i, hr_sub(w.x[wp], zx), hr_sub(w.y[wp], zy));
            uncorrected_begin = i + 1;
            zx = w.x[wp], zy = w.y[wp];
            const bool rebase = w.rebase[wp] != 0;
            wp++;
            next_index = w.index[wp];
            if (rebase)
                break;
        }
        ox.push_back(zx), oy.push_back(zy);
        rc_one_iter(zx, zy, cxLow, cyLow);
    }
    uint64_t j = 0;
    hr dzX = zx, dzY = zy;
    for (; i < target; i++, j++) {
        zx = hr_add(dzX, ox[j]);
        zy = hr_add(dzY, oy[j]);
        if (i == next_index) {
            if (w.rebase[wp]) {
                dzX = zx, dzY = zy;
                j = 0;
            }
            correct(uncorrected_begin, i, hr_sub(w.x[wp], dzX), hr_sub(w.y[wp], dzY));
            uncorrected_begin = i + 1;
            dzX = w.x[wp], dzY = w.y[wp];
            zx = hr_add(dzX, ox[j]);
            zy = hr_add(dzY, oy[j]);
            wp++;
            next_index = w.index[wp];
        } else if (i == next_rebase) {
            rb++;
            next_rebase = w.rebases[rb];
            dzX = zx, dzY = zy;
            j = 0;
        } else {
            const hr norm_z = mc_norm(zx, zy), norm_dz = mc_norm(dzX, dzY);
            if (hr_cmp_pos(norm_z, norm_dz) < 0) {
                dzX = zx, dzY = zy;
                j = 0;
            }
        }
        ox.push_back(zx), oy.push_back(zy);
        mc_dz_step<F>(dzX, dzY, ox[j], oy[j]);
    }
}

struct ImHR { // Imagina::HRReal = HDRFloat<double, Left, int64_t>
    double mantissa;
    int64_t exp;
};
static_assert(sizeof(ImHR) == 16, "HRReal");

template <class F> ImHR im_hr(hreal<F> v) { return ImHR{(double)v.m, (int64_t)v.e}; }
// Imagina::HRReal{T} for a plain T: the templated HDRFloat(const U number) constructor, which normalises (HDRFloat.h:295-363)
template <class T> ImHR im_hr(preal<T> v)
{
    const hreal<double> h = hr_from_number<double>((double)v.m);
    return ImHR{h.m, (int64_t)h.e};
}

// F: the number family (float / double: HDRFloat<F>, ExtendedRange = true, 40-byte waypoints of two HRReal and the index field;
// plain<float> / plain<double>: ExtendedRange = false, 24-byte waypoints of two doubles and the index field,
// PerturbationResults.cpp:2047-2075).  The magic follows the SubType (RefOrbitCalc.cpp:3052-3062).
template <class F, class Orb>
int save_im_orbit(const Orb &ob, real_t<F> orbitXLow, real_t<F> orbitYLow, uint64_t num_iterations, int compression_exp,
                  const char *path, int exp_bytes)
{
    if ((exp_bytes != 4 && exp_bytes != 8) || ob.x.size() < 2)
        return -1;
    FILE *f = fopen(path, "wb");
    if (!f)
        return -1;
    uint64_t header[4] = {sizeof(scalar_t<F>) == 4 ? kSharksMagic : kImMagic, 0, 32, 0};
    fwrite(header, 8, 4, f);
    const ImHR halfH = im_hr(ob.maxRadius); // Imagina::HRReal{results.GetMaxRadius()}
    fwrite(&halfH, sizeof(halfH), 1, f);
    const uint64_t iteration_limit = num_iterations; // GetMaxIterations() - 1, MaxIterations = NumIterations + 1 (:859)
    fwrite(&iteration_limit, 8, 1, f);
    im_write_mpf(f, ob.cx.v, exp_bytes);
    im_write_mpf(f, ob.cy.v, exp_bytes);
    const uint64_t reference_offset = (uint64_t)ftell(f);
    const MaxWaypoints<F> w = compress_max<F>(ob, orbitXLow, orbitYLow, compression_exp);
    const uint8_t extended_range = num<F>::is_plain ? 0 : 1; // results.IsHDR
    fwrite(&extended_range, 1, 1, f);
    // ReferenceTrivialContent
    const ImHR trivial[3] = {ImHR{2.0, -(int64_t)mpf_get_prec(ob.cx.v)}, ImHR{0.0, 0}, im_hr(ob.maxRadius)};
    fwrite(trivial, sizeof(ImHR), 3, f);
    // LAReferenceTrivialContent, 192 bytes: Refc @0, RefIt @16, MaxIt @24, bools @32..35 (IsPeriodic @34), AT @40, LAStageCount @184
    unsigned char la[192];
    memset(la, 0, sizeof(la));
    const double refc[2] = {mpf_get_d(ob.cx.v), mpf_get_d(ob.cy.v)};
    memcpy(la + 0, refc, 16);
    const uint64_t ref_it = ob.x.size() - 1, max_it = num_iterations + 1 - 2;
    memcpy(la + 16, &ref_it, 8);
    memcpy(la + 24, &max_it, 8);
    la[34] = ob.period != 0 ? 1 : 0;
    fwrite(la, 1, sizeof(la), f);
    const uint64_t n = w.x.size();
    fwrite(&n, 8, 1, f);
    for (uint64_t k = 0; k < n; k++) {
        if constexpr (num<F>::is_plain) {
            const double xy[2] = {(double)w.x[k].m, (double)w.y[k].m};
            fwrite(xy, sizeof(double), 2, f);
        } else {
            const ImHR xy[2] = {im_hr(w.x[k]), im_hr(w.y[k])};
            fwrite(xy, sizeof(ImHR), 2, f);
        }
        const uint64_t field = (w.index[k] & 0x7FFFFFFFFFFFFFFFull) | ((uint64_t)w.rebase[k] << 63);
        fwrite(&field, 8, 1, f);
    }
    const uint64_t r = w.rebases.size();
    fwrite(&r, 8, 1, f);
    if (r)
        fwrite(w.rebases.data(), 8, r, f);
    header[3] = reference_offset;
    fseek(f, 0, SEEK_SET);
    fwrite(header, 8, 4, f);
    const bool ok = !ferror(f);
    fclose(f);
    return ok ? 0 : -1;
}

// the stored orbit of an open file (positioned anywhere), into ob: LoadOrbitBin + DecompressMax<Disable>
template <class F, class Orb>
bool load_im_orbit(FILE *f, uint64_t reference_offset, Orb &ob, uint64_t file_iteration_limit, ImHR halfH)
{
    if (fseek(f, (long)reference_offset, SEEK_SET) != 0)
        return false;
    uint8_t extended_range = 0;
    ImHR trivial[3];
    unsigned char la[192];
    uint64_t n = 0;
    if (fread(&extended_range, 1, 1, f) != 1 || fread(trivial, sizeof(ImHR), 3, f) != 3 || fread(la, 1, sizeof(la), f) != sizeof(la) ||
        fread(&n, 8, 1, f) != 1 || (extended_range != 0) == num<F>::is_plain || n == 0 || n > (1ull << 32))
        return false;
    uint64_t ref_it;
    memcpy(&ref_it, la + 16, 8);
    // a file is untrusted input: the orbit it announces must fit the device-side 32-bit indices and may not be longer than
    // the iteration limit it was computed for (a crafted RefIt would otherwise size the vectors below)
    if (ref_it >= (1ull << 32) || (file_iteration_limit != 0 && ref_it > file_iteration_limit))
        return false;
    // ... nor longer than this host can expand (the iteration limit comes from the same file): DecompressMax sizes its vectors
    // for ref_it entries of 2 x 16 bytes, and with overcommit a crafted length ends in an OOM kill rather than a bad_alloc
    if (ref_it > kMaxImOrbitEntries)
        return false;
    const bool periodic = la[34] != 0;
    MaxWaypoints<F> w;
    auto exp_fits = [](const ImHR &h) { return h.exp >= INT32_MIN && h.exp <= INT32_MAX; };
    if (!exp_fits(halfH))
        return false;
    uint64_t prev_index = 0;
    for (uint64_t k = 0; k < n; k++) {
        ImHR xy[2];
        double pxy[2] = {0.0, 0.0};
        uint64_t field;
        if constexpr (num<F>::is_plain) {
            if (fread(pxy, sizeof(double), 2, f) != 2 || fread(&field, 8, 1, f) != 1)
                return false;
            xy[0] = xy[1] = ImHR{0.0, 0};
        } else {
            if (fread(xy, sizeof(ImHR), 2, f) != 2 || fread(&field, 8, 1, f) != 1)
                return false;
        }
        // waypoints are strictly increasing orbit positions >= 1 (entry 0 is the implicit zero start; DecompressMax reads the
        // entry BEFORE a rebasing waypoint), inside the announced orbit, with exponents that fit HDRFloat's int32
        const uint64_t idx = field & 0x7FFFFFFFFFFFFFFFull;
        if (idx == 0 || idx <= prev_index || idx > ref_it || !exp_fits(xy[0]) || !exp_fits(xy[1]))
            return false;
        prev_index = idx;
        if constexpr (num<F>::is_plain) {
            w.x.push_back(real_t<F>{(scalar_t<F>)pxy[0]}); // static_cast<T>(x), PerturbationResults.cpp:2177-2183
            w.y.push_back(real_t<F>{(scalar_t<F>)pxy[1]});
        } else {
            w.x.push_back(hreal<F>{(F)xy[0].mantissa, (int32_t)xy[0].exp});
            w.y.push_back(hreal<F>{(F)xy[1].mantissa, (int32_t)xy[1].exp});
        }
        w.index.push_back(field & 0x7FFFFFFFFFFFFFFFull);
        w.rebase.push_back((uint8_t)(field >> 63));
    }
    uint64_t r = 0;
    // every rebase is an orbit position, so a file that announces more of them than the (already bounded) orbit has entries
    // is not one CompressMax wrote -- refused BEFORE anything is allocated for it; the list itself is read in chunks, so the
    // memory that gets touched follows the bytes that are really there
    if (fread(&r, 8, 1, f) != 1 || r > ref_it + 1)
        return false;
    for (uint64_t done = 0; done < r;) {
        const uint64_t chunk = std::min<uint64_t>(r - done, 1u << 16);
        w.rebases.resize(done + chunk);
        if (fread(w.rebases.data() + done, 8, chunk, f) != chunk)
            return false;
        done += chunk;
    }
    // the reader's terminators (:2208-2210): {{}, {}, ~0ull, false} and ~0ull
    w.x.push_back(mk<F>::zero()), w.y.push_back(mk<F>::zero()), w.index.push_back(0x7FFFFFFFFFFFFFFFull), w.rebase.push_back(0);
    w.rebases.push_back(~0ull);
    // InitResults(DontSaveForReuse, orbitX, orbitY, radius, fileProvidedIters - 1, 0), :2125-2134
    const uint64_t count = ref_it + 1; // m_UncompressedItersInOrbit
    ob.period = periodic ? ref_it + 1 : 0;
    ob.compressed = false;
    if constexpr (num<F>::is_plain) {
        using T = scalar_t<F>;
        // static_cast<T>(halfH.toDouble()), :2136-2137; toDouble = mantissa * getMultiplier(exp), HDRFloat.h:553-557
        ob.maxRadius = preal<T>{(T)(halfH.mantissa * multiplier<double>((int32_t)halfH.exp))};
        ob.orbitXLow = (T)mpf_get_d(ob.cx.v);                                    // T{orbitX}
        ob.orbitYLow = (T)mpf_get_d(ob.cy.v);
        decompress_max(w, preal<T>{ob.orbitXLow}, preal<T>{ob.orbitYLow}, count, ob.x, ob.y);
    } else {
        ob.maxRadius = hr_reduced(hreal<F>{(F)halfH.mantissa, (int32_t)halfH.exp});
        ob.orbitXLow = hr_from_mpf<F>(ob.cx.v);
        ob.orbitYLow = hr_from_mpf<F>(ob.cy.v);
        decompress_max(w, ob.orbitXLow, ob.orbitYLow, count, ob.x, ob.y);
        ob.bad.assign(ob.x.size(), 0);
    }
    return ob.x.size() == count;
}

} // namespace

extern "C" int fsh_orbit_save_im(const fsh_orbit *o, uint64_t num_iterations, int compression_exp, const char *path,
                                 int exp_bytes)
{
    return o->is64 ? save_im_orbit<double>(o->d, o->d.orbitXLow, o->d.orbitYLow, num_iterations, compression_exp, path, exp_bytes)
                   : save_im_orbit<float>(o->f, o->f.orbitXLow, o->f.orbitYLow, num_iterations, compression_exp, path, exp_bytes);
}

static fsh_orbit *orbit_load_im(FILE *f, uint64_t *iteration_limit);

extern "C" fsh_orbit *fsh_orbit_load_im(const char *path, uint64_t *iteration_limit)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return nullptr;
    // nothing may unwind through the C boundary (a corrupt file can still make a vector throw): NULL, like any other refusal
    fsh_orbit *out = nullptr;
    try {
        out = orbit_load_im(f, iteration_limit);
    } catch (...) {
        out = nullptr;
    }
    fclose(f);
    return out;
}

static fsh_orbit *orbit_load_im(FILE *f, uint64_t *iteration_limit)
{
    uint64_t header[4];
    ImHR hh;
    uint64_t limit = 0;
    std::unique_ptr<fsh_orbit> out;
    if (fread(header, 8, 4, f) == 4 && (header[0] == kImMagic || header[0] == kSharksMagic) && header[3] != 0 &&
        fseek(f, (long)header[2], SEEK_SET) == 0 && fread(&hh, sizeof(hh), 1, f) == 1 && fread(&limit, 8, 1, f) == 1) {
        // (a file is untrusted input: GMP aborts on an absurd precision, it does not throw -- the deepest view the renderer's
        // int32 exponents can express needs 2^31 bits)
        if (hh.exp < -(int64_t)kMaxImPrecisionBits)
            return nullptr;
        const uint64_t precision = (uint64_t)(-std::min<int64_t>(0, hh.exp)) + 120u;
        mpf_set_default_prec(precision);
        Mp X(precision, 0), Y(precision, 0);
        const long at = ftell(f);
        int width_ok = 0;
        for (int exp_bytes : {4, 8}) {
            fseek(f, at, SEEK_SET);
            if (im_read_mpf(f, X.v, exp_bytes) && im_read_mpf(f, Y.v, exp_bytes) && (uint64_t)ftell(f) == header[3]) {
                width_ok = exp_bytes;
                break;
            }
        }
        if (width_ok) {
            out = std::make_unique<fsh_orbit>();
            out->is64 = header[0] == kImMagic ? 1 : 0; // SubType double <-> Imagina's magic, float <-> "Sharks:)" (:3052-3058)
            bool ok;
            if (out->is64) {
                out->d.cx = X, out->d.cy = Y, out->d.prec_bits = precision;
                ok = load_im_orbit<double>(f, header[3], out->d, limit, ImHR{hh.mantissa, hh.exp});
            } else {
                out->f.cx = X, out->f.cy = Y, out->f.prec_bits = precision;
                ok = load_im_orbit<float>(f, header[3], out->f, limit, ImHR{hh.mantissa, hh.exp});
            }
            if (!ok)
                out.reset();
            else if (iteration_limit)
                *iteration_limit = limit;
        }
    }
    return out.release();
}

extern "C" fsh_orbit *fsh_orbit_create_ex(const fsh_view *v, int is64, uint64_t max_iter, int periodicity,
                                          int compression_exp)
{
    auto ob = std::make_unique<fsh_orbit>();
    ob->is64 = is64;
    if (is64)
        build_orbit<double>(*v, max_iter, periodicity != 0, ob->d, compression_exp);
    else
        build_orbit<float>(*v, max_iter, periodicity != 0, ob->f, compression_exp);
    return ob.release();
}
extern "C" fsh_orbit *fsh_orbit_create(const fsh_view *v, int is64, uint64_t max_iter, int periodicity)
{
    return fsh_orbit_create_ex(v, is64, max_iter, periodicity, -1);
}
extern "C" int fsh_orbit_is64(const fsh_orbit *o) { return o->is64; }
extern "C" int fsh_orbit_is_compressed(const fsh_orbit *o) { return (o->is64 ? o->d.compressed : o->f.compressed) ? 1 : 0; }
extern "C" uint64_t fsh_orbit_compressed_count(const fsh_orbit *o)
{
    return o->is64 ? o->d.wp_index.size() : o->f.wp_index.size();
}
// GPUReferenceIter<HDRFloat<float>, SimpleCompression>[] (24 B: index, x, y) -- what FractalShark's compressed
// PerturbationResults holds and what fs_upload_orbit_compressed consumes.
extern "C" const fs_orbit_hdr32_rc *fsh_orbit_compressed_data_hdr32(fsh_orbit *o)
{
    if (o->is64 || !o->f.compressed)
        return nullptr;
    auto &ob = o->f;
    if (ob.packed_rc32.size() != ob.wp_index.size()) {
        ob.packed_rc32.resize(ob.wp_index.size());
        for (size_t k = 0; k < ob.wp_index.size(); k++)
            ob.packed_rc32[k] = fs_orbit_hdr32_rc{ob.wp_index[k], ob.wp_x[k].m, ob.wp_x[k].e, ob.wp_y[k].e, ob.wp_y[k].m};
    }
    return ob.packed_rc32.data();
}
extern "C" const fs_orbit_hdr64_rc *fsh_orbit_compressed_data_hdr64(fsh_orbit *o)
{
    if (!o->is64 || !o->d.compressed)
        return nullptr;
    auto &ob = o->d;
    if (ob.packed_rc64.size() != ob.wp_index.size()) {
        ob.packed_rc64.resize(ob.wp_index.size());
        for (size_t k = 0; k < ob.wp_index.size(); k++)
            ob.packed_rc64[k] =
                fs_orbit_hdr64_rc{ob.wp_index[k], ob.wp_x[k].m, ob.wp_x[k].e, 0, ob.wp_y[k].e, 0, ob.wp_y[k].m};
    }
    return ob.packed_rc64.data();
}
extern "C" void fsh_orbit_low_hdr64(const fsh_orbit *o, fs_real_hdr64 out[2])
{
    out[0] = fs_real_hdr64{o->d.orbitXLow.m, o->d.orbitXLow.e, 0};
    out[1] = fs_real_hdr64{o->d.orbitYLow.m, o->d.orbitYLow.e, 0};
}
extern "C" void fsh_orbit_low_hdr32(const fsh_orbit *o, fs_real_hdr32 out[2])
{
    out[0] = fs_real_hdr32{o->f.orbitXLow.m, o->f.orbitXLow.e};
    out[1] = fs_real_hdr32{o->f.orbitYLow.m, o->f.orbitYLow.e};
}
extern "C" void fsh_orbit_destroy(fsh_orbit *o) { delete o; }
extern "C" uint64_t fsh_orbit_count(const fsh_orbit *o) { return o->is64 ? o->d.x.size() : o->f.x.size(); }
// Test hook: entries idx[k] of the (uncompressed) orbit scaled by 2^exp2[k].  The result is no longer the orbit of any view;
// it is an orbit-shaped input with period boundaries where a test wants them (what the LA builders make of adjacent deep
// minima, of a minimum right behind a worker's first index...).  Returns the number of entries changed.
extern "C" uint64_t fsh_orbit_scale_entries(fsh_orbit *o, const uint64_t *idx, const int32_t *exp2, uint64_t n)
{
    uint64_t changed = 0;
    auto apply = [&](auto &ob) {
        if (ob.compressed)
            return;
        for (uint64_t k = 0; k < n; k++) {
            if (idx[k] >= ob.x.size())
                continue;
            ob.x[idx[k]].e += exp2[k];
            ob.y[idx[k]].e += exp2[k];
            changed++;
        }
        ob.packed32.clear();
        ob.packed64.clear();
    };
    if (o->is64)
        apply(o->d);
    else
        apply(o->f);
    return changed;
}
extern "C" uint64_t fsh_orbit_period(const fsh_orbit *o) { return o->is64 ? o->d.period : o->f.period; }

extern "C" const fs_orbit_hdr32 *fsh_orbit_data_hdr32(fsh_orbit *o)
{
    if (o->is64)
        return nullptr;
    auto &ob = o->f;
    if (ob.packed32.size() != ob.x.size()) {
        ob.packed32.resize(ob.x.size());
        for (size_t i = 0; i < ob.x.size(); i++)
            ob.packed32[i] = fs_orbit_hdr32{ob.x[i].m, ob.x[i].e, ob.y[i].e, ob.y[i].m};
    }
    return ob.packed32.data();
}

extern "C" const fs_orbit_hdr64 *fsh_orbit_data_hdr64(fsh_orbit *o)
{
    if (!o->is64)
        return nullptr;
    auto &ob = o->d;
    if (ob.packed64.size() != ob.x.size()) {
        ob.packed64.resize(ob.x.size());
        for (size_t i = 0; i < ob.x.size(); i++)
            ob.packed64[i] = fs_orbit_hdr64{ob.x[i].m, ob.x[i].e, 0, ob.y[i].e, 0, ob.y[i].m};
    }
    return ob.packed64.data();
}

// PerturbExtras::Bad orbits for the scaled kernel (GpuHDRx32PerturbedScaled): the HDRFloat<float> orbit with its flags,
// and its binary32 copy (RefOrbitCalc::CopyUsefulPerturbationResults -> CopyFullOrbitVector, PerturbationResults.cpp:
// 239-262: (float)x = HDRFloat::operator T() = mantissa * getMultiplier(exp), HDRFloat.h:557-568).
extern "C" const fs_orbit_hdr32_bad *fsh_orbit_data_hdr32_bad(fsh_orbit *o)
{
    if (o->is64)
        return nullptr;
    auto &ob = o->f;
    if (ob.packed32_bad.size() != ob.x.size()) {
        ob.packed32_bad.resize(ob.x.size());
        for (size_t i = 0; i < ob.x.size(); i++)
            ob.packed32_bad[i] = fs_orbit_hdr32_bad{ob.bad[i], 0u, ob.x[i].m, ob.x[i].e, ob.y[i].e, ob.y[i].m};
    }
    return ob.packed32_bad.data();
}
extern "C" const fs_orbit_f32_bad *fsh_orbit_data_f32_bad(fsh_orbit *o)
{
    if (o->is64)
        return nullptr;
    auto &ob = o->f;
    if (ob.packed_f32_bad.size() != ob.x.size()) {
        ob.packed_f32_bad.resize(ob.x.size());
        for (size_t i = 0; i < ob.x.size(); i++)
            ob.packed_f32_bad[i] = fs_orbit_f32_bad{ob.bad[i] != 0 ? 1u : 0u, 0u, hr_to_native(ob.x[i]), hr_to_native(ob.y[i])};
    }
    return ob.packed_f32_bad.data();
}
extern "C" uint64_t fsh_orbit_bad_count(const fsh_orbit *o)
{
    uint64_t n = 0;
    for (uint8_t b : (o->is64 ? o->d.bad : o->f.bad))
        n += b;
    return n;
}

extern "C" void fsh_orbit_max_radius_hdr64(const fsh_orbit *o, fs_real_hdr64 *out)
{
    out->m = o->d.maxRadius.m;
    out->e = o->d.maxRadius.e;
    out->pad_ = 0;
}
extern "C" void fsh_orbit_max_radius_hdr32(const fsh_orbit *o, fs_real_hdr32 *out)
{
    out->m = o->f.maxRadius.m;
    out->e = o->f.maxRadius.e;
}

// dx, dy, centerX, centerY for the perturbation paths -- Fractal.cpp:2230-2238 / 2513-2521:
//   dx = Reduce(T((maxX-minX)/HP(W*AA))), centerX = Reduce(T(refX - minX)), centerY = Reduce(T(refY - maxY)).
template <class F> static void perturb_coords(const fsh_view &v, const OrbitT<F> &ob, uint32_t w_aa, uint32_t h_aa,
                                              hreal<F> out[4])
{
    mpf_set_default_prec(v.prec_bits);
    Mp dx = (v.maxX - v.minX) / Mp::from_ui(w_aa);
    Mp dy = (v.maxY - v.minY) / Mp::from_ui(h_aa);
    Mp cX = ob.cx - v.minX;
    Mp cY = ob.cy - v.maxY;
    out[0] = hr_reduced(hr_from_mpf<F>(dx.v));
    out[1] = hr_reduced(hr_from_mpf<F>(dy.v));
    out[2] = hr_reduced(hr_from_mpf<F>(cX.v));
    out[3] = hr_reduced(hr_from_mpf<F>(cY.v));
}

extern "C" void fsh_view_coords_perturb_hdr32(const fsh_view *v, const fsh_orbit *o, uint32_t w_aa, uint32_t h_aa,
                                              fs_real_hdr32 out[4])
{
    hreal<float> t[4];
    perturb_coords<float>(*v, o->f, w_aa, h_aa, t);
    for (int i = 0; i < 4; i++)
        out[i] = fs_real_hdr32{t[i].m, t[i].e};
}

extern "C" void fsh_view_coords_perturb_hdr64(const fsh_view *v, const fsh_orbit *o, uint32_t w_aa, uint32_t h_aa,
                                              fs_real_hdr64 out[4])
{
    hreal<double> t[4];
    perturb_coords<double>(*v, o->d, w_aa, h_aa, t);
    for (int i = 0; i < 4; i++)
        out[i] = fs_real_hdr64{t[i].m, t[i].e, 0};
}

// ------------------------------------------------------------------ 2x32 (HDRFloat<CudaDblflt<MattDblflt>>) inputs
// The reference never computes 2x32 inputs directly: RefOrbitCalc builds the HDRFloat<double> orbit and LA table
// (ConditionalT, HDRFloat.h:1734-1760; Fractal.cpp:2771-2806) and PerturbationResults::CopyPerturbationResults /
// LAReference::CopyLAReference convert them field by field (PerturbationResults.cpp:239-347, RefOrbitCalc.cpp:2490-2513,
// LAReference.h:163-213, LAInfoDeep.h:94-107, ATInfo.h:18-30).  Each mantissa goes through MattDblflt(double)
// (dblflt.h:37-52): split into (float)d and the float remainder, then one two-sum; exponents are carried over.
namespace {

inline void df_from_double(double other, float &head, float &tail)
{
    const float a = (float)other;
    const float b = (float)(other - (double)a);
    head = a + b;
    float t1 = head - a;
    float t2 = head - t1;
    t1 = b - t1;
    t2 = a - t2;
    tail = t1 + t2;
}
inline fs_real_2x32 real_2x32(const fs_real_hdr64 &r)
{
    fs_real_2x32 o;
    df_from_double(r.m, o.head, o.tail);
    o.e = r.e;
    return o;
}
inline fs_cplx_2x32 cplx_2x32(const fs_cplx_hdr64 &c)
{
    fs_cplx_2x32 o;
    df_from_double(c.re, o.re_head, o.re_tail);
    df_from_double(c.im, o.im_head, o.im_tail);
    o.e = c.e;
    return o;
}

} // namespace

// Host-side entry points into the product's 2x32 arithmetic (csrc/df32_math.hpp is one header for host and device): the
// CPU test-suite cross-checks them bit for bit against the oracle's independent restatement.
extern "C" void fsh_df32_op(int op, const float a[2], const float b[2], float out[2])
{
    const fs::df32 x(a[0], a[1]), y(b[0], b[1]);
    const fs::df32 r = op == 0 ? x + y : (op == 1 ? x - y : x * y);
    out[0] = r.head;
    out[1] = r.tail;
}
extern "C" void fsh_hr2_reduce(fs_real_2x32 *v)
{
    fs::hreal<fs::df32> h{fs::df32(v->head, v->tail), v->e};
    fs::hr_reduce(h);
    *v = fs_real_2x32{h.m.head, h.m.tail, h.e};
}
extern "C" void fsh_hr2_add(const fs_real_2x32 *a, const fs_real_2x32 *b, int subtract, fs_real_2x32 *out)
{
    const fs::hreal<fs::df32> x{fs::df32(a->head, a->tail), a->e}, y{fs::df32(b->head, b->tail), b->e};
    const fs::hreal<fs::df32> r = subtract ? fs::hr_sub(x, y) : fs::hr_add(x, y);
    *out = fs_real_2x32{r.m.head, r.m.tail, r.e};
}
extern "C" void fsh_hc2_reduce(fs_cplx_2x32 *v)
{
    fs::hcplx<fs::df32> c{fs::df32(v->re_head, v->re_tail), fs::df32(v->im_head, v->im_tail), v->e};
    fs::hc_reduce(c);
    *v = fs_cplx_2x32{c.re.head, c.re.tail, c.im.head, c.im.tail, c.e};
}

extern "C" void fsh_convert_orbit_hdr64_to_2x32(const fs_orbit_hdr64 *in, uint64_t n, fs_orbit_2x32 *out)
{
    for (uint64_t i = 0; i < n; i++) {
        fs_orbit_2x32 o;
        df_from_double(in[i].mx, o.x_head, o.x_tail);
        o.ex = in[i].ex;
        df_from_double(in[i].my, o.y_head, o.y_tail);
        o.ey = in[i].ey;
        out[i] = o;
    }
}

extern "C" void fsh_convert_orbit_rc_hdr64_to_2x32(const fs_orbit_hdr64_rc *in, uint64_t n, fs_orbit_2x32_rc *out)
{
    for (uint64_t i = 0; i < n; i++) {
        fs_orbit_2x32_rc o;
        o.index_and_rebase = in[i].index_and_rebase & 0x7FFFFFFFFFFFFFFFull;
        df_from_double(in[i].mx, o.x_head, o.x_tail);
        o.ex = in[i].ex;
        df_from_double(in[i].my, o.y_head, o.y_tail);
        o.ey = in[i].ey;
        out[i] = o;
    }
}

// m_OrbitXLow / m_OrbitYLow of the converted results: static_cast<T>(Convert<HighPrecision, double>(m_OrbitX))
// (PerturbationResults.cpp:306-307), T = HDRFloat<CudaDblflt>: the double is split into exponent and a mantissa in
// [1, 2), the mantissa into head + tail (HDRFloat.h:341-349).
extern "C" void fsh_orbit_low_2x32(const fsh_orbit *o, fs_real_2x32 out[2])
{
    const Mp *c[2] = {o->is64 ? &o->d.cx : &o->f.cx, o->is64 ? &o->d.cy : &o->f.cy};
    for (int k = 0; k < 2; k++) {
        const double number = mpf_get_d(c[k]->v);
        fs_real_2x32 r{0.0f, 0.0f, fs::kMinBigExp};
        if (number != 0.0) {
            uint64_t bits;
            memcpy(&bits, &number, 8);
            const int64_t f_exp = (int64_t)((bits & 0x7FF0000000000000ull) >> 52) - 1023;
            const uint64_t val = (bits & 0x800FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
            double f_val;
            memcpy(&f_val, &val, 8);
            df_from_double(f_val, r.head, r.tail);
            r.e = (int32_t)f_exp;
        }
        out[k] = r;
    }
}

extern "C" void fsh_convert_la_hdr64_to_2x32(const fs_la_hdr64_u32 *in, uint64_t n, fs_la_2x32_u32 *out)
{
    for (uint64_t i = 0; i < n; i++) {
        fs_la_2x32_u32 o;
        o.Ref = cplx_2x32(in[i].Ref);
        o.ZCoeff = cplx_2x32(in[i].ZCoeff);
        o.CCoeff = cplx_2x32(in[i].CCoeff);
        o.LAThreshold = real_2x32(in[i].LAThreshold);
        o.LAThresholdC = real_2x32(in[i].LAThresholdC);
        o.MinMag = real_2x32(in[i].MinMag);
        o.StepLength = in[i].StepLength;
        o.NextStageLAIndex = in[i].NextStageLAIndex;
        out[i] = o;
    }
}

extern "C" void fsh_convert_at_hdr64_to_2x32(const fs_at_hdr64_u32 *in, fs_at_2x32_u32 *out)
{
    out->StepLength = in->StepLength;
    out->ThresholdC = real_2x32(in->ThresholdC);
    out->SqrEscapeRadius = real_2x32(in->SqrEscapeRadius);
    out->RefC = cplx_2x32(in->RefC);
    out->ZCoeff = cplx_2x32(in->ZCoeff);
    out->CCoeff = cplx_2x32(in->CCoeff);
    out->InvZCoeff = cplx_2x32(in->InvZCoeff);
    out->CCoeffSqrInvZCoeff = cplx_2x32(in->CCoeffSqrInvZCoeff);
    out->CCoeffInvZCoeff = cplx_2x32(in->CCoeffInvZCoeff);
    out->CCoeffNormSqr = real_2x32(in->CCoeffNormSqr);
    out->RefCNormSqr = real_2x32(in->RefCNormSqr);
    out->factor = real_2x32(in->factor);
}

// FillGpuCoords / FillCoord(HighPrecision, HDRFloat<CudaDblflt<MattDblflt>>&), Fractal.cpp:1826-1844,2834-2840:
// HDRFloat(const mpf_t) (HDRFloat.h:366-391) -- mantissa from mpf_get_d_2exp in [0.5,1), head = (float)m,
// tail = (float)(m - head), *not* reduced and not re-normalised.
extern "C" void fsh_view_coords_perturb_2x32(const fsh_view *v, const fsh_orbit *o, uint32_t w_aa, uint32_t h_aa,
                                             fs_real_2x32 out[4])
{
    mpf_set_default_prec(v->prec_bits);
    const Mp dx = (v->maxX - v->minX) / Mp::from_ui(w_aa);
    const Mp dy = (v->maxY - v->minY) / Mp::from_ui(h_aa);
    const Mp cX = (o->is64 ? o->d.cx : o->f.cx) - v->minX;
    const Mp cY = (o->is64 ? o->d.cy : o->f.cy) - v->maxY;
    const Mp *src[4] = {&dx, &dy, &cX, &cY};
    for (int i = 0; i < 4; i++) {
        if (mpf_cmp_ui(src[i]->v, 0) == 0) {
            out[i] = fs_real_2x32{0.0f, 0.0f, fs::kMinBigExp};
            continue;
        }
        long e;
        const double m = mpf_get_d_2exp(&e, src[i]->v);
        const float head = (float)m;
        out[i] = fs_real_2x32{head, (float)(m - (double)head), (int32_t)e};
    }
}

// ------------------------------------------------------------------ LAv2 table
namespace {

// (the record arithmetic -- LAParams, the number families, LAInfo, la_init / la_step / la_composite / la_detect_period,
// ATInfoT, la_create_at, at_usable -- lives in csrc/la_math.hpp, shared with the device builder)
template <class F> struct LATable {
    std::vector<LAInfo<F>> las;
    std::vector<fs_la_stage_u32> stages; // only [0, stageCount) meaningful
    uint32_t stageCount = 0;
    bool useAT = false;
    bool isValid = false;
    ATInfoT<F> at;
    std::vector<fs_la_hdr32_u32> packed32;
    std::vector<fs_la_stage_u32> packedStages;
};

constexpr uint32_t kLowBound = 64;   // LAReference.h:56
// periodDivisor: 2 for PerturbExtras::Disable, 8 for SimpleCompression (LAReference.cpp:12-19)
constexpr uint32_t kMaxLAStages = 1024;

template <class F> struct orbit_of {
    using type = OrbitT<F>;
};
// PerturbationResults<uint32_t, T, Disable> for a plain T (only what the LA builder reads)
template <class T> struct PlainOrbit {
    std::vector<preal<T>> x, y; // entry 0 = {0, 0}
    uint64_t period = 0;
    preal<T> maxRadius{};
    Mp cx, cy;
    // PerturbExtras::SimpleCompression: waypoints; x / y then hold the orbit as RuntimeDecompressor reproduces it
    bool compressed = false;
    std::vector<uint64_t> wp_index;
    std::vector<T> wp_x, wp_y;
    T orbitXLow{}, orbitYLow{};
};
template <class T> struct orbit_of<plain<T>> {
    using type = PlainOrbit<T>;
};
template <class F> struct LABuilder {
    using Orb = typename orbit_of<F>::type;
    const LAParams p;
    const Orb &ob;
    LATable<F> &T;
    const int kPeriodDivisor;
    LABuilder(const Orb &o, LATable<F> &t) : p{}, ob(o), T(t), kPeriodDivisor(o.compressed ? 8 : 2) {}

    cplx_t<F> Z(uint64_t i) const { return hc_from_hr(ob.x[i], ob.y[i]); } // GetComplex<SubType>()

    uint32_t la_size() const { return (uint32_t)T.las.size(); }

    // Shared prologue of CreateLAFromOrbit / CreateLAFromOrbitMT (LAReference.cpp:40-151 == :256-352).
    // Returns 0 = finished with `false`, 1 = continue into the main scan.
    int stage0_prologue(uint32_t maxRef, LAInfo<F> &LA, uint32_t &i, uint32_t &Period, uint32_t &PeriodBegin,
                        uint32_t &PeriodEnd, uint32_t &nextStageLAIndex)
    {
        T.isValid = false;
        T.stages.assign(kMaxLAStages, fs_la_stage_u32{0, 0});
        T.useAT = false;
        T.stageCount = 0;
        T.stages[0].LAIndex = 0;

        Period = 0;
        LA = la_init<F>(p, mk<F>::czero());
        LA = la_step_new(p, LA, Z(1));
        nextStageLAIndex = 0;
        if (LA.ZCoeff.re == scalar_t<F>(0) && LA.ZCoeff.im == scalar_t<F>(0)) // isZCoeffZero
            return 0;

        for (i = 2; i < maxRef; i++) {
            LAInfo<F> NewLA;
            const bool PeriodDetected = la_step(p, LA, NewLA, Z(i));
            if (!PeriodDetected) {
                LA = NewLA;
                continue;
            }
            Period = i;
            LA.StepLength = Period;
            LA.NextStageLAIndex = nextStageLAIndex;
            T.las.push_back(LA);
            nextStageLAIndex = i;
            if (i + 1 < maxRef) {
                LA = la_step_new(p, la_init<F>(p, Z(i)), Z(i + 1));
                i += 2;
            } else {
                LA = la_init<F>(p, Z(i));
                i += 1;
            }
            break;
        }
        T.stageCount = 1;
        PeriodBegin = Period;
        PeriodEnd = PeriodBegin + Period;

        if (Period == 0) {
            if (maxRef > kLowBound) {
                LA = la_step_new(p, la_init<F>(p, Z(0)), Z(1));
                nextStageLAIndex = 0;
                i = 2;
                const double NthRoot = std::round(std::log2((double)maxRef) / kPeriodDivisor);
                Period = (uint32_t)std::round(std::pow((double)maxRef, 1.0 / NthRoot));
                PeriodBegin = 0;
                PeriodEnd = Period;
            } else {
                LA.StepLength = maxRef;
                LA.NextStageLAIndex = nextStageLAIndex;
                T.las.push_back(LA);
                T.las.push_back(la_init<F>(p, Z(maxRef)));
                T.stages[0].MacroItCount = 1;
                return 0;
            }
        } else if (Period > kLowBound) {
            T.las.pop_back();
            LA = la_step_new(p, la_init<F>(p, Z(0)), Z(1));
            nextStageLAIndex = 0;
            i = 2;
            const double NthRoot = std::round(std::log2((double)maxRef) / kPeriodDivisor);
            Period = (uint32_t)std::round(std::pow((double)maxRef, 1.0 / NthRoot));
            PeriodBegin = 0;
            PeriodEnd = Period;
        }
        return 1;
    }

    // CreateLAFromOrbit, LAReference.cpp:28-210
    bool create_stage0_st(uint32_t maxRef)
    {
        LAInfo<F> LA;
        uint32_t i, Period, PeriodBegin, PeriodEnd, nextIdx;
        if (!stage0_prologue(maxRef, LA, i, Period, PeriodBegin, PeriodEnd, nextIdx))
            return false;
        for (; i < maxRef; i++) {
            LAInfo<F> NewLA;
            const bool PeriodDetected = la_step(p, LA, NewLA, Z(i));
            if (!PeriodDetected && i < PeriodEnd) {
                LA = NewLA;
                continue;
            }
            LA.StepLength = i - PeriodBegin;
            LA.NextStageLAIndex = nextIdx;
            T.las.push_back(LA);
            nextIdx = i;
            PeriodBegin = i;
            PeriodEnd = PeriodBegin + Period;
            const uint32_t ip1 = i + 1;
            const bool detected = la_detect_period(p, NewLA, Z(ip1));
            if (detected || ip1 >= maxRef) {
                LA = la_init<F>(p, Z(i));
            } else {
                LA = la_step_new(p, la_init<F>(p, Z(i)), Z(ip1));
                i++;
            }
        }
        LA.StepLength = i - PeriodBegin;
        LA.NextStageLAIndex = nextIdx;
        T.las.push_back(LA);
        T.stages[0].MacroItCount = la_size();
        LAInfo<F> LA2 = la_init<F>(p, Z(maxRef));
        T.las.push_back(LA2);
        return true;
    }

    // CreateNewLAStage, LAReference.cpp:774-966
    bool create_new_stage(uint32_t maxRef)
    {
        LAInfo<F> LA;
        uint32_t liStep = 0, liNext = 0; // LAI
        uint32_t i, PeriodBegin, PeriodEnd;
        const uint32_t PrevStage = T.stageCount - 1;
        const uint32_t CurrentStage = T.stageCount;
        const uint32_t PrevStageLAIndex = T.stages[PrevStage].LAIndex;
        const uint32_t PrevStageMacroItCount = T.stages[PrevStage].MacroItCount;
        const LAInfo<F> PrevStageLA = T.las[PrevStageLAIndex];
        const uint32_t PrevStageLAI_Step = PrevStageLA.StepLength;
        const LAInfo<F> PrevStageLAp1 = T.las[PrevStageLAIndex + 1];
        const uint32_t PrevStageLAIp1_Step = PrevStageLAp1.StepLength;
        uint32_t Period = 0;

        if (CurrentStage >= kMaxLAStages)
            return false;
        T.stages[CurrentStage].LAIndex = la_size();

        LA = la_composite_new(p, PrevStageLA, PrevStageLAp1);
        liNext = 0;
        i = PrevStageLAI_Step + PrevStageLAIp1_Step;
        uint32_t j;
        for (j = 2; j < PrevStageMacroItCount; j++) {
            LAInfo<F> NewLA;
            const uint32_t idxj = PrevStageLAIndex + j;
            const LAInfo<F> PrevStageLAj = T.las[idxj];
            const bool PeriodDetected = la_composite(p, LA, NewLA, PrevStageLAj);
            if (PeriodDetected) {
                if (PrevStageLAj.LAThreshold.m == scalar_t<F>(0)) // isLAThresholdZero
                    break;
                Period = i;
                liStep = Period;
                LA.StepLength = liStep;
                LA.NextStageLAIndex = liNext;
                T.las.push_back(LA);
                liNext = j;
                const LAInfo<F> PrevStageLAjp1 = T.las[idxj + 1];
                if (la_detect_period(p, NewLA, PrevStageLAjp1.Ref) || j + 1 >= PrevStageMacroItCount) {
                    LA = PrevStageLAj;
                    i += PrevStageLAj.StepLength;
                    j++;
                } else {
                    LA = la_composite_new(p, PrevStageLAj, PrevStageLAjp1);
                    i += PrevStageLAj.StepLength + PrevStageLAjp1.StepLength;
                    j += 2;
                }
                break;
            }
            LA = NewLA;
            i += T.las[PrevStageLAIndex + j].StepLength;
        }
        T.stageCount++;

        PeriodBegin = Period;
        PeriodEnd = PeriodBegin + Period;

        if (Period == 0) {
            if (maxRef > PrevStageLAI_Step * kLowBound) {
                LA = la_composite_new(p, PrevStageLA, PrevStageLAp1);
                i = PrevStageLAI_Step + PrevStageLAIp1_Step;
                liNext = 0;
                j = 2;
                const double Ratio = ((double)maxRef) / PrevStageLAI_Step;
                const double NthRoot = std::round(std::log2((double)maxRef) / kPeriodDivisor);
                Period = PrevStageLAI_Step * (uint32_t)std::round(std::pow(Ratio, 1.0 / NthRoot));
                PeriodBegin = 0;
                PeriodEnd = Period;
            } else {
                liStep = maxRef;
                LA.StepLength = liStep;
                LA.NextStageLAIndex = liNext;
                T.las.push_back(LA);
                LAInfo<F> LA2 = la_init<F>(p, Z(maxRef));
                T.las.push_back(LA2);
                T.stages[CurrentStage].MacroItCount = 1;
                return false;
            }
        } else if (Period > PrevStageLAI_Step * kLowBound) {
            T.las.pop_back();
            LA = la_composite_new(p, PrevStageLA, PrevStageLAp1);
            i = PrevStageLAI_Step + PrevStageLAIp1_Step;
            liNext = 0;
            j = 2;
            const double Ratio = ((double)Period) / PrevStageLAI_Step;
            const double NthRoot = std::round(std::log2((double)maxRef) / kPeriodDivisor);
            Period = PrevStageLAI_Step * ((uint32_t)std::round(std::pow(Ratio, 1.0 / NthRoot)));
            PeriodBegin = 0;
            PeriodEnd = Period;
        }

        for (; j < PrevStageMacroItCount; j++) {
            LAInfo<F> NewLA;
            const uint32_t idxj = PrevStageLAIndex + j;
            const LAInfo<F> PrevStageLAj = T.las[idxj];
            const bool PeriodDetected = la_composite(p, LA, NewLA, PrevStageLAj);
            if (PeriodDetected || i >= PeriodEnd) {
                liStep = i - PeriodBegin;
                LA.StepLength = liStep;
                LA.NextStageLAIndex = liNext;
                T.las.push_back(LA);
                liNext = j;
                PeriodBegin = i;
                PeriodEnd = PeriodBegin + Period;
                const LAInfo<F> PrevStageLAjp1 = T.las[idxj + 1];
                if (la_detect_period(p, NewLA, PrevStageLAjp1.Ref) || j + 1 >= PrevStageMacroItCount) {
                    LA = PrevStageLAj;
                } else {
                    LA = la_composite_new(p, PrevStageLAj, PrevStageLAjp1);
                    i += T.las[idxj].StepLength;
                    j++;
                }
            } else {
                LA = NewLA;
            }
            i += T.las[PrevStageLAIndex + j].StepLength;
        }
        liStep = i - PeriodBegin;
        LA.StepLength = liStep;
        LA.NextStageLAIndex = liNext;
        T.las.push_back(LA);
        T.stages[CurrentStage].MacroItCount = la_size() - T.stages[CurrentStage].LAIndex;
        LA = la_init<F>(p, Z(maxRef));
        T.las.push_back(LA);
        return true;
    }

    // CreateATFromLA, LAReference.cpp:1050-1074
    // UseSmallExponents = UsingDblflt (RefOrbitCalc.cpp:2346): true when the HDRFloat<double> table is built to be
    // converted to 2x32, so that the AT escape radius fits a binary32 mantissa (lim = 2^32 instead of 2^256).
    bool useSmallExponents = false;
    void create_at(real_t<F> radius, bool useSmallExponents)
    {
        const real_t<F> SqrRadius = hr_reduced(hr_square(radius));
        for (uint32_t Stage = T.stageCount; Stage > 0;) {
            Stage--;
            const uint32_t LAIndex = T.stages[Stage].LAIndex;
            la_create_at(T.las[LAIndex], T.at, T.las[LAIndex + 1], useSmallExponents);
            T.at.StepLength = T.las[LAIndex].StepLength;
            if (T.at.StepLength > 0 && at_usable(T.at, SqrRadius)) {
                T.useAT = true;
                return;
            }
        }
        T.useAT = false;
    }

    // GenerateApproximationData, LAReference.cpp:971-1013
    void generate(int threads)
    {
        T.las.clear();
        const uint32_t maxRef = (uint32_t)ob.x.size() - 1;
        if (maxRef == 0) {
            T.isValid = false;
            return;
        }
        bool PeriodDetected;
        // CreateLAFromOrbitMT falls back to the single-threaded scan when
        // maxRefIteration / 50000 (capped by hardware_concurrency) is < 2 (LAReference.cpp:236-251).
        size_t threadCount = maxRef / 50000;
        if (threadCount > (size_t)threads)
            threadCount = threads;
        if (threadCount <= 1) {
            PeriodDetected = create_stage0_st(maxRef);
        } else {
            PeriodDetected = create_stage0_mt(maxRef, threadCount);
        }
        if (!PeriodDetected)
            return;
        while (create_new_stage(maxRef)) {
        }
        create_at(ob.maxRadius, useSmallExponents);
        T.isValid = true;
    }

    bool create_stage0_mt(uint32_t maxRef, size_t threadCount);
};

// CreateLAFromOrbitMT, LAReference.cpp:215-770, replayed sequentially.  The reference runs a "Starter"
// (thread 0, continues the scan from the prologue) and Workers 1..N-1 that each begin at a fixed orbit
// index, find their first period boundary, publish it (StartIndexPromise) and scan on until they pass
// the *next* worker's published start.  All cross-thread values are futures, so the outcome does not
// depend on timing: replaying the workers from last to first and the starter last reproduces it.
template <class F> bool LABuilder<F>::create_stage0_mt(uint32_t maxRef, size_t ThreadCount)
{
    LAInfo<F> LA;
    uint32_t i, Period, PeriodBegin, PeriodEnd, nextIdx;
    if (!stage0_prologue(maxRef, LA, i, Period, PeriodBegin, PeriodEnd, nextIdx))
        return false;

    std::vector<int64_t> StartIndex(ThreadCount, 0);
    std::vector<int64_t> FinishIndex(ThreadCount, 0);
    std::vector<std::vector<LAInfo<F>>> LAsPerThread(ThreadCount);
    std::vector<LAInfo<F>> LastLAPerThread(ThreadCount);

    // Workers, last first (each only waits on StartIndexFuture of higher thread ids).
    for (size_t ThreadID = ThreadCount - 1; ThreadID >= 1; ThreadID--) {
        size_t NextThread = ThreadID + 1;
        const size_t LastThread = ThreadCount - 1;
        const uint32_t Begin = (uint32_t)((uint64_t)maxRef * ThreadID / ThreadCount);
        uint32_t j = Begin;
        const uint32_t End = (uint32_t)((uint64_t)maxRef * NextThread / ThreadCount);

        uint32_t liNext = j;
        LAInfo<F> LA_2 = la_step_new(p, la_init<F>(p, Z(j)), Z(j + 1));
        uint32_t j2 = j + 2;
        LAInfo<F> LA_ = la_step_new(p, la_init<F>(p, Z(j - 1)), Z(j));
        uint32_t j1 = j + 1;

        uint32_t wPeriodBegin = 0, wPeriodEnd = 0;
        bool PeriodDetected = false, PeriodDetected2 = false;

        for (; j2 < maxRef || j1 < maxRef; j1++, j2++) {
            LAInfo<F> NewLA;
            PeriodDetected = la_step(p, LA_, NewLA, Z(j1));
            if (PeriodDetected) {
                liNext = j1;
                wPeriodBegin = j1;
                wPeriodEnd = wPeriodBegin + Period;
                if (j1 + 1 >= maxRef) {
                    LA_ = la_init<F>(p, Z(j1));
                    j1 += 1;
                } else {
                    LA_ = la_step_new(p, la_init<F>(p, Z(j1)), Z(j1 + 1));
                    j1 += 2;
                }
                break;
            }
            LA_ = NewLA;
            if (j2 < maxRef) {
                LAInfo<F> NewLA2;
                PeriodDetected2 = la_step(p, LA_2, NewLA2, Z(j2));
                if (PeriodDetected2) {
                    liNext = j2;
                    wPeriodBegin = j2;
                    wPeriodEnd = wPeriodBegin + Period;
                    const uint32_t jp1 = j2 + 1;
                    if (jp1 >= maxRef) {
                        LA_2 = la_init<F>(p, Z(j2));
                        j2++;
                    } else {
                        LA_2 = la_step_new(p, la_init<F>(p, Z(j2)), Z(jp1));
                        j2 += 2;
                    }
                    break;
                }
                LA_2 = NewLA2;
            }
        }
        if (PeriodDetected2) {
            LA_ = LA_2;
            j = j2;
        } else if (PeriodDetected) {
            j = j1;
        } else {
            j = maxRef;
        }

        if (ThreadID == LastThread || (j >= Begin && j < End)) {
            StartIndex[ThreadID] = j;
        } else {
            const int64_t nextStart = StartIndex[NextThread];
            StartIndex[ThreadID] = nextStart;
            FinishIndex[ThreadID] = -1;
            continue;
        }

        bool returned = false;
        for (; j < maxRef; j++) {
            LAInfo<F> NewLA;
            PeriodDetected = la_step(p, LA_, NewLA, Z(j));
            if (!PeriodDetected && j < wPeriodEnd) {
                LA_ = NewLA;
                continue;
            }
            LA_.StepLength = j - wPeriodBegin;
            LA_.NextStageLAIndex = liNext;
            LAsPerThread[ThreadID].push_back(LA_);
            liNext = j;
            wPeriodBegin = j;
            wPeriodEnd = wPeriodBegin + Period;
            const uint32_t jp1 = j + 1;
            const bool detected = la_detect_period(p, NewLA, Z(jp1));
            if (detected || jp1 >= maxRef) {
                LA_ = la_init<F>(p, Z(j));
            } else {
                LA_ = la_step_new(p, la_init<F>(p, Z(j)), Z(jp1));
                j++;
            }
            if (j > End) {
                if (j >= maxRef)
                    break;
                if (NextThread < ThreadCount) {
                    const int64_t nextStart = StartIndex[NextThread];
                    if (nextStart < 0) {
                        returned = true;
                        break;
                    }
                    if ((int64_t)j == nextStart - 1) {
                        j++;
                        break;
                    } else if ((int64_t)j >= nextStart) {
                        NextThread++;
                    }
                }
            }
        }
        if (returned)
            continue;
        FinishIndex[ThreadID] = (int64_t)j;
        LA_.StepLength = j - wPeriodBegin;
        LA_.NextStageLAIndex = liNext;
        LastLAPerThread[ThreadID] = LA_;
    }

    // Starter (thread 0), LAReference.cpp:392-484.
    {
        const uint32_t threadBoundary = maxRef / (uint32_t)ThreadCount;
        size_t NextThread = 1;
        bool returned = false;
        for (; i < maxRef; i++) {
            LAInfo<F> NewLA;
            const bool PeriodDetected = la_step(p, LA, NewLA, Z(i));
            if (!PeriodDetected && i < PeriodEnd) {
                LA = NewLA;
                continue;
            }
            LA.StepLength = i - PeriodBegin;
            LA.NextStageLAIndex = nextIdx;
            T.las.push_back(LA);
            nextIdx = i;
            PeriodBegin = i;
            PeriodEnd = PeriodBegin + Period;
            const uint32_t ip1 = i + 1;
            const bool detected = la_detect_period(p, NewLA, Z(ip1));
            if (detected || ip1 >= maxRef) {
                LA = la_init<F>(p, Z(i));
            } else {
                LA = la_step_new(p, la_init<F>(p, Z(i)), Z(ip1));
                i++;
            }
            if (i > threadBoundary) {
                if (i >= maxRef)
                    break;
                if (NextThread < ThreadCount) {
                    const int64_t nextStart = StartIndex[NextThread];
                    if (nextStart < 0) {
                        returned = true;
                        break;
                    }
                    if ((int64_t)i == nextStart - 1) {
                        i++;
                        break;
                    } else if ((int64_t)i >= nextStart) {
                        NextThread++;
                    }
                }
            }
        }
        if (!returned) {
            FinishIndex[0] = (int64_t)i;
            LA.StepLength = i - PeriodBegin;
            LA.NextStageLAIndex = nextIdx;
            LastLAPerThread[0] = LA;
        }
    }

    // Stitch, LAReference.cpp:711-760.
    {
        size_t lastThreadToAdd = 0;
        size_t index = 0;
        size_t j = index;
        while ((index < ThreadCount - 1) && (FinishIndex[j] > StartIndex[index + 1]))
            index++;
        index++;
        for (; index < ThreadCount; index++) {
            const auto &threadData = LAsPerThread[index];
            T.las.insert(T.las.end(), threadData.begin(), threadData.end());
            if (FinishIndex[index] > StartIndex[index])
                lastThreadToAdd = index;
            j = index;
            while ((index < ThreadCount - 1) && (FinishIndex[j] > StartIndex[index + 1]))
                index++;
        }
        T.las.push_back(LastLAPerThread[lastThreadToAdd]);
    }
    T.stages[0].MacroItCount = la_size();
    T.las.push_back(la_init<F>(p, Z(maxRef)));
    return true;
}

} // namespace

struct fsh_la {
    int is64 = 0;
    LATable<float> t32;
    LATable<double> t64;
    std::vector<fs_la_hdr64_u32> packed64;
};

namespace {
template <class F> void pack_la(const LATable<F> &t, std::vector<fs_la_hdr32_u32> *o32, std::vector<fs_la_hdr64_u32> *o64)
{
    for (size_t k = 0; k < t.las.size(); k++) {
        const auto &s = t.las[k];
        if (o32) {
            fs_la_hdr32_u32 r;
            r.Ref = fs_cplx_hdr32{(float)s.Ref.re, (float)s.Ref.im, s.Ref.e};
            r.ZCoeff = fs_cplx_hdr32{(float)s.ZCoeff.re, (float)s.ZCoeff.im, s.ZCoeff.e};
            r.CCoeff = fs_cplx_hdr32{(float)s.CCoeff.re, (float)s.CCoeff.im, s.CCoeff.e};
            r.LAThreshold = fs_real_hdr32{(float)s.LAThreshold.m, s.LAThreshold.e};
            r.LAThresholdC = fs_real_hdr32{(float)s.LAThresholdC.m, s.LAThresholdC.e};
            r.MinMag = fs_real_hdr32{(float)s.MinMag.m, s.MinMag.e};
            r.StepLength = s.StepLength;
            r.NextStageLAIndex = s.NextStageLAIndex;
            o32->push_back(r);
        } else {
            fs_la_hdr64_u32 r;
            memset(&r, 0, sizeof(r));
            r.Ref = fs_cplx_hdr64{(double)s.Ref.re, (double)s.Ref.im, s.Ref.e, 0};
            r.ZCoeff = fs_cplx_hdr64{(double)s.ZCoeff.re, (double)s.ZCoeff.im, s.ZCoeff.e, 0};
            r.CCoeff = fs_cplx_hdr64{(double)s.CCoeff.re, (double)s.CCoeff.im, s.CCoeff.e, 0};
            r.LAThreshold = fs_real_hdr64{(double)s.LAThreshold.m, s.LAThreshold.e, 0};
            r.LAThresholdC = fs_real_hdr64{(double)s.LAThresholdC.m, s.LAThresholdC.e, 0};
            r.MinMag = fs_real_hdr64{(double)s.MinMag.m, s.MinMag.e, 0};
            r.StepLength = s.StepLength;
            r.NextStageLAIndex = s.NextStageLAIndex;
            o64->push_back(r);
        }
    }
}
} // namespace

extern "C" fsh_la *fsh_la_create_ex(const fsh_orbit *o, int host_threads, int use_small_exponents);
extern "C" fsh_la *fsh_la_create(const fsh_orbit *o, int host_threads) { return fsh_la_create_ex(o, host_threads, 0); }
extern "C" fsh_la *fsh_la_create_ex(const fsh_orbit *o, int host_threads, int use_small_exponents)
{
    auto la = std::make_unique<fsh_la>();
    la->is64 = o->is64;
    const int th = host_threads < 1 ? 1 : host_threads;
    if (o->is64) {
        LABuilder<double> b(o->d, la->t64);
        b.useSmallExponents = use_small_exponents != 0;
        b.generate(th);
        pack_la<double>(la->t64, nullptr, &la->packed64);
        auto &t = la->t64;
        t.packedStages.assign(t.stages.begin(), t.stages.begin() + std::min<size_t>(t.stages.size(), t.stageCount));
    } else {
        LABuilder<float> b(o->f, la->t32);
        b.generate(th);
        pack_la<float>(la->t32, &la->t32.packed32, nullptr);
        auto &t = la->t32;
        t.packedStages.assign(t.stages.begin(), t.stages.begin() + std::min<size_t>(t.stages.size(), t.stageCount));
    }
    return la.release();
}
extern "C" fsh_la *fsh_la_create_hdr32(const fsh_orbit *o, int host_threads)
{
    return o->is64 ? nullptr : fsh_la_create(o, host_threads);
}
extern "C" void fsh_la_destroy(fsh_la *l) { delete l; }
extern "C" int fsh_la_is64(const fsh_la *l) { return l->is64; }
extern "C" uint32_t fsh_la_count(const fsh_la *l)
{
    return l->is64 ? (uint32_t)l->packed64.size() : (uint32_t)l->t32.packed32.size();
}
extern "C" const void *fsh_la_data(const fsh_la *l)
{
    return l->is64 ? (const void *)l->packed64.data() : (const void *)l->t32.packed32.data();
}
extern "C" uint32_t fsh_la_stage_count(const fsh_la *l) { return l->is64 ? l->t64.stageCount : l->t32.stageCount; }
extern "C" const fs_la_stage_u32 *fsh_la_stages(const fsh_la *l)
{
    return l->is64 ? l->t64.packedStages.data() : l->t32.packedStages.data();
}
extern "C" int fsh_la_is_valid(const fsh_la *l) { return (l->is64 ? l->t64.isValid : l->t32.isValid) ? 1 : 0; }
extern "C" int fsh_la_use_at(const fsh_la *l) { return (l->is64 ? l->t64.useAT : l->t32.useAT) ? 1 : 0; }
// out: fs_at_hdr32_u32 (116 B) or fs_at_hdr64_u32 (232 B) depending on fsh_la_is64().
extern "C" void fsh_la_at(const fsh_la *l, void *outp)
{
    if (!l->is64) {
        const auto &a = l->t32.at;
        fs_at_hdr32_u32 *out = (fs_at_hdr32_u32 *)outp;
        auto R = [](hreal<float> h) { return fs_real_hdr32{h.m, h.e}; };
        auto C = [](hcplx<float> c) { return fs_cplx_hdr32{c.re, c.im, c.e}; };
        out->StepLength = a.StepLength;
        out->ThresholdC = R(a.ThresholdC);
        out->SqrEscapeRadius = R(a.SqrEscapeRadius);
        out->RefC = C(a.RefC);
        out->ZCoeff = C(a.ZCoeff);
        out->CCoeff = C(a.CCoeff);
        out->InvZCoeff = C(a.InvZCoeff);
        out->CCoeffSqrInvZCoeff = C(a.CCoeffSqrInvZCoeff);
        out->CCoeffInvZCoeff = C(a.CCoeffInvZCoeff);
        out->CCoeffNormSqr = R(a.CCoeffNormSqr);
        out->RefCNormSqr = R(a.RefCNormSqr);
        out->factor = R(a.factor);
    } else {
        const auto &a = l->t64.at;
        fs_at_hdr64_u32 *out = (fs_at_hdr64_u32 *)outp;
        memset(out, 0, sizeof(*out));
        auto R = [](hreal<double> h) { return fs_real_hdr64{h.m, h.e, 0}; };
        auto C = [](hcplx<double> c) { return fs_cplx_hdr64{c.re, c.im, c.e, 0}; };
        out->StepLength = a.StepLength;
        out->ThresholdC = R(a.ThresholdC);
        out->SqrEscapeRadius = R(a.SqrEscapeRadius);
        out->RefC = C(a.RefC);
        out->ZCoeff = C(a.ZCoeff);
        out->CCoeff = C(a.CCoeff);
        out->InvZCoeff = C(a.InvZCoeff);
        out->CCoeffSqrInvZCoeff = C(a.CCoeffSqrInvZCoeff);
        out->CCoeffInvZCoeff = C(a.CCoeffInvZCoeff);
        out->CCoeffNormSqr = R(a.CCoeffNormSqr);
        out->RefCNormSqr = R(a.RefCNormSqr);
        out->factor = R(a.factor);
    }
}

// ------------------------------------------------------------------ BLA table
// BLAS<uint32_t, HDRFloat<float>>::Init, BLAS.cpp:25-255 with BLA<T> helpers BLA.cuh:7-110.  The reference
// fills level 2 and merges upwards on several threads; every element is a pure function of the orbit, so a
// sequential fill gives the same table.
namespace {

// (BlaRec<F> and the record arithmetic: csrc/bla_math.hpp, shared with the device builder)

template <class F> struct BlaBuilder {
    const OrbitT<F> &ob;
    std::vector<std::vector<BlaRec<F>>> B;
    std::vector<size_t> elementsPerLevel;
    int32_t LM2 = 0;
    size_t L = 0;
    static constexpr size_t kFirstLevel = 2; // BLAS.h:22

    explicit BlaBuilder(const OrbitT<F> &o) : ob(o) {}

    // BLAS::CreateOneStep, BLAS.cpp:74-93
    BlaRec<F> one_step(size_t m, hreal<F> epsilon) const { return bla_one_step<F>(hc_from_hr(ob.x[m], ob.y[m]), epsilon); }

    // BLAS::MergeTwoBlas, BLAS.cpp:25-47
    BlaRec<F> merge(const BlaRec<F> &x, const BlaRec<F> &y, hreal<F> blaSize) const { return bla_merge<F>(x, y, blaSize); }

    // BLAS::CreateLStep, BLAS.cpp:49-72
    BlaRec<F> l_step(size_t level, size_t m, hreal<F> blaSize, hreal<F> epsilon) const
    {
        if (level == 0)
            return one_step(m, epsilon);
        const size_t m2 = m << 1, mx = m2 - 1, my = m2, levelm1 = level - 1;
        if (my <= elementsPerLevel[levelm1]) {
            const BlaRec<F> x = l_step(levelm1, mx, blaSize, epsilon);
            const BlaRec<F> y = l_step(levelm1, my, blaSize, epsilon);
            return merge(x, y, blaSize);
        }
        return l_step(levelm1, mx, blaSize, epsilon);
    }

    // BLAS::Init, BLAS.cpp:212-255
    void init(size_t InM, hreal<F> blaSize)
    {
        const hreal<F> precision = hr_div(hr_from_number<F>(F(1)), hr_from_mant<F>(F(8388608))); // T(1)/T{1L<<23}
        size_t m = InM - 1;
        if (InM == 0 || m == 0)
            return;
        elementsPerLevel.clear();
        for (; m > 1; m = (m + 1) >> 1)
            elementsPerLevel.push_back(m);
        elementsPerLevel.push_back(m);
        L = elementsPerLevel.size();
        B.clear();
        B.resize(L);
        LM2 = (int32_t)L - 2;
        if (LM2 < 0)
            LM2 = 0;
        if (kFirstLevel >= elementsPerLevel.size())
            return;
        for (size_t l = kFirstLevel; l < B.size(); l++)
            B[l].resize(elementsPerLevel[l]);
        // InitInternal, BLAS.cpp:97-139
        const size_t elements = elementsPerLevel[kFirstLevel] + 1;
        for (size_t mm = 1; mm < elements; mm++)
            B[kFirstLevel][mm - 1] = l_step(kFirstLevel, mm, blaSize, precision);
        // Merge, BLAS.cpp:141-210
        size_t src = kFirstLevel;
        const size_t maxLevel = elementsPerLevel.size() - 1;
        for (size_t elementsSrc = elementsPerLevel[src]; src < maxLevel && elementsSrc > 1; src++) {
            const size_t dst = src + 1;
            const size_t elementsDst = elementsPerLevel[dst];
            for (size_t k = 0; k < elementsDst; k++) {
                const size_t mx = k << 1, my = mx + 1;
                if (my < elementsSrc)
                    B[dst][k] = merge(B[src][mx], B[src][my], blaSize);
                else
                    B[dst][k] = B[src][mx];
            }
            elementsSrc = elementsDst;
        }
    }
};

} // namespace

struct fsh_bla {
    int is64 = 0;
    std::vector<std::vector<fs_bla_hdr32>> levels32;
    std::vector<std::vector<fs_bla_hdr64>> levels64;
    std::vector<const void *> ptrs;
    std::vector<uint64_t> sizes;
    int32_t lm2 = 0;
};

extern "C" fsh_bla *fsh_bla_create(const fsh_orbit *o)
{
    auto r = std::make_unique<fsh_bla>();
    r->is64 = o->is64;
    if (!o->is64) {
        BlaBuilder<float> b(o->f);
        b.init(o->f.x.size(), o->f.maxRadius);
        r->lm2 = b.LM2;
        r->levels32.resize(b.B.size());
        for (size_t l = 0; l < b.B.size(); l++) {
            r->levels32[l].resize(b.B[l].size());
            for (size_t k = 0; k < b.B[l].size(); k++) {
                const auto &s = b.B[l][k];
                auto R = [](hreal<float> h) { return fs_real_hdr32{h.m, h.e}; };
                r->levels32[l][k] = fs_bla_hdr32{R(s.r2), R(s.Ax), R(s.Ay), R(s.Bx), R(s.By), s.l};
            }
            r->ptrs.push_back(r->levels32[l].empty() ? nullptr : (const void *)r->levels32[l].data());
            r->sizes.push_back(r->levels32[l].size());
        }
    } else {
        BlaBuilder<double> b(o->d);
        b.init(o->d.x.size(), o->d.maxRadius);
        r->lm2 = b.LM2;
        r->levels64.resize(b.B.size());
        for (size_t l = 0; l < b.B.size(); l++) {
            r->levels64[l].resize(b.B[l].size());
            for (size_t k = 0; k < b.B[l].size(); k++) {
                const auto &s = b.B[l][k];
                auto R = [](hreal<double> h) { return fs_real_hdr64{h.m, h.e, 0}; };
                r->levels64[l][k] = fs_bla_hdr64{R(s.r2), R(s.Ax), R(s.Ay), R(s.Bx), R(s.By), s.l, 0};
            }
            r->ptrs.push_back(r->levels64[l].empty() ? nullptr : (const void *)r->levels64[l].data());
            r->sizes.push_back(r->levels64[l].size());
        }
    }
    return r.release();
}
extern "C" fsh_bla *fsh_bla_create_hdr32(const fsh_orbit *o) { return o->is64 ? nullptr : fsh_bla_create(o); }
extern "C" void fsh_bla_destroy(fsh_bla *b) { delete b; }
extern "C" int32_t fsh_bla_num_levels(const fsh_bla *b) { return (int32_t)b->ptrs.size(); }
extern "C" int32_t fsh_bla_lm2(const fsh_bla *b) { return b->lm2; }
extern "C" const void *const *fsh_bla_level_ptrs(const fsh_bla *b) { return b->ptrs.data(); }
extern "C" const uint64_t *fsh_bla_level_sizes(const fsh_bla *b) { return b->sizes.data(); }

// ------------------------------------------------------------------ plain double (Cpu64PerturbedBLA / Gpu1x64PerturbedBLA)
// PerturbationResults<uint32_t,double,Disable> + BLAS<uint32_t,double>: the floatOrDouble branches of
// AddPerturbationReferencePointST (RefOrbitCalc.cpp:481-488,524-530,564-604,617-622) and BLAS.cpp with T = double.
struct fsh_orbit_f64 {
    std::vector<fs_orbit_f64> z; // entry 0 = {0,0}
    std::vector<uint8_t> bad;    // PerturbExtras::Bad flag per entry (RefOrbitCalc.cpp:550-562,625-627)
    std::vector<fs_orbit_f64_bad> packed_bad;
    std::vector<fs_orbit_f32_bad> packed_f32_bad;
    uint64_t period = 0;
    double maxRadius = 0;
    Mp cx, cy;
    std::vector<std::vector<fs_bla_f64>> levels;
    std::vector<const void *> ptrs;
    std::vector<uint64_t> sizes;
    int32_t lm2 = 0;
};

extern "C" fsh_orbit_f64 *fsh_orbit_f64_create(const fsh_view *vwp, uint64_t max_iter, int periodicity)
{
    const fsh_view &vw = *vwp;
    auto ob = std::make_unique<fsh_orbit_f64>();
    mpf_set_default_prec(vw.prec_bits);
    {
        Mp two = Mp::from_ui(2);
        ob->cx = (vw.maxX + vw.minX) / two;
        ob->cy = (vw.maxY + vw.minY) / two;
        Mp delta = vw.maxY - vw.minY;
        ob->maxRadius = mpf_get_d(delta.v) / 2.0; // T{delta} / T{2.0f}
    }
    ob->z.push_back(fs_orbit_f64{0.0, 0.0});
    ob->bad.push_back(0);
    mpf_t cx, cy, zx, zy, zx2, t1, t2;
    mpf_init(cx);
    mpf_set(cx, ob->cx.v);
    mpf_init(cy);
    mpf_set(cy, ob->cy.v);
    mpf_init(zx);
    mpf_init(zy);
    mpf_init(zx2);
    mpf_init(t1);
    mpf_init(t2);
    double dzdcX = 1.0, dzdcY = 0.0;
    const double cx_cast = mpf_get_d(cx), cy_cast = mpf_get_d(cy);
    mpf_set(zx, cx);
    mpf_set(zy, cy);
    for (uint64_t i = 0; i < max_iter; i++) {
        mpf_mul_2exp(zx2, zx, 1);
        const double double_zx = mpf_get_d(zx), double_zy = mpf_get_d(zy);
        ob->z.push_back(fs_orbit_f64{double_zx, double_zy});
        {
            const double small_float = 1.1754944e-38, glitch = 0.0000001;
            const double norm = (double_zx * double_zx + double_zy * double_zy) * glitch;
            ob->bad.push_back((std::fabs(double_zx) <= small_float || std::fabs(double_zy) <= small_float ||
                               norm <= small_float)
                                  ? 1
                                  : 0);
        }
        if (periodicity) {
            const double n2 = std::max(std::fabs(double_zx), std::fabs(double_zy));
            const double r0 = std::max(std::fabs(dzdcX), std::fabs(dzdcY));
            const double n3 = ob->maxRadius * r0 * 2.0;
            if (n2 < n3) {
                ob->period = ob->z.size();
                break;
            } else {
                const double dzdcXOrig = dzdcX;
                dzdcX = 2.0 * (double_zx * dzdcX - double_zy * dzdcY) + 1.0;
                dzdcY = 2.0 * (double_zx * dzdcY + double_zy * dzdcXOrig);
            }
        }
        mpf_mul(t1, zx, zx);
        mpf_mul(t2, zy, zy);
        mpf_sub(zx, t1, t2);
        mpf_add(zx, zx, cx);
        mpf_mul(zy, zx2, zy);
        mpf_add(zy, zy, cy);
        const double tempZX = double_zx + cx_cast, tempZY = double_zy + cy_cast;
        const double zn = tempZX * tempZX + tempZY * tempZY;
        if (zn > 256.0)
            break;
    }
    mpf_clear(cx);
    mpf_clear(cy);
    mpf_clear(zx);
    mpf_clear(zy);
    mpf_clear(zx2);
    mpf_clear(t1);
    mpf_clear(t2);

    ob->bad.back() = 0; // results->SetBad(false), RefOrbitCalc.cpp:625-627

    // BLAS<uint32_t,double>::Init(count, maxRadius), BLAS.cpp:25-255 with plain double (BLA.cuh:40-91)
    {
        const auto &Z = ob->z;
        const double blaSize = ob->maxRadius;
        const double epsilon = 1.0 / 8388608.0; // T(1) / T{1L << 23}
        std::vector<size_t> epl;
        size_t m = Z.size() - 1;
        if (Z.size() != 0 && m != 0) {
            for (; m > 1; m = (m + 1) >> 1)
                epl.push_back(m);
            epl.push_back(m);
            const size_t L = epl.size();
            ob->levels.resize(L);
            ob->lm2 = (int32_t)L - 2 < 0 ? 0 : (int32_t)L - 2;
            auto one_step = [&](size_t mm) {
                const double RealA = Z[mm].x * 2, ImagA = Z[mm].y * 2;
                const double mA = std::sqrt(RealA * RealA + ImagA * ImagA);
                const double r = mA * epsilon;
                return fs_bla_f64{r * r, RealA, ImagA, 1.0, 0.0, 1, 0};
            };
            auto merge = [&](const fs_bla_f64 &x, const fs_bla_f64 &y) {
                const int32_t l = x.l + y.l;
                const double RealA = y.Ax * x.Ax - y.Ay * x.Ay;
                const double ImagA = y.Ax * x.Ay + y.Ay * x.Ax;
                const double RealB = y.Ax * x.Bx - y.Ay * x.By + y.Bx;
                const double ImagB = y.Ax * x.By + y.Ay * x.Bx + y.By;
                const double xA = std::sqrt(x.Ax * x.Ax + x.Ay * x.Ay);
                const double xB = std::sqrt(x.Bx * x.Bx + x.By * x.By);
                const double tempR = (std::sqrt(y.r2) - xB * blaSize) / xA;
                const double mx = (0.0 > tempR) ? 0.0 : tempR;           // HdrMaxReduced(T(0), tempR)
                const double sx = std::sqrt(x.r2);
                const double r = (sx < mx) ? sx : mx;                    // HdrMinPositiveReduced
                return fs_bla_f64{r * r, RealA, ImagA, RealB, ImagB, l, 0};
            };
            std::function<fs_bla_f64(size_t, size_t)> l_step = [&](size_t level, size_t mm) -> fs_bla_f64 {
                if (level == 0)
                    return one_step(mm);
                const size_t m2 = mm << 1, mx = m2 - 1, my = m2;
                if (my <= epl[level - 1])
                    return merge(l_step(level - 1, mx), l_step(level - 1, my));
                return l_step(level - 1, mx);
            };
            if (2 < epl.size()) {
                for (size_t l = 2; l < L; l++)
                    ob->levels[l].resize(epl[l]);
                const size_t elements = epl[2] + 1;
                for (size_t mm = 1; mm < elements; mm++)
                    ob->levels[2][mm - 1] = l_step(2, mm);
                size_t src = 2;
                const size_t maxLevel = epl.size() - 1;
                for (size_t elementsSrc = epl[src]; src < maxLevel && elementsSrc > 1; src++) {
                    const size_t dst = src + 1, elementsDst = epl[dst];
                    for (size_t k = 0; k < elementsDst; k++) {
                        const size_t mx = k << 1, my = mx + 1;
                        ob->levels[dst][k] = my < elementsSrc ? merge(ob->levels[src][mx], ob->levels[src][my])
                                                              : ob->levels[src][mx];
                    }
                    elementsSrc = elementsDst;
                }
            }
        }
        for (auto &lv : ob->levels) {
            ob->ptrs.push_back(lv.empty() ? nullptr : (const void *)lv.data());
            ob->sizes.push_back(lv.size());
        }
    }
    return ob.release();
}
// PerturbExtras::Bad form of the double orbit + its binary32 copy (Gpu1x32PerturbedScaled)
extern "C" const fs_orbit_f64_bad *fsh_orbit_f64_data_bad(fsh_orbit_f64 *o)
{
    if (o->packed_bad.size() != o->z.size()) {
        o->packed_bad.resize(o->z.size());
        for (size_t i = 0; i < o->z.size(); i++)
            o->packed_bad[i] = fs_orbit_f64_bad{o->bad[i], 0u, o->z[i].x, o->z[i].y};
    }
    return o->packed_bad.data();
}
extern "C" const fs_orbit_f32_bad *fsh_orbit_f64_data_f32_bad(fsh_orbit_f64 *o)
{
    if (o->packed_f32_bad.size() != o->z.size()) {
        o->packed_f32_bad.resize(o->z.size());
        for (size_t i = 0; i < o->z.size(); i++)
            o->packed_f32_bad[i] = fs_orbit_f32_bad{o->bad[i] != 0 ? 1u : 0u, 0u, (float)o->z[i].x, (float)o->z[i].y};
    }
    return o->packed_f32_bad.data();
}

extern "C" void fsh_orbit_f64_destroy(fsh_orbit_f64 *o) { delete o; }
extern "C" uint64_t fsh_orbit_f64_count(const fsh_orbit_f64 *o) { return o->z.size(); }
extern "C" uint64_t fsh_orbit_f64_period(const fsh_orbit_f64 *o) { return o->period; }
extern "C" const fs_orbit_f64 *fsh_orbit_f64_data(const fsh_orbit_f64 *o) { return o->z.data(); }
extern "C" int32_t fsh_orbit_f64_bla_num_levels(const fsh_orbit_f64 *o) { return (int32_t)o->ptrs.size(); }
extern "C" int32_t fsh_orbit_f64_bla_lm2(const fsh_orbit_f64 *o) { return o->lm2; }
extern "C" const void *const *fsh_orbit_f64_bla_level_ptrs(const fsh_orbit_f64 *o) { return o->ptrs.data(); }
extern "C" const uint64_t *fsh_orbit_f64_bla_level_sizes(const fsh_orbit_f64 *o) { return o->sizes.data(); }
// {dx, dy, centerX, centerY} as doubles (Fractal.cpp:2230-2238 with T = double: mpf_get_d, no reduction).
extern "C" void fsh_view_coords_perturb_f64(const fsh_view *v, const fsh_orbit_f64 *o, uint32_t w_aa, uint32_t h_aa,
                                            double out[4])
{
    mpf_set_default_prec(v->prec_bits);
    Mp dx = (v->maxX - v->minX) / Mp::from_ui(w_aa);
    Mp dy = (v->maxY - v->minY) / Mp::from_ui(h_aa);
    Mp cX = o->cx - v->minX;
    Mp cY = o->cy - v->maxY;
    out[0] = mpf_get_d(dx.v);
    out[1] = mpf_get_d(dy.v);
    out[2] = mpf_get_d(cX.v);
    out[3] = mpf_get_d(cY.v);
}

// ------------------------------------------------------------------ plain float / double orbit + LAv2 table
// Inputs of Gpu1x32PerturbedLAv2* (T = float), Gpu1x64PerturbedLAv2* (T = double) and, converted field-wise,
// Gpu2x32PerturbedLAv2* (T = CudaDblflt<MattDblflt>, built as double: DoubleTo2x32Converter, Fractal.cpp:2771-2772).
// Orbit: the floatOrDouble arms of AddPerturbationReferencePointST (RefOrbitCalc.cpp:481-488,524-530,564-604,617-622)
// with T = float | double; table: LAReference<uint32_t,T,T,Disable> through the builder above with F = plain<T>.
namespace {
// runOneIter for a plain T (PerturbationResultsHelpers.h:51-58; HdrReduce is the identity)
template <class T> void rc_one_iter_plain(T &zx, T &zy, T cxLow, T cyLow)
{
    const T zx_old = zx;
    zx = zx * zx - zy * zy + cxLow;
    zy = T(2.0f) * zx_old * zy + cyLow;
}

// compression_exp >= 0: PerturbExtras::SimpleCompression through RefOrbitCompressor<IterType, T, SimpleCompression>
// (PerturbationResults.cpp:2334-2381) in plain T arithmetic.
template <class T>
void build_plain_orbit(const fsh_view &vw, uint64_t max_iter, int periodicity, PlainOrbit<T> &ob, int compression_exp = -1)
{
    mpf_set_default_prec(vw.prec_bits);
    {
        Mp two = Mp::from_ui(2);
        ob.cx = (vw.maxX + vw.minX) / two;
        ob.cy = (vw.maxY + vw.minY) / two;
        Mp delta = vw.maxY - vw.minY;
        ob.maxRadius = preal<T>{(T)((T)mpf_get_d(delta.v) / T(2.0f))}; // T{delta} / T{2.0f}, PerturbationResults.cpp:823-824
    }
    ob.x.push_back(preal<T>{T(0)});
    ob.y.push_back(preal<T>{T(0)});
    mpf_t cx, cy, zx, zy, zx2, t1, t2;
    mpf_init(cx);
    mpf_set(cx, ob.cx.v);
    mpf_init(cy);
    mpf_set(cy, ob.cy.v);
    mpf_init(zx);
    mpf_init(zy);
    mpf_init(zx2);
    mpf_init(t1);
    mpf_init(t2);
    T dzdcX = T(1), dzdcY = T(0);
    const T cx_cast = (T)mpf_get_d(cx), cy_cast = (T)mpf_get_d(cy);
    ob.orbitXLow = cx_cast; // m_OrbitXLow = T{cx}, PerturbationResults.cpp:852-853
    ob.orbitYLow = cy_cast;
    ob.compressed = compression_exp >= 0;
    ob.wp_index.assign(1, 0);
    ob.wp_x.assign(1, T(0));
    ob.wp_y.assign(1, T(0));
    uint64_t count = 1; // m_UncompressedItersInOrbit
    T rc_zx = ob.orbitXLow, rc_zy = ob.orbitYLow;
    const T rc_err = static_cast<T>(std::pow(10, compression_exp));
    mpf_set(zx, cx);
    mpf_set(zy, cy);
    for (uint64_t i = 0; i < max_iter; i++) {
        mpf_mul_2exp(zx2, zx, 1);
        const T double_zx = (T)mpf_get_d(zx), double_zy = (T)mpf_get_d(zy);
        if (!ob.compressed) {
            ob.x.push_back(preal<T>{double_zx});
            ob.y.push_back(preal<T>{double_zy});
        } else {
            // MaybeAddCompressedIteration({double_zx, double_zy, i + 1})
            const T errX = rc_zx - double_zx, errY = rc_zy - double_zy;
            const T norm_z = double_zx * double_zx + double_zy * double_zy;
            const T err = (errX * errX + errY * errY) * rc_err;
            if (err >= norm_z) {
                ob.wp_index.push_back(i + 1);
                ob.wp_x.push_back(double_zx);
                ob.wp_y.push_back(double_zy);
                rc_zx = double_zx;
                rc_zy = double_zy;
            }
            rc_one_iter_plain(rc_zx, rc_zy, ob.orbitXLow, ob.orbitYLow);
        }
        count++;
        if (periodicity) {
            const T n2 = std::max(std::fabs(double_zx), std::fabs(double_zy));
            const T r0 = std::max(std::fabs(dzdcX), std::fabs(dzdcY));
            const T n3 = ob.maxRadius.m * r0 * T(2);
            if (n2 < n3) {
                ob.period = count;
                break;
            } else {
                const T dzdcXOrig = dzdcX;
                dzdcX = T(2) * (double_zx * dzdcX - double_zy * dzdcY) + T(1);
                dzdcY = T(2) * (double_zx * dzdcY + double_zy * dzdcXOrig);
            }
        }
        mpf_mul(t1, zx, zx);
        mpf_mul(t2, zy, zy);
        mpf_sub(zx, t1, t2);
        mpf_add(zx, zx, cx);
        mpf_mul(zy, zx2, zy);
        mpf_add(zy, zy, cy);
        const T tempZX = double_zx + cx_cast, tempZY = double_zy + cy_cast;
        const T zn = tempZX * tempZX + tempZY * tempZY;
        if (zn > T(256))
            break;
    }
    mpf_clear(cx);
    mpf_clear(cy);
    mpf_clear(zx);
    mpf_clear(zy);
    mpf_clear(zx2);
    mpf_clear(t1);
    mpf_clear(t2);
    if (ob.compressed) {
        // the orbit every host consumer (the LA builder) sees: RuntimeDecompressor::GetCompressedComplex
        ob.x.assign(count, preal<T>{T(0)});
        ob.y.assign(count, preal<T>{T(0)});
        for (size_t k = 0; k < ob.wp_index.size(); k++) {
            const uint64_t i0 = ob.wp_index[k];
            const uint64_t i1 = k + 1 < ob.wp_index.size() ? ob.wp_index[k + 1] : count;
            T zx_ = ob.wp_x[k], zy_ = ob.wp_y[k];
            for (uint64_t i = i0; i < i1; i++) {
                ob.x[i] = preal<T>{zx_};
                ob.y[i] = preal<T>{zy_};
                rc_one_iter_plain(zx_, zy_, ob.orbitXLow, ob.orbitYLow);
            }
        }
    }
}

template <class T> struct plain_recs;
template <> struct plain_recs<float> {
    using orbit_rc = fs_orbit_f32_rc;
    using orbit = fs_orbit_f32;
    using la = fs_la_f32_u32;
    using at = fs_at_f32_u32;
    using cplx = fs_cplx_f32;
};
template <> struct plain_recs<double> {
    using orbit_rc = fs_orbit_f64_rc;
    using orbit = fs_orbit_f64;
    using la = fs_la_f64_u32;
    using at = fs_at_f64_u32;
    using cplx = fs_cplx_f64;
};

template <class T> struct PlainInputs {
    PlainOrbit<T> ob;
    LATable<plain<T>> t;
    std::vector<typename plain_recs<T>::orbit> orbit_packed;
    std::vector<typename plain_recs<T>::orbit_rc> rc_packed; // GPUReferenceIter<T, SimpleCompression>[]
    std::vector<typename plain_recs<T>::la> la_packed;
    typename plain_recs<T>::at at_packed;

    void build(const fsh_view &vw, uint64_t max_iter, int periodicity, int host_threads, int compression_exp)
    {
        build_plain_orbit<T>(vw, max_iter, periodicity, ob, compression_exp);
        finish(host_threads);
    }
    // everything derived from the orbit (packed records, LA table, ATInfo): also what a loaded ".im" orbit goes through
    void finish(int host_threads)
    {
        using R = plain_recs<T>;
        orbit_packed.resize(ob.x.size());
        for (size_t i = 0; i < ob.x.size(); i++)
            orbit_packed[i] = typename R::orbit{ob.x[i].m, ob.y[i].m};
        if (ob.compressed) {
            rc_packed.resize(ob.wp_index.size());
            for (size_t k = 0; k < ob.wp_index.size(); k++)
                rc_packed[k] = typename R::orbit_rc{ob.wp_index[k], ob.wp_x[k], ob.wp_y[k]};
        }
        LABuilder<plain<T>> b(ob, t);
        b.generate(host_threads < 1 ? 1 : host_threads);
        auto C = [](pcplx<T> c) { return typename R::cplx{c.re, c.im}; };
        la_packed.resize(t.las.size());
        for (size_t k = 0; k < t.las.size(); k++) {
            const auto &s = t.las[k];
            typename R::la r;
            memset(&r, 0, sizeof(r));
            r.Ref = C(s.Ref);
            r.ZCoeff = C(s.ZCoeff);
            r.CCoeff = C(s.CCoeff);
            r.LAThreshold = s.LAThreshold.m;
            r.LAThresholdC = s.LAThresholdC.m;
            r.MinMag = s.MinMag.m;
            r.StepLength = s.StepLength;
            r.NextStageLAIndex = s.NextStageLAIndex;
            la_packed[k] = r;
        }
        t.packedStages.assign(t.stages.begin(), t.stages.begin() + std::min<size_t>(t.stages.size(), t.stageCount));
        const auto &a = t.at;
        memset(&at_packed, 0, sizeof(at_packed));
        at_packed.StepLength = a.StepLength;
        at_packed.ThresholdC = a.ThresholdC.m;
        at_packed.SqrEscapeRadius = a.SqrEscapeRadius.m;
        at_packed.RefC = C(a.RefC);
        at_packed.ZCoeff = C(a.ZCoeff);
        at_packed.CCoeff = C(a.CCoeff);
        at_packed.InvZCoeff = C(a.InvZCoeff);
        at_packed.CCoeffSqrInvZCoeff = C(a.CCoeffSqrInvZCoeff);
        at_packed.CCoeffInvZCoeff = C(a.CCoeffInvZCoeff);
        at_packed.CCoeffNormSqr = a.CCoeffNormSqr.m;
        at_packed.RefCNormSqr = a.RefCNormSqr.m;
        at_packed.factor = a.factor.m;
    }
};
} // namespace

struct fsh_plain {
    int kind = 0; // 0 float, 1 double
    PlainInputs<float> f;
    PlainInputs<double> d;
};

extern "C" fsh_plain *fsh_plain_create_ex(const fsh_view *v, int kind, uint64_t max_iter, int periodicity, int host_threads,
                                          int compression_exp)
{
    if (kind != 0 && kind != 1)
        return nullptr;
    auto h = std::make_unique<fsh_plain>();
    h->kind = kind;
    if (kind == 0)
        h->f.build(*v, max_iter, periodicity, host_threads, compression_exp);
    else
        h->d.build(*v, max_iter, periodicity, host_threads, compression_exp);
    return h.release();
}
extern "C" fsh_plain *fsh_plain_create(const fsh_view *v, int kind, uint64_t max_iter, int periodicity, int host_threads)
{
    return fsh_plain_create_ex(v, kind, max_iter, periodicity, host_threads, -1);
}
extern "C" void fsh_plain_destroy(fsh_plain *h) { delete h; }

// ".im" files with a reference orbit for the non-ExtendedRange types (float: "Sharks:)" magic, double: Imagina's;
// ReferenceHeader::ExtendedRange = false; RefOrbitCalc.cpp:3039-3115 / :3386-3412, PerturbationResults.cpp:2047-2075, 2177-2183)
extern "C" int fsh_plain_save_im(const fsh_plain *h, uint64_t num_iterations, int compression_exp, const char *path, int exp_bytes)
{
    if (h->kind == 0)
        return save_im_orbit<plain<float>>(h->f.ob, preal<float>{h->f.ob.orbitXLow}, preal<float>{h->f.ob.orbitYLow},
                                           num_iterations, compression_exp, path, exp_bytes);
    return save_im_orbit<plain<double>>(h->d.ob, preal<double>{h->d.ob.orbitXLow}, preal<double>{h->d.ob.orbitYLow},
                                        num_iterations, compression_exp, path, exp_bytes);
}

static fsh_plain *plain_load_im(FILE *f, uint64_t *iteration_limit, int host_threads)
{
    uint64_t header[4];
    ImHR hh;
    uint64_t limit = 0;
    std::unique_ptr<fsh_plain> out;
    if (fread(header, 8, 4, f) == 4 && (header[0] == kImMagic || header[0] == kSharksMagic) && header[3] != 0 &&
        fseek(f, (long)header[2], SEEK_SET) == 0 && fread(&hh, sizeof(hh), 1, f) == 1 && fread(&limit, 8, 1, f) == 1) {
        // (untrusted input, as in orbit_load_im: bound the precision GMP is asked for)
        if (hh.exp < -(int64_t)kMaxImPrecisionBits)
            return nullptr;
        const uint64_t precision = (uint64_t)(-std::min<int64_t>(0, hh.exp)) + 120u;
        mpf_set_default_prec(precision);
        Mp X(precision, 0), Y(precision, 0);
        const long at = ftell(f);
        int width_ok = 0;
        for (int exp_bytes : {4, 8}) {
            fseek(f, at, SEEK_SET);
            if (im_read_mpf(f, X.v, exp_bytes) && im_read_mpf(f, Y.v, exp_bytes) && (uint64_t)ftell(f) == header[3]) {
                width_ok = exp_bytes;
                break;
            }
        }
        if (width_ok) {
            out = std::make_unique<fsh_plain>();
            out->kind = header[0] == kImMagic ? 1 : 0; // T double <-> Imagina's magic, float <-> "Sharks:)" (:3386-3412)
            bool ok;
            if (out->kind == 1) {
                out->d.ob.cx = X, out->d.ob.cy = Y;
                ok = load_im_orbit<plain<double>>(f, header[3], out->d.ob, limit, ImHR{hh.mantissa, hh.exp});
                if (ok)
                    out->d.finish(host_threads);
            } else {
                out->f.ob.cx = X, out->f.ob.cy = Y;
                ok = load_im_orbit<plain<float>>(f, header[3], out->f.ob, limit, ImHR{hh.mantissa, hh.exp});
                if (ok)
                    out->f.finish(host_threads);
            }
            if (!ok)
                out.reset();
            else if (iteration_limit)
                *iteration_limit = limit;
        }
    }
    return out.release();
}

extern "C" fsh_plain *fsh_plain_load_im(const char *path, uint64_t *iteration_limit, int host_threads)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return nullptr;
    fsh_plain *out = nullptr;
    try { // nothing may unwind through the C boundary
        out = plain_load_im(f, iteration_limit, host_threads);
    } catch (...) {
        out = nullptr;
    }
    fclose(f);
    return out;
}
extern "C" int fsh_plain_kind(const fsh_plain *h) { return h->kind; }
#define FS_PLAIN_GET(EXPR_F, EXPR_D) (h->kind == 0 ? (EXPR_F) : (EXPR_D))
extern "C" uint64_t fsh_plain_orbit_count(const fsh_plain *h) { return FS_PLAIN_GET(h->f.ob.x.size(), h->d.ob.x.size()); }
extern "C" uint64_t fsh_plain_orbit_period(const fsh_plain *h) { return FS_PLAIN_GET(h->f.ob.period, h->d.ob.period); }
extern "C" const void *fsh_plain_orbit_data(const fsh_plain *h)
{
    return FS_PLAIN_GET((const void *)h->f.orbit_packed.data(), (const void *)h->d.orbit_packed.data());
}
extern "C" int fsh_plain_is_compressed(const fsh_plain *h) { return FS_PLAIN_GET(h->f.ob.compressed, h->d.ob.compressed) ? 1 : 0; }
extern "C" uint64_t fsh_plain_compressed_count(const fsh_plain *h)
{
    return FS_PLAIN_GET(h->f.rc_packed.size(), h->d.rc_packed.size());
}
extern "C" const void *fsh_plain_compressed_data(const fsh_plain *h)
{
    return FS_PLAIN_GET((const void *)h->f.rc_packed.data(), (const void *)h->d.rc_packed.data());
}
extern "C" void fsh_plain_orbit_low(const fsh_plain *h, void *out)
{
    if (h->kind == 0) {
        ((float *)out)[0] = h->f.ob.orbitXLow;
        ((float *)out)[1] = h->f.ob.orbitYLow;
    } else {
        ((double *)out)[0] = h->d.ob.orbitXLow;
        ((double *)out)[1] = h->d.ob.orbitYLow;
    }
}
extern "C" uint32_t fsh_plain_la_count(const fsh_plain *h)
{
    return (uint32_t)FS_PLAIN_GET(h->f.la_packed.size(), h->d.la_packed.size());
}
extern "C" const void *fsh_plain_la_data(const fsh_plain *h)
{
    return FS_PLAIN_GET((const void *)h->f.la_packed.data(), (const void *)h->d.la_packed.data());
}
extern "C" uint32_t fsh_plain_la_stage_count(const fsh_plain *h) { return FS_PLAIN_GET(h->f.t.stageCount, h->d.t.stageCount); }
extern "C" const fs_la_stage_u32 *fsh_plain_la_stages(const fsh_plain *h)
{
    return FS_PLAIN_GET(h->f.t.packedStages.data(), h->d.t.packedStages.data());
}
extern "C" int fsh_plain_la_is_valid(const fsh_plain *h) { return FS_PLAIN_GET(h->f.t.isValid, h->d.t.isValid) ? 1 : 0; }
extern "C" int fsh_plain_la_use_at(const fsh_plain *h) { return FS_PLAIN_GET(h->f.t.useAT, h->d.t.useAT) ? 1 : 0; }
extern "C" void fsh_plain_la_at(const fsh_plain *h, void *out)
{
    if (h->kind == 0)
        memcpy(out, &h->f.at_packed, sizeof(h->f.at_packed));
    else
        memcpy(out, &h->d.at_packed, sizeof(h->d.at_packed));
}
#undef FS_PLAIN_GET
// {dx, dy, centerX, centerY} (FillGpuCoords / FillCoord, Fractal.cpp:1782-1786,1813-1817,1833-1844, :2828-2832):
// float[4] for kind 0, double[4] for kind 1.
extern "C" void fsh_plain_coords(const fsh_view *v, const fsh_plain *h, uint32_t w_aa, uint32_t h_aa, void *out)
{
    mpf_set_default_prec(v->prec_bits);
    Mp dx = (v->maxX - v->minX) / Mp::from_ui(w_aa);
    Mp dy = (v->maxY - v->minY) / Mp::from_ui(h_aa);
    const Mp &ocx = h->kind == 0 ? h->f.ob.cx : h->d.ob.cx;
    const Mp &ocy = h->kind == 0 ? h->f.ob.cy : h->d.ob.cy;
    Mp cX = ocx - v->minX;
    Mp cY = ocy - v->maxY;
    const double d[4] = {mpf_get_d(dx.v), mpf_get_d(dy.v), mpf_get_d(cX.v), mpf_get_d(cY.v)};
    for (int i = 0; i < 4; i++) {
        if (h->kind == 0)
            ((float *)out)[i] = (float)d[i];
        else
            ((double *)out)[i] = d[i];
    }
}
// field-wise double -> CudaDblflt conversions (CudaDblflt(double) = MattDblflt(double), dblflt.h:13-23)
namespace {
fs_real_p2x32 real_p2x32(double v)
{
    fs_real_p2x32 r;
    df_from_double(v, r.head, r.tail);
    return r;
}
fs_cplx_p2x32 cplx_p2x32(fs_cplx_f64 c)
{
    fs_cplx_p2x32 r;
    df_from_double(c.re, r.re_head, r.re_tail);
    df_from_double(c.im, r.im_head, r.im_tail);
    return r;
}
} // namespace
extern "C" void fsh_convert_orbit_f64_to_p2x32(const fs_orbit_f64 *in, uint64_t n, fs_orbit_p2x32 *out)
{
    for (uint64_t i = 0; i < n; i++) {
        df_from_double(in[i].x, out[i].x_head, out[i].x_tail);
        df_from_double(in[i].y, out[i].y_head, out[i].y_tail);
    }
}
// CopyFullOrbitVector's SimpleCompression arm (PerturbationResults.cpp:265-268): x, y converted, index kept
extern "C" void fsh_convert_orbit_rc_f64_to_p2x32(const fs_orbit_f64_rc *in, uint64_t n, fs_orbit_p2x32_rc *out)
{
    for (uint64_t i = 0; i < n; i++) {
        out[i].index_and_rebase = in[i].index_and_rebase & 0x7FFFFFFFFFFFFFFFull;
        df_from_double(in[i].x, out[i].x_head, out[i].x_tail);
        df_from_double(in[i].y, out[i].y_head, out[i].y_tail);
    }
}
extern "C" void fsh_convert_la_f64_to_p2x32(const fs_la_f64_u32 *in, uint64_t n, fs_la_p2x32_u32 *out)
{
    for (uint64_t i = 0; i < n; i++) {
        fs_la_p2x32_u32 o;
        o.Ref = cplx_p2x32(in[i].Ref);
        o.ZCoeff = cplx_p2x32(in[i].ZCoeff);
        o.CCoeff = cplx_p2x32(in[i].CCoeff);
        o.LAThreshold = real_p2x32(in[i].LAThreshold);
        o.LAThresholdC = real_p2x32(in[i].LAThresholdC);
        o.MinMag = real_p2x32(in[i].MinMag);
        o.StepLength = in[i].StepLength;
        o.NextStageLAIndex = in[i].NextStageLAIndex;
        out[i] = o;
    }
}
extern "C" void fsh_convert_at_f64_to_p2x32(const fs_at_f64_u32 *in, fs_at_p2x32_u32 *out)
{
    out->StepLength = in->StepLength;
    out->ThresholdC = real_p2x32(in->ThresholdC);
    out->SqrEscapeRadius = real_p2x32(in->SqrEscapeRadius);
    out->RefC = cplx_p2x32(in->RefC);
    out->ZCoeff = cplx_p2x32(in->ZCoeff);
    out->CCoeff = cplx_p2x32(in->CCoeff);
    out->InvZCoeff = cplx_p2x32(in->InvZCoeff);
    out->CCoeffSqrInvZCoeff = cplx_p2x32(in->CCoeffSqrInvZCoeff);
    out->CCoeffInvZCoeff = cplx_p2x32(in->CCoeffInvZCoeff);
    out->CCoeffNormSqr = real_p2x32(in->CCoeffNormSqr);
    out->RefCNormSqr = real_p2x32(in->RefCNormSqr);
    out->factor = real_p2x32(in->factor);
}
extern "C" void fsh_convert_coords_f64_to_p2x32(const double in[4], fs_real_p2x32 out[4])
{
    for (int i = 0; i < 4; i++)
        out[i] = real_p2x32(in[i]);
}
