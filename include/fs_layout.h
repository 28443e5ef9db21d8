/* fs_layout.h -- plain-C record layouts that cross the fsmi355 C ABI.
 *
 * Every struct here is byte-for-byte the host layout FractalShark already keeps in memory for the
 * per-pixel pass, so a maintainer can hand the existing buffers over without conversion:
 *
 *   fs_orbit_hdr32      = GPUReferenceIter<HDRFloat<float>, PerturbExtras::Disable>
 *                         (HpSharkFloatLib/GPU_ReferenceIter.h:52-127; x is {mantissa,exp}, y is
 *                          HDROrder::Right = {exp,mantissa}; 16 B)
 *   fs_orbit_hdr64      = GPUReferenceIter<HDRFloat<double>, Disable>              (32 B)
 *   fs_la_hdr32_u32     = LAInfoDeep<uint32_t, HDRFloat<float>, float, Disable>
 *                         (HpSharkFloatLib/LAInfoDeep.h:36-43; 68 B, static-asserted against the GPU twin
 *                          at FractalSharkLib/GPU_LAReference.h:118-133)
 *   fs_la_stage_u32     = LAStageInfo<uint32_t>  (HpSharkFloatLib/LAInfoI.h:5-16; 8 B)
 *   fs_at_hdr32_u32     = ATInfo<uint32_t, HDRFloat<float>, float> (HpSharkFloatLib/ATInfo.h:84-99; 116 B)
 *   fs_bla_hdr32        = BLA<HDRFloat<float>>   (FractalSharkLib/BLA.h:9-16; 44 B)
 *   fs_orbit_2x32 / fs_la_2x32_u32 / fs_at_2x32_u32 = the HDRFloat<CudaDblflt<MattDblflt>> twins (24 / 104 / 184 B)
 *   fs_color16, fs_reduction = Color16 / ReductionResults (FractalSharkLib/GPU_Types.h:14-16,40-50)
 */
#ifndef FS_LAYOUT_H
#define FS_LAYOUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_real_hdr32 {
    float m;
    int32_t e;
} fs_real_hdr32;

typedef struct fs_cplx_hdr32 {
    float re;
    float im;
    int32_t e;
} fs_cplx_hdr32;

typedef struct fs_real_hdr64 {
    double m;
    int32_t e;
    int32_t pad_;
} fs_real_hdr64;

typedef struct fs_cplx_hdr64 {
    double re;
    double im;
    int32_t e;
    int32_t pad_;
} fs_cplx_hdr64;

typedef struct fs_orbit_hdr32 {
    float mx;
    int32_t ex;
    int32_t ey;
    float my;
} fs_orbit_hdr32;

/* GPUReferenceIter<HDRFloat<float>, PerturbExtras::SimpleCompression>: the CompressionIndexField base
 * (63-bit uncompressed index + 1 rebase bit, GPU_ReferenceIter.h:26-49) comes first, then x, y; 24 B. */
typedef struct fs_orbit_hdr32_rc {
    uint64_t index_and_rebase; /* bits 0..62 CompressionIndex, bit 63 Rebase */
    float mx;
    int32_t ex;
    int32_t ey;
    float my;
} fs_orbit_hdr32_rc;

/* GPUReferenceIter<HDRFloat<double>, PerturbExtras::SimpleCompression>: 40 B. */
typedef struct fs_orbit_hdr64_rc {
    uint64_t index_and_rebase;
    double mx;
    int32_t ex;
    int32_t pad0_;
    int32_t ey;
    int32_t pad1_;
    double my;
} fs_orbit_hdr64_rc;

typedef struct fs_orbit_hdr64 {
    double mx;
    int32_t ex;
    int32_t pad0_;
    int32_t ey;
    int32_t pad1_;
    double my;
} fs_orbit_hdr64;

/* PerturbExtras::Bad orbit entries (scaled kernels): the BadField base {uint32 bad; uint32 padding}
 * (GPU_ReferenceIter.h:10-24) comes first.  GPUReferenceIter<HDRFloat<float>,Bad> = 24 B, <float,Bad> = 16 B. */
typedef struct fs_orbit_hdr32_bad {
    uint32_t bad;
    uint32_t padding;
    float mx;
    int32_t ex;
    int32_t ey;
    float my;
} fs_orbit_hdr32_bad;

/* GPUReferenceIter<double, Bad>: 24 B */
typedef struct fs_orbit_f64_bad {
    uint32_t bad;
    uint32_t padding;
    double x;
    double y;
} fs_orbit_f64_bad;

typedef struct fs_orbit_f32_bad {
    uint32_t bad;
    uint32_t padding;
    float x;
    float y;
} fs_orbit_f32_bad;

typedef struct fs_orbit_f64 {
    double x;
    double y;
} fs_orbit_f64;

typedef struct fs_la_hdr32_u32 {
    fs_cplx_hdr32 Ref;
    fs_cplx_hdr32 ZCoeff;
    fs_cplx_hdr32 CCoeff;
    fs_real_hdr32 LAThreshold;
    fs_real_hdr32 LAThresholdC;
    fs_real_hdr32 MinMag;
    uint32_t StepLength;
    uint32_t NextStageLAIndex;
} fs_la_hdr32_u32;

typedef struct fs_la_stage_u32 {
    uint32_t LAIndex;
    uint32_t MacroItCount;
} fs_la_stage_u32;

typedef struct fs_at_hdr32_u32 {
    uint32_t StepLength;
    fs_real_hdr32 ThresholdC;
    fs_real_hdr32 SqrEscapeRadius;
    fs_cplx_hdr32 RefC;
    fs_cplx_hdr32 ZCoeff;
    fs_cplx_hdr32 CCoeff;
    fs_cplx_hdr32 InvZCoeff;
    fs_cplx_hdr32 CCoeffSqrInvZCoeff;
    fs_cplx_hdr32 CCoeffInvZCoeff;
    fs_real_hdr32 CCoeffNormSqr;
    fs_real_hdr32 RefCNormSqr;
    fs_real_hdr32 factor;
} fs_at_hdr32_u32;

typedef struct fs_bla_hdr32 {
    fs_real_hdr32 r2;
    fs_real_hdr32 Ax;
    fs_real_hdr32 Ay;
    fs_real_hdr32 Bx;
    fs_real_hdr32 By;
    int32_t l;
} fs_bla_hdr32;

/* HDRFloat<double> twins: LAInfoDeep<uint32_t,HDRFloat<double>,double,Disable> (128 B), ATInfo (232 B),
 * BLA<HDRFloat<double>> (88 B). */
typedef struct fs_la_hdr64_u32 {
    fs_cplx_hdr64 Ref;
    fs_cplx_hdr64 ZCoeff;
    fs_cplx_hdr64 CCoeff;
    fs_real_hdr64 LAThreshold;
    fs_real_hdr64 LAThresholdC;
    fs_real_hdr64 MinMag;
    uint32_t StepLength;
    uint32_t NextStageLAIndex;
} fs_la_hdr64_u32;

typedef struct fs_at_hdr64_u32 {
    uint32_t StepLength;
    uint32_t pad_;
    fs_real_hdr64 ThresholdC;
    fs_real_hdr64 SqrEscapeRadius;
    fs_cplx_hdr64 RefC;
    fs_cplx_hdr64 ZCoeff;
    fs_cplx_hdr64 CCoeff;
    fs_cplx_hdr64 InvZCoeff;
    fs_cplx_hdr64 CCoeffSqrInvZCoeff;
    fs_cplx_hdr64 CCoeffInvZCoeff;
    fs_real_hdr64 CCoeffNormSqr;
    fs_real_hdr64 RefCNormSqr;
    fs_real_hdr64 factor;
} fs_at_hdr64_u32;

typedef struct fs_bla_hdr64 {
    fs_real_hdr64 r2;
    fs_real_hdr64 Ax;
    fs_real_hdr64 Ay;
    fs_real_hdr64 Bx;
    fs_real_hdr64 By;
    int32_t l;
    int32_t pad_;
} fs_bla_hdr64;

/* BLA<double> (FractalSharkLib/BLA.h:9-16): 48 B. */
typedef struct fs_bla_f64 {
    double r2;
    double Ax;
    double Ay;
    double Bx;
    double By;
    int32_t l;
    int32_t pad_;
} fs_bla_f64;

/* ---- 2x32 ("float-float + exponent") records: HDRFloat<CudaDblflt<MattDblflt>> family.
 * MattDblflt is {head, tail} under #pragma pack(4) (HpSharkFloatLib/dblflt.h:5-62), CudaDblflt wraps it
 * (CudaDblflt.h:24-28), so HDRFloat<CudaDblflt> is 12 B {head, tail, exp}, HDROrder::Right is {exp, head, tail},
 * HDRFloatComplex<CudaDblflt> is 20 B and the orbit entry GPUReferenceIter<HDRFloat<CudaDblflt>,Disable> 24 B. */
typedef struct fs_real_2x32 {
    float head;
    float tail;
    int32_t e;
} fs_real_2x32;

typedef struct fs_cplx_2x32 {
    float re_head;
    float re_tail;
    float im_head;
    float im_tail;
    int32_t e;
} fs_cplx_2x32;

typedef struct fs_orbit_2x32 {
    float x_head;
    float x_tail;
    int32_t ex;
    int32_t ey;
    float y_head;
    float y_tail;
} fs_orbit_2x32;

/* LAInfoDeep<uint32_t, HDRFloat<CudaDblflt<MattDblflt>>, CudaDblflt<MattDblflt>, Disable>: 104 B. */
typedef struct fs_la_2x32_u32 {
    fs_cplx_2x32 Ref;
    fs_cplx_2x32 ZCoeff;
    fs_cplx_2x32 CCoeff;
    fs_real_2x32 LAThreshold;
    fs_real_2x32 LAThresholdC;
    fs_real_2x32 MinMag;
    uint32_t StepLength;
    uint32_t NextStageLAIndex;
} fs_la_2x32_u32;

/* ATInfo<uint32_t, HDRFloat<CudaDblflt<MattDblflt>>, CudaDblflt<MattDblflt>>: 184 B. */
typedef struct fs_at_2x32_u32 {
    uint32_t StepLength;
    fs_real_2x32 ThresholdC;
    fs_real_2x32 SqrEscapeRadius;
    fs_cplx_2x32 RefC;
    fs_cplx_2x32 ZCoeff;
    fs_cplx_2x32 CCoeff;
    fs_cplx_2x32 InvZCoeff;
    fs_cplx_2x32 CCoeffSqrInvZCoeff;
    fs_cplx_2x32 CCoeffInvZCoeff;
    fs_real_2x32 CCoeffNormSqr;
    fs_real_2x32 RefCNormSqr;
    fs_real_2x32 factor;
} fs_at_2x32_u32;

/* ---- IterType = uint64_t twins of the records that embed iteration counts (LAInfoI<uint64_t>, LAStageInfo<uint64_t>,
 * ATInfo<uint64_t,..>::StepLength; natural alignment, no packing pragma in the reference).  The device works with the
 * uint32_t records: fs_upload_la converts these on upload and refuses tables whose counts do not fit 32 bits. */
typedef struct fs_la_stage_u64 {
    uint64_t LAIndex;
    uint64_t MacroItCount;
} fs_la_stage_u64;

typedef struct fs_la_hdr32_u64 {
    fs_cplx_hdr32 Ref;
    fs_cplx_hdr32 ZCoeff;
    fs_cplx_hdr32 CCoeff;
    fs_real_hdr32 LAThreshold;
    fs_real_hdr32 LAThresholdC;
    fs_real_hdr32 MinMag;
    uint32_t pad_;
    uint64_t StepLength;
    uint64_t NextStageLAIndex;
} fs_la_hdr32_u64;

typedef struct fs_la_hdr64_u64 {
    fs_cplx_hdr64 Ref;
    fs_cplx_hdr64 ZCoeff;
    fs_cplx_hdr64 CCoeff;
    fs_real_hdr64 LAThreshold;
    fs_real_hdr64 LAThresholdC;
    fs_real_hdr64 MinMag;
    uint64_t StepLength;
    uint64_t NextStageLAIndex;
} fs_la_hdr64_u64;

typedef struct fs_la_2x32_u64 {
    fs_cplx_2x32 Ref;
    fs_cplx_2x32 ZCoeff;
    fs_cplx_2x32 CCoeff;
    fs_real_2x32 LAThreshold;
    fs_real_2x32 LAThresholdC;
    fs_real_2x32 MinMag;
    uint64_t StepLength;
    uint64_t NextStageLAIndex;
} fs_la_2x32_u64;

/* ATInfo<uint64_t, ..>: StepLength is 8 bytes, everything after it is laid out like the uint32_t record after its
 * StepLength (+ pad for the double type). */
typedef struct fs_at_hdr32_u64 {
    uint64_t StepLength;
    uint8_t rest[sizeof(fs_at_hdr32_u32) - 4]; /* ThresholdC .. factor */
} fs_at_hdr32_u64;
typedef struct fs_at_hdr64_u64 {
    uint64_t StepLength;
    uint8_t rest[sizeof(fs_at_hdr64_u32) - 8];
} fs_at_hdr64_u64;
typedef struct fs_at_2x32_u64 {
    uint64_t StepLength;
    uint8_t rest[sizeof(fs_at_2x32_u32) - 4];
    uint32_t pad_;
} fs_at_2x32_u64;

/* ---- plain (non-HDR) LAv2 families: T = float (Gpu1x32PerturbedLAv2*), double (Gpu1x64PerturbedLAv2*) and
 * CudaDblflt<MattDblflt> (Gpu2x32PerturbedLAv2*).  The complex type is FloatComplex<T> = {re, im}
 * (HpSharkFloatLib/FloatComplex.h:7-12), records are LAInfoDeep<uint32_t,T,T,Disable> (LAInfoDeep.h:35-41),
 * ATInfo<uint32_t,T,T> (ATInfo.h:84-95) and GPUReferenceIter<T,Disable> = {x, y} (GPU_ReferenceIter.h:56-127);
 * CudaDblflt is {head, tail} under #pragma pack(4). */
typedef struct fs_orbit_f32 {
    float x;
    float y;
} fs_orbit_f32;

typedef struct fs_orbit_p2x32 {
    float x_head;
    float x_tail;
    float y_head;
    float y_tail;
} fs_orbit_p2x32;

/* GPUReferenceIter<T, PerturbExtras::SimpleCompression> for the non-HDR types and for HDRFloat<CudaDblflt>: the
 * CompressionIndexField base first (GPU_ReferenceIter.h:26-49), then x, y. */
typedef struct fs_orbit_f32_rc {
    uint64_t index_and_rebase;
    float x;
    float y;
} fs_orbit_f32_rc;

typedef struct fs_orbit_f64_rc {
    uint64_t index_and_rebase;
    double x;
    double y;
} fs_orbit_f64_rc;

typedef struct fs_orbit_p2x32_rc {
    uint64_t index_and_rebase;
    float x_head;
    float x_tail;
    float y_head;
    float y_tail;
} fs_orbit_p2x32_rc;

typedef struct fs_orbit_2x32_rc {
    uint64_t index_and_rebase;
    float x_head;
    float x_tail;
    int32_t ex;
    int32_t ey;
    float y_head;
    float y_tail;
} fs_orbit_2x32_rc;

typedef struct fs_cplx_f32 {
    float re;
    float im;
} fs_cplx_f32;

typedef struct fs_cplx_f64 {
    double re;
    double im;
} fs_cplx_f64;

typedef struct fs_real_p2x32 {
    float head;
    float tail;
} fs_real_p2x32;

typedef struct fs_cplx_p2x32 {
    float re_head;
    float re_tail;
    float im_head;
    float im_tail;
} fs_cplx_p2x32;

#define FS_DECL_PLAIN_LA(NAME, CPLX, REAL, ITER)                                                                        \
    typedef struct NAME {                                                                                               \
        CPLX Ref;                                                                                                       \
        CPLX ZCoeff;                                                                                                    \
        CPLX CCoeff;                                                                                                    \
        REAL LAThreshold;                                                                                               \
        REAL LAThresholdC;                                                                                              \
        REAL MinMag;                                                                                                    \
        ITER StepLength;                                                                                                \
        ITER NextStageLAIndex;                                                                                          \
    } NAME
#define FS_DECL_PLAIN_AT(NAME, CPLX, REAL, ITER)                                                                        \
    typedef struct NAME {                                                                                               \
        ITER StepLength;                                                                                                \
        REAL ThresholdC;                                                                                                \
        REAL SqrEscapeRadius;                                                                                           \
        CPLX RefC;                                                                                                      \
        CPLX ZCoeff;                                                                                                    \
        CPLX CCoeff;                                                                                                    \
        CPLX InvZCoeff;                                                                                                 \
        CPLX CCoeffSqrInvZCoeff;                                                                                        \
        CPLX CCoeffInvZCoeff;                                                                                           \
        REAL CCoeffNormSqr;                                                                                             \
        REAL RefCNormSqr;                                                                                               \
        REAL factor;                                                                                                    \
    } NAME
FS_DECL_PLAIN_LA(fs_la_f32_u32, fs_cplx_f32, float, uint32_t);           /* 44 B */
FS_DECL_PLAIN_LA(fs_la_f64_u32, fs_cplx_f64, double, uint32_t);          /* 80 B */
FS_DECL_PLAIN_LA(fs_la_p2x32_u32, fs_cplx_p2x32, fs_real_p2x32, uint32_t); /* 80 B */
FS_DECL_PLAIN_LA(fs_la_f32_u64, fs_cplx_f32, float, uint64_t);           /* 56 B */
FS_DECL_PLAIN_LA(fs_la_f64_u64, fs_cplx_f64, double, uint64_t);          /* 88 B */
FS_DECL_PLAIN_LA(fs_la_p2x32_u64, fs_cplx_p2x32, fs_real_p2x32, uint64_t); /* 88 B */
FS_DECL_PLAIN_AT(fs_at_f32_u32, fs_cplx_f32, float, uint32_t);           /* 72 B */
FS_DECL_PLAIN_AT(fs_at_f64_u32, fs_cplx_f64, double, uint32_t);          /* 144 B */
FS_DECL_PLAIN_AT(fs_at_p2x32_u32, fs_cplx_p2x32, fs_real_p2x32, uint32_t); /* 140 B */
FS_DECL_PLAIN_AT(fs_at_f32_u64, fs_cplx_f32, float, uint64_t);           /* 80 B */
FS_DECL_PLAIN_AT(fs_at_f64_u64, fs_cplx_f64, double, uint64_t);          /* 144 B */
FS_DECL_PLAIN_AT(fs_at_p2x32_u64, fs_cplx_p2x32, fs_real_p2x32, uint64_t); /* 144 B */

typedef struct fs_color16 {
    uint16_t r, g, b, a;
} fs_color16;

typedef struct fs_reduction {
    uint64_t Min;
    uint64_t Max;
    uint64_t Sum;
} fs_reduction;

#ifdef __cplusplus
}
static_assert(sizeof(fs_orbit_hdr32) == 16, "orbit entry");
static_assert(sizeof(fs_orbit_hdr32_rc) == 24 && sizeof(fs_orbit_hdr64_rc) == 40, "compressed orbit entry");
static_assert(sizeof(fs_orbit_f32_rc) == 16 && sizeof(fs_orbit_f64_rc) == 24 && sizeof(fs_orbit_p2x32_rc) == 24 &&
                  sizeof(fs_orbit_2x32_rc) == 32,
              "compressed orbit entry");
static_assert(sizeof(fs_orbit_hdr64) == 32, "orbit entry (double)");
static_assert(sizeof(fs_orbit_hdr32_bad) == 24 && sizeof(fs_orbit_f32_bad) == 16 && sizeof(fs_orbit_f64_bad) == 24,
              "PerturbExtras::Bad orbit entries");
static_assert(sizeof(fs_la_hdr32_u32) == 68, "LA record");
static_assert(sizeof(fs_at_hdr32_u32) == 116, "AT record");
static_assert(sizeof(fs_bla_hdr32) == 44, "BLA record");
static_assert(sizeof(fs_real_hdr64) == 16 && sizeof(fs_cplx_hdr64) == 24, "double HDR");
static_assert(sizeof(fs_la_hdr64_u32) == 128, "LA record (double)");
static_assert(sizeof(fs_at_hdr64_u32) == 232, "AT record (double)");
static_assert(sizeof(fs_bla_hdr64) == 88, "BLA record (double)");
static_assert(sizeof(fs_real_2x32) == 12 && sizeof(fs_cplx_2x32) == 20 && sizeof(fs_orbit_2x32) == 24, "2x32 records");
static_assert(sizeof(fs_la_2x32_u32) == 104 && sizeof(fs_at_2x32_u32) == 184, "2x32 LA / AT records");
static_assert(sizeof(fs_la_stage_u64) == 16 && sizeof(fs_la_hdr32_u64) == 80 && sizeof(fs_la_hdr64_u64) == 136 &&
                  sizeof(fs_la_2x32_u64) == 112,
              "uint64_t IterType LA records");
static_assert(sizeof(fs_at_hdr32_u64) == 120 && sizeof(fs_at_hdr64_u64) == 232 && sizeof(fs_at_2x32_u64) == 192,
              "uint64_t IterType AT records");
static_assert(sizeof(fs_bla_f64) == 48 && sizeof(fs_orbit_f64) == 16, "plain double records");
static_assert(sizeof(fs_orbit_f32) == 8 && sizeof(fs_orbit_p2x32) == 16, "plain orbit entries");
static_assert(sizeof(fs_la_f32_u32) == 44 && sizeof(fs_la_f64_u32) == 80 && sizeof(fs_la_p2x32_u32) == 80 &&
                  sizeof(fs_la_f32_u64) == 56 && sizeof(fs_la_f64_u64) == 88 && sizeof(fs_la_p2x32_u64) == 88,
              "plain LA records");
static_assert(sizeof(fs_at_f32_u32) == 72 && sizeof(fs_at_f64_u32) == 144 && sizeof(fs_at_p2x32_u32) == 140 &&
                  sizeof(fs_at_f32_u64) == 80 && sizeof(fs_at_f64_u64) == 144 && sizeof(fs_at_p2x32_u64) == 144,
              "plain AT records");
#endif

#endif /* FS_LAYOUT_H */
