/* fs_inputs.h -- C ABI of libfsinputs.so: host-side builders for the inputs of the per-pixel pass.
 *
 * These stand in for the parts of FractalShark that sit *upstream* of GPURenderer when FractalShark itself
 * is not built (bench.py, tests, the oracle harness): view geometry, the GMP reference orbit and the LAv2
 * table.  Inside FractalShark none of this is needed -- Fractal.cpp hands the renderer its own
 * PerturbationResults / LAReference buffers, which have the layouts in fs_layout.h.
 *
 * Reference code each entry point follows is cited in fractalshark_amd/host/refinputs.cpp.
 */
#ifndef FS_INPUTS_H
#define FS_INPUTS_H

#include <stddef.h>
#include <stdint.h>

#include "fs_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fsh_view fsh_view;
typedef struct fsh_orbit fsh_orbit;
typedef struct fsh_la fsh_la;
typedef struct fsh_bla fsh_bla;

/* Fractal::View(n) for a window of width x height: parse the bounding box at 1e6 bits, pick the working
 * precision, square the box to the window aspect ratio. */
fsh_view *fsh_view_create(const char *minX, const char *minY, const char *maxX, const char *maxY,
                          uint32_t width, uint32_t height);
void fsh_view_destroy(fsh_view *v);
uint64_t fsh_view_precision_bits(const fsh_view *v);
/* Imagina ".im" location files, the form the reference writes for a view without a stored orbit
 * (RefOrbitCalc::SaveOrbitResults(filename), RefOrbitCalc.cpp:3117-3166) and reads back in LoadOrbitConstInternal
 * (:3425-3520): IMFileHeader, halfH (HDRFloat<double, Left, int64_t>), iteration limit, the centre as two MPIR raw-stream
 * mpf values (MpirSerialization.cpp:157-187).  fsh_view_load_im also accepts files that carry a reference orbit
 * (*has_orbit = 1) and loads their location; the stored orbit (the reference's MaxCompression intermediate form) is
 * not read.  exp_bytes is sizeof(long) of the build whose files are meant: 4 = the reference's Windows build and Imagina,
 * 8 = the reference built on Linux (the mpf exponent field is a raw `long`); the loader finds it from the section's
 * length and reports it.  Returns 0 / a view, or -1 / NULL. */
int fsh_view_save_im(const fsh_view *v, uint64_t iteration_limit, const char *path, int exp_bytes);
/* The integer stream the two mpf values of an ".im" file are written in (MpirSerialization::mpz_out_raw_stream /
 * mpz_inp_raw_stream = MPIR's mpz_out_raw: big-endian signed byte count, magnitude most significant byte first), memory to
 * memory: what tests/test_mpir_wire_format.py pins to the byte vectors of the reference's own unit tests. */
/* The LAParameters the LAv2 tables are built with (the reference's defaults, LAParameters.h:66-75; pinned by the values of its
 * unit tests, tests/test_host_builder_vectors.py): detection method, then the six exponents in the order of the getters. */
void fsh_la_default_params(int32_t out[7]);
size_t fsh_mpz_raw_write(const char *value, int base, unsigned char *out, size_t cap);
size_t fsh_mpz_raw_read(const unsigned char *in, size_t n, char *out_decimal, size_t cap);
fsh_view *fsh_view_load_im(const char *path, uint32_t width, uint32_t height, uint64_t *iteration_limit, int *has_orbit,
                           int *exp_bytes_out);
/* ".im" files WITH a reference orbit (RefOrbitCalc::SaveOrbitResults(results, filename), RefOrbitCalc.cpp:3039-3115; the
 * reader: LoadOrbitConst :3318-3423 -> LoadOrbitBin + DecompressMax, PerturbationResults.cpp:2104-2211,1660-1850): the
 * location section as above, then the orbit under "max compression" (CompressMax, PerturbationResults.cpp:1347-1640;
 * compression_exp: the reference's default is 20).  HDRFloat<float> orbits carry the "Sharks:)" magic, HDRFloat<double>
 * ones Imagina's; ExtendedRange = true.  fsh_orbit_load_im rebuilds the full orbit from the file (no GMP iteration);
 * files without the ExtendedRange flag (plain float / double orbits) are refused (NULL). */
int fsh_orbit_save_im(const fsh_orbit *o, uint64_t num_iterations, int compression_exp, const char *path, int exp_bytes);
fsh_orbit *fsh_orbit_load_im(const char *path, uint64_t *iteration_limit);
/* which: 0=minX 1=minY 2=maxX 3=maxY; printf("%.Fe") of the squared bounding box. */
int fsh_view_bbox_str(const fsh_view *v, int which, char *buf, size_t buflen);

/* out = {dx, dy, minX, maxY} as un-reduced HDRFloat (CpuHDR32 / CpuHDR64 direct kernels). */
void fsh_view_coords_direct_hdr32(const fsh_view *v, uint32_t w_aa, uint32_t h_aa, fs_real_hdr32 out[4]);
void fsh_view_coords_direct_hdr64(const fsh_view *v, uint32_t w_aa, uint32_t h_aa, fs_real_hdr64 out[4]);

/* Gpu1x32 / Gpu2x32 / Gpu2x64 / Gpu4x32 / Gpu4x64 direct kernels: {cx = minX, cy = minY, dx, dy} (Fractal::FillGpuCoords).
 * kind 0: float[4]; kind 1: float[8] = (head, tail) pairs (MattDblflt); kind 2: double[8] (MattDbldbl); kind 3: float[16]
 * = (x, y, z, w) quadruples, most significant first (MattQFltflt); kind 4: double[16] (MattQDbldbl). */
void fsh_view_coords_direct_lp(const fsh_view *v, uint32_t w_aa, uint32_t h_aa, int kind, void *out);

/* out = {dx, dy, minX, maxY} as doubles (Cpu64 / direct kernels). */
void fsh_view_coords_direct_f64(const fsh_view *v, uint32_t w_aa, uint32_t h_aa, double out[4]);

/* Reference orbit at the bounding-box centre. is64: 0 = HDRFloat<float>, 1 = HDRFloat<double>.
 * periodicity: 1 = STPeriodicity (what PerturbationAlg::Auto picks below 1e150), 0 = ST. */
fsh_orbit *fsh_orbit_create(const fsh_view *v, int is64, uint64_t max_iter, int periodicity);
/* compression_exp < 0: uncompressed; >= 0: PerturbExtras::SimpleCompression with CompressionError 10^exp (default 20).
 * The uncompressed accessors (fsh_orbit_data_*, count) then expose the orbit as RuntimeDecompressor reproduces it. */
fsh_orbit *fsh_orbit_create_ex(const fsh_view *v, int is64, uint64_t max_iter, int periodicity, int compression_exp);
int fsh_orbit_is64(const fsh_orbit *o); /* 1: HDRFloat<double> entries, 0: HDRFloat<float> */
int fsh_orbit_is_compressed(const fsh_orbit *o);
uint64_t fsh_orbit_compressed_count(const fsh_orbit *o);
const fs_orbit_hdr32_rc *fsh_orbit_compressed_data_hdr32(fsh_orbit *o);
void fsh_orbit_low_hdr32(const fsh_orbit *o, fs_real_hdr32 out[2]); /* {OrbitXLow, OrbitYLow} */
const fs_orbit_hdr64_rc *fsh_orbit_compressed_data_hdr64(fsh_orbit *o);
void fsh_orbit_low_hdr64(const fsh_orbit *o, fs_real_hdr64 out[2]);
void fsh_orbit_destroy(fsh_orbit *o);
uint64_t fsh_orbit_count(const fsh_orbit *o);  /* GetCountOrbitEntries(), includes the zero entry */
/* Test hook: entries idx[k] of an uncompressed orbit scaled by 2^exp2[k] (period boundaries where a test of the LA builders
 * wants them; the result is not the orbit of any view).  Returns the number of entries changed. */
uint64_t fsh_orbit_scale_entries(fsh_orbit *o, const uint64_t *idx, const int32_t *exp2, uint64_t n);
uint64_t fsh_orbit_period(const fsh_orbit *o); /* GetPeriodMaybeZero() */
const fs_orbit_hdr32 *fsh_orbit_data_hdr32(fsh_orbit *o);
const fs_orbit_hdr64 *fsh_orbit_data_hdr64(fsh_orbit *o);
void fsh_orbit_max_radius_hdr32(const fsh_orbit *o, fs_real_hdr32 *out);
void fsh_orbit_max_radius_hdr64(const fsh_orbit *o, fs_real_hdr64 *out);
/* PerturbExtras::Bad form of an hdr32 orbit + its binary32 copy (inputs of RenderPerturbBLAScaled). */
const fs_orbit_hdr32_bad *fsh_orbit_data_hdr32_bad(fsh_orbit *o);
const fs_orbit_f32_bad *fsh_orbit_data_f32_bad(fsh_orbit *o);
uint64_t fsh_orbit_bad_count(const fsh_orbit *o);

/* out = {dx, dy, centerX, centerY}, each reduced (perturbation paths). */
void fsh_view_coords_perturb_hdr32(const fsh_view *v, const fsh_orbit *o, uint32_t w_aa, uint32_t h_aa,
                                   fs_real_hdr32 out[4]);
void fsh_view_coords_perturb_hdr64(const fsh_view *v, const fsh_orbit *o, uint32_t w_aa, uint32_t h_aa,
                                   fs_real_hdr64 out[4]);

/* 2x32 (HDRFloat<CudaDblflt<MattDblflt>>) inputs.  FractalShark derives them from the HDRFloat<double> orbit / LA
 * table by field-wise conversion (PerturbationResults::CopyPerturbationResults, LAReference::CopyLAReference);
 * these restate that conversion.  Coordinates come straight from the high-precision view (orbit must be hdr64). */
void fsh_convert_orbit_hdr64_to_2x32(const fs_orbit_hdr64 *in, uint64_t n, fs_orbit_2x32 *out);
/* SimpleCompression form: waypoints converted, indices kept (CopyFullOrbitVector, PerturbationResults.cpp:265-268);
 * out[2] = {OrbitXLow, OrbitYLow} of the converted results (the double cast of the reference point). */
void fsh_convert_orbit_rc_hdr64_to_2x32(const fs_orbit_hdr64_rc *in, uint64_t n, fs_orbit_2x32_rc *out);
void fsh_orbit_low_2x32(const fsh_orbit *o, fs_real_2x32 out[2]);
void fsh_convert_la_hdr64_to_2x32(const fs_la_hdr64_u32 *in, uint64_t n, fs_la_2x32_u32 *out);
void fsh_convert_at_hdr64_to_2x32(const fs_at_hdr64_u32 *in, fs_at_2x32_u32 *out);
/* out = {dx, dy, centerX, centerY}; mantissas in [0.5,1), not reduced (FillCoord, Fractal.cpp:1826-1832). */
void fsh_view_coords_perturb_2x32(const fsh_view *v, const fsh_orbit *o, uint32_t w_aa, uint32_t h_aa,
                                  fs_real_2x32 out[4]);

/* Host instantiation of the product's 2x32 arithmetic (fractalshark_amd/csrc/df32_math.hpp), for CPU-side cross-checks.
 * op: 0 add, 1 sub, 2 mul on (head, tail) pairs. */
void fsh_df32_op(int op, const float a[2], const float b[2], float out[2]);
void fsh_hr2_reduce(fs_real_2x32 *v);
void fsh_hr2_add(const fs_real_2x32 *a, const fs_real_2x32 *b, int subtract, fs_real_2x32 *out);
void fsh_hc2_reduce(fs_cplx_2x32 *v);

/* LAv2 table (LAReference::GenerateApproximationData).  host_threads = std::thread::hardware_concurrency()
 * of the machine being mirrored: the reference's multi-threaded stage-0 scan splits the orbit into
 * min(count/50000, host_threads) chunks and the chunking can move record boundaries. */
fsh_la *fsh_la_create(const fsh_orbit *o, int host_threads);       /* type follows the orbit (hdr32 / hdr64) */
/* use_small_exponents = the reference's UsingDblflt flag (RefOrbitCalc.cpp:2346): set it when the hdr64 table is
 * built to be converted to 2x32; it caps the AT escape radius at 2^32 instead of 2^256 (LAInfoDeep.h:484-496). */
fsh_la *fsh_la_create_ex(const fsh_orbit *o, int host_threads, int use_small_exponents);
fsh_la *fsh_la_create_hdr32(const fsh_orbit *o, int host_threads); /* NULL for an hdr64 orbit */
void fsh_la_destroy(fsh_la *l);
int fsh_la_is64(const fsh_la *l);
uint32_t fsh_la_count(const fsh_la *l);
const void *fsh_la_data(const fsh_la *l); /* fs_la_hdr32_u32[] or fs_la_hdr64_u32[] */
uint32_t fsh_la_stage_count(const fsh_la *l);
const fs_la_stage_u32 *fsh_la_stages(const fsh_la *l);
int fsh_la_is_valid(const fsh_la *l);
int fsh_la_use_at(const fsh_la *l);
void fsh_la_at(const fsh_la *l, void *out); /* fs_at_hdr32_u32 or fs_at_hdr64_u32 */

/* BLA table (BLAS<uint32_t,HDRFloat<float>>::Init with blaSize = orbit max radius).  Levels 0 and 1 are never
 * materialised (m_FirstLevel = 2): their pointers are NULL and sizes 0. */
fsh_bla *fsh_bla_create(const fsh_orbit *o);       /* type follows the orbit */
fsh_bla *fsh_bla_create_hdr32(const fsh_orbit *o); /* NULL for an hdr64 orbit */
void fsh_bla_destroy(fsh_bla *b);
int32_t fsh_bla_num_levels(const fsh_bla *b); /* m_B.size() */
int32_t fsh_bla_lm2(const fsh_bla *b);        /* m_LM2 */
const void *const *fsh_bla_level_ptrs(const fsh_bla *b); /* fs_bla_hdr32[] / fs_bla_hdr64[] per level */
const uint64_t *fsh_bla_level_sizes(const fsh_bla *b);

/* Plain double orbit + BLA table (PerturbationResults<uint32_t,double,Disable>, BLAS<uint32_t,double>):
 * Cpu64PerturbedBLA / Gpu1x64PerturbedBLA. */
typedef struct fsh_orbit_f64 fsh_orbit_f64;
fsh_orbit_f64 *fsh_orbit_f64_create(const fsh_view *v, uint64_t max_iter, int periodicity);
void fsh_orbit_f64_destroy(fsh_orbit_f64 *o);
uint64_t fsh_orbit_f64_count(const fsh_orbit_f64 *o);
uint64_t fsh_orbit_f64_period(const fsh_orbit_f64 *o);
const fs_orbit_f64 *fsh_orbit_f64_data(const fsh_orbit_f64 *o);
const fs_orbit_f64_bad *fsh_orbit_f64_data_bad(fsh_orbit_f64 *o);         /* PerturbExtras::Bad form ... */
const fs_orbit_f32_bad *fsh_orbit_f64_data_f32_bad(fsh_orbit_f64 *o);     /* ... and its binary32 copy */
int32_t fsh_orbit_f64_bla_num_levels(const fsh_orbit_f64 *o);
int32_t fsh_orbit_f64_bla_lm2(const fsh_orbit_f64 *o);
const void *const *fsh_orbit_f64_bla_level_ptrs(const fsh_orbit_f64 *o); /* fs_bla_f64[] per level */
const uint64_t *fsh_orbit_f64_bla_level_sizes(const fsh_orbit_f64 *o);
void fsh_view_coords_perturb_f64(const fsh_view *v, const fsh_orbit_f64 *o, uint32_t w_aa, uint32_t h_aa, double out[4]);

/* Plain float / double orbit + LAv2 table: PerturbationResults<uint32_t,T,Disable> + LAReference<uint32_t,T,T,Disable>
 * for T = float (kind 0: Gpu1x32PerturbedLAv2*) or double (kind 1: Gpu1x64PerturbedLAv2*, and the source of the
 * Gpu2x32PerturbedLAv2* inputs, which FractalShark converts field-wise from the double ones). */
typedef struct fsh_plain fsh_plain;
fsh_plain *fsh_plain_create(const fsh_view *v, int kind, uint64_t max_iter, int periodicity, int host_threads);
/* compression_exp >= 0: PerturbExtras::SimpleCompression (see fsh_orbit_create_ex); the LA table is then built from
 * the orbit as RuntimeDecompressor reproduces it, with the SimpleCompression period divisor. */
fsh_plain *fsh_plain_create_ex(const fsh_view *v, int kind, uint64_t max_iter, int periodicity, int host_threads,
                               int compression_exp);
/* ".im" files with a reference orbit for the non-ExtendedRange types (RefOrbitCalc::SaveOrbitResults(results, filename) /
 * LoadOrbitConst for T = float | double, RefOrbitCalc.cpp:3039-3115, :3386-3412; SaveOrbitBin / LoadOrbitBin,
 * PerturbationResults.cpp:2047-2075, :2177-2183): ReferenceHeader::ExtendedRange = false, waypoints of two doubles and the
 * index field, "Sharks:)" magic for float and Imagina's for double.  fsh_plain_load_im rebuilds the orbit from the waypoints
 * (DecompressMax) and builds the LAv2 table from it; NULL for anything else (an ExtendedRange file: fsh_orbit_load_im). */
int fsh_plain_save_im(const fsh_plain *h, uint64_t num_iterations, int compression_exp, const char *path, int exp_bytes);
fsh_plain *fsh_plain_load_im(const char *path, uint64_t *iteration_limit, int host_threads);
int fsh_plain_is_compressed(const fsh_plain *h);
uint64_t fsh_plain_compressed_count(const fsh_plain *h);
const void *fsh_plain_compressed_data(const fsh_plain *h); /* fs_orbit_f32_rc[] / fs_orbit_f64_rc[] */
void fsh_plain_orbit_low(const fsh_plain *h, void *out);   /* {OrbitXLow, OrbitYLow}: float[2] / double[2] */
void fsh_plain_destroy(fsh_plain *h);
int fsh_plain_kind(const fsh_plain *h);
uint64_t fsh_plain_orbit_count(const fsh_plain *h);
uint64_t fsh_plain_orbit_period(const fsh_plain *h);
const void *fsh_plain_orbit_data(const fsh_plain *h); /* fs_orbit_f32[] / fs_orbit_f64[] */
uint32_t fsh_plain_la_count(const fsh_plain *h);
const void *fsh_plain_la_data(const fsh_plain *h);    /* fs_la_f32_u32[] / fs_la_f64_u32[] */
uint32_t fsh_plain_la_stage_count(const fsh_plain *h);
const fs_la_stage_u32 *fsh_plain_la_stages(const fsh_plain *h);
int fsh_plain_la_is_valid(const fsh_plain *h);
int fsh_plain_la_use_at(const fsh_plain *h);
void fsh_plain_la_at(const fsh_plain *h, void *out);  /* fs_at_f32_u32 / fs_at_f64_u32 */
/* out = {dx, dy, centerX, centerY}: float[4] (kind 0) or double[4] (kind 1) */
void fsh_plain_coords(const fsh_view *v, const fsh_plain *h, uint32_t w_aa, uint32_t h_aa, void *out);
void fsh_convert_orbit_f64_to_p2x32(const fs_orbit_f64 *in, uint64_t n, fs_orbit_p2x32 *out);
void fsh_convert_orbit_rc_f64_to_p2x32(const fs_orbit_f64_rc *in, uint64_t n, fs_orbit_p2x32_rc *out);
void fsh_convert_la_f64_to_p2x32(const fs_la_f64_u32 *in, uint64_t n, fs_la_p2x32_u32 *out);
void fsh_convert_at_f64_to_p2x32(const fs_at_f64_u32 *in, fs_at_p2x32_u32 *out);
void fsh_convert_coords_f64_to_p2x32(const double in[4], fs_real_p2x32 out[4]);

#ifdef __cplusplus
}
#endif

#endif /* FS_INPUTS_H */
