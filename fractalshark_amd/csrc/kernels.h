// kernels.h -- argument blocks and launch entry points shared by renderer.cpp (host) and the kernel translation units (kernels*.hip).
// Kernel arguments are plain PODs passed by value (well under the 4 KB kernarg limit), instead of the
// reference's by-value copies of non-trivial classes (GPUPerturbSingleResults, GPU_LAReference incl. ATInfo,
// GPU_BLAS -- SURVEY.md appendix B).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"

enum { FS_MODE_FULL = 0, FS_MODE_PO = 1, FS_MODE_LAO = 2 };
enum { FS_PARITY_LITERAL = 0, FS_PARITY_GPUSTAGE = 1 };
// kernel variant: 0 = tuned loop (default), 1 = literal operation-by-operation transcription (A/B reference)
enum { FS_VARIANT_TUNED = 0, FS_VARIANT_LITERAL = 1, FS_VARIANT_TUNED_NOSCALE = 2, FS_VARIANT_BASE_MASK = 0xff };
// ... ORed with the A/B flags of fs_set_kernel_variant (include/fsmi355.h): orbit entries of the scaled runs through LDS
// (k_lav2_hdr32_fast<kLds>), persistent lane-refilling launch of the BLA kernel (k_perturb_scalar<kRefill>)
enum { FS_VARIANT_FLAG_LDS_ORBIT = 0x100, FS_VARIANT_FLAG_REFILL = 0x200, FS_VARIANT_FLAG_WIDE = 0x400,
       FS_VARIANT_FLAG_NATURAL_ORDER = 0x800, FS_VARIANT_FLAG_BLA_POOL = 0x1000 };

// Frame geometry + the row-band layout of the local iteration buffer.
struct FsFrame {
    uint32_t width;         // full frame width  (incl. antialiasing)
    uint32_t height;        // full frame height (incl. antialiasing)
    uint32_t rounded_width; // row stride of the iteration buffer (multiple of 16, GPU_Render.cu:73-79)
    uint32_t local_rows;    // rows held locally (before padding to 8)
    uint32_t band_first;    // first global row of band 0
    uint32_t band_rows;     // rows per band
    uint32_t band_stride;   // global-row distance between consecutive owned bands
    uint32_t iter_u64;      // 1: the iteration buffer holds uint64_t elements (IterType = uint64_t), 0: uint32_t
    uint32_t wide;          // 1: launch the instantiation that COUNTS in 64 bits (iteration cap >= 2^32, or the
                            // FS_VARIANT_FLAG_WIDE test switch), 0: 32-bit counters
};

// Per-numeric-type device records.  F = float -> HDRFloat<float> (hdr32 ABI records), F = double -> HDRFloat<double>.
struct FsZ64 { // prepared orbit entry for double: {re, im, exp, -, 2^(8-2exp)}; 32 B
    double re;
    double im;
    int32_t e;
    int32_t pad_;
    double w;
};
template <class F> struct FsDev;
template <> struct FsDev<float> {
    using Z = float4; // {re, im, bitcast(exp), 2^(8-2exp)}
    using Orbit = fs_orbit_hdr32;
    using Real = fs_real_hdr32;
    using Cplx = fs_cplx_hdr32;
    using LA = fs_la_hdr32_u32;
    using AT = fs_at_hdr32_u32;
    using BLA = fs_bla_hdr32;
};
template <> struct FsDev<double> {
    using Z = FsZ64;
    using Orbit = fs_orbit_hdr64;
    using Real = fs_real_hdr64;
    using Cplx = fs_cplx_hdr64;
    using LA = fs_la_hdr64_u32;
    using AT = fs_at_hdr64_u32;
    using BLA = fs_bla_hdr64;
};

template <class F> struct FsCoordsT {
    fs::hreal<F> dx, dy, centerX, centerY;
};
using FsCoords32 = FsCoordsT<float>;

// PerformAT's result for one pixel (HDRFloat<double>): dz = z * InvZCoeff, reduced, and the AT iterations taken; i = all ones:
// the AT step does not apply to this pixel (ATInfo::isValid false).
struct FsAtRes {
    double re, im;
    int32_t e;
    uint32_t i;
};
template <class F> struct FsLav2ArgsT {
    uint32_t *out;
    const typename FsDev<F>::Z *zref; // prepared orbit
    const float4 *zq;                 // tuned HDRFloat<float> loop only: {re, im, ~exp | poison, -} (k_make_quiet_orbit)
    const float4 *zs;                 // ... and its scaled runs: {2Z.re, 2Z.im, 2^-5 max|Z| | -1, -} in true scale
    const float2 *zs2;                // the same 2Z alone, 8 B per entry, and
    const float4 *zqb;                // the block bounds of entries i + 3, + 7, + 11, + 15: what a 16-step body of the untested loop reads
    const typename FsDev<F>::LA *las;
    const fs_la_stage_u32 *stages;
    uint64_t *stats;
    FsFrame frame;
    FsCoordsT<F> coords;
    typename FsDev<F>::AT at;
    uint32_t orbit_count;
    uint32_t period;
    uint32_t stage_count;
    uint32_t n_iterations;    // low 32 bits of the iteration cap
    uint32_t n_iterations_hi; // high 32 bits: non-zero only for the IterType = uint64_t kernels (k_lav2_lit<.., uint64_t>)
    int la_valid;
    int use_at;
    int parity;
    // SimpleCompression orbit kept compressed in HBM and decompressed by the kernel as it walks it (fsk_lav2_seq): the
    // waypoints in the reference layout (fs_orbit_hdr32_rc / fs_orbit_hdr64_rc), their number, and the constant c of the
    // runtime decompressor (GPUPerturbResults::OrbitXLow / OrbitYLow); zref is NULL then
    const void *wp;
    uint32_t n_wp;
    typename FsDev<F>::Real cxLow, cyLow;
    // "longest tiles first" from the costs the previous frame recorded (k_lav2_hdr32_fast; fsk_tile_order_by_cost): wave w
    // of the launch renders the 8 x 8 tile tile_order[w] (row-major tile number of the LOCAL buffer, tiles_x per row,
    // 0xFFFFFFFF = none) instead of the one its block index names; null = natural order.  tile_cost (null = not recorded):
    // one word per tile, written by the wave that rendered it = the longest lane's perturbation steps (+ 8 per LA step).
    const uint32_t *tile_order;
    uint32_t *tile_cost;
    uint32_t tiles_x;
    // "pixels in the order of the previous frame's counts" (kernels_order.hip; k_lav2_lit only): lane s of the launch renders the
    // element pixel_order[s] of the local iteration buffer (row * rounded_width + column) instead of its tile's pixel; null =
    // the tile mapping
    const uint32_t *pixel_order;
    // (round 5) what the order is made from: when not null, every pixel records its COST -- the AT iterations it needs by itself
    // (20 bits) above its perturbation steps (12 bits, saturating) -- at its own position of this buffer (same geometry as the iteration buffer); the next frames of the
    // view are sorted by it.  (The count alone no longer tells: since the AT loop's cycle search a pixel inside the set costs
    // what its cycle took to show, not what the iteration limit asks.)
    uint32_t *pixel_cost;
    // (round 5, HDRFloat<double>) the AT iteration in a pass of its own (fsk_at_pass64): at_res = what PerformAT leaves per pixel
    // (buffer geometry of the iteration buffer) -- written by that pass, read by the frame's kernel INSTEAD of iterating;
    // at_cost (the pass only): every pixel's own AT iterations, recorded for the pass's order.  Why two passes: the AT loop reads
    // no memory, so its lanes can be grouped by cost alone, while the LA stages and the perturbation steps want neighbours
    // (table records, orbit entries) -- one launch cannot have both orders.
    struct FsAtRes *at_res;
    uint32_t *at_cost;
    // IterType = uint64_t POSITIONS (the waypoint-resident kernel with 64-bit counters, fsk_lav2_seq `wide`): high words of
    // the orbit's uncompressed length, of its period and of the AT step length; la_u64 = 1: `las` holds the reference's
    // uint64_t records (fs_la_hdr32_u64 / fs_la_hdr64_u64: 64-bit StepLength / NextStageLAIndex) instead of the narrowed ones
    uint32_t orbit_count_hi, period_hi, at_step_hi, la_u64;
};
using FsLav2Args32 = FsLav2ArgsT<float>;
template <class F> struct FsLaU64;
template <> struct FsLaU64<float> {
    using T = fs_la_hdr32_u64;
};
template <> struct FsLaU64<double> {
    using T = fs_la_hdr64_u64;
};

// Device-native BLA table of the HDRFloat<float> kernel (k_bla_make_native, kernels_tables.hip), built once per (table,
// orbit) pair next to the reference-layout levels.  All levels >= 2 back to back; position of element ix of level L =
// level_off[L] + ix.
//   FsBlaRec  (48 B, three 16-byte loads that never straddle a cache line): the four mantissas, the four exponents (the
//             reference record is 44 B, straddles alignment boundaries and interleaves mantissas with exponents), then the
//             step count AND the orbit value the jump arrives at: element ix of level L only ever applies at orbit index
//             (ix << L) + 1, so its arrival entry Z[(ix << L) + 1 + l] is a property of the record, and the dependent
//             orbit load behind every jump (its index is known only once l has arrived) disappears.  Stale once table or
//             orbit change.
//   ladder    (32 B per position): the r2 of this element and of the first elements of its left sub-trees on the three
//             levels below -- exactly the (level, index) pairs BLAS::LookupBackwards probes next when this one fails
//             (BLAS.cpp:256-310) -- as four 64-bit keys (exponent << 32 | mantissa bits): for reduced non-negative values
//             the reference's lexicographic (exponent, mantissa) compare is ONE signed 64-bit integer compare, and a whole
//             round of four probes is two 16-byte loads.  Levels below 2 carry INT64_MIN ("never valid").
//   (Measured and dropped, round 3: the ladder as four 32-bit order keys inside a 64-byte record -- one load per round,
//   equal keys decided exactly on the reference-layout record: 253 vs 229 ms on C5; the tie bookkeeping costs more vector
//   instructions than the second load, and this kernel is bound by vector issue, profiles/r03_c5_*.)
struct FsBlaRec {
    float Axm, Aym, Bxm, Bym;
    int32_t Axe, Aye, Bxe, Bye;
    float Zre, Zim; // prepared orbit entry at the arrival index (zeros when the jump would leave the orbit: never taken)
    int32_t Ze;
    uint32_t l;
};
static_assert(sizeof(FsBlaRec) == 48, "device-native BLA record");
constexpr int kBlaMaxLevels = 40;

template <class F> struct FsBlaArgsT {
    uint32_t *out;
    const typename FsDev<F>::Z *zref;
    const float4 *zq;                            // float, perturbation-only: quiet-run companion of zref
    const float4 *zs;                            // ... and the scaled runs' companion (see FsLav2ArgsT)
    const float2 *zs2;                           // ... in the compact form of the 16-step body (see FsLav2ArgsT)
    const float4 *zqb;
    const typename FsDev<F>::BLA *const *levels; // device array of device pointers, indexed by level
    uint64_t *stats;
    uint32_t *queue; // frame-wide pixel counter of the persistent (lane-refilling) launch, zeroed before each launch
    // "long tiles first" (fsk_tile_order, perturbation only): wave w of the launch renders the 8 x 8 tile tile_order[w]
    // (row-major tile number, 0xFFFFFFFF = none) instead of the one its block index names; null = natural order
    const uint32_t *tile_order;
    // probe launch: lane (x, l) runs the CENTRE pixel (8x + 4, 8l + 4) of tile (x, l) and stores its count at
    // probe_out[l * probe_pitch + x]; the frame fields describe the real frame
    uint32_t *probe_out;
    uint32_t probe_pitch;
    FsFrame frame;
    FsCoordsT<F> coords;
    uint32_t orbit_count;
    uint32_t n_iterations;    // low 32 bits of the iteration cap
    uint32_t n_iterations_hi; // high 32 bits: non-zero only for the 64-bit counting instantiation
    int32_t lm2;
    // HDRFloat<float> only: the device-native table (NULL = use `levels`, the reference-layout records)
    const FsBlaRec *nrec;
    const int4 *nlad; // two int4 per position
    const long long *nkmax; // per orbit index m = 4 q + 1: the largest r2 key LookupBackwards can meet there (pre-test), [q]
    uint32_t level_off[kBlaMaxLevels];
    // ... and its heap-numbered copy for the hand-written kernel (kernels_bla_fast.hip; NULL = not made): element ix of level L
    // at position 2^(H - L) + ix of hrec / hlad; hq[q] = {pre-test key, start position, start level} per orbit index 4 q + 1;
    // zb = the orbit with the quiet step's arrival bound in .w
    const float4 *zb;
    const int4 *hq;
    const int4 *hlad;
    const FsBlaRec *hrec;
};
using FsBlaArgs32 = FsBlaArgsT<float>;

// Plain double perturbation + BLA (Gpu1x64PerturbedBLA <-> Cpu64PerturbedBLA).
struct FsBlaArgsF64 {
    uint32_t *out;
    const fs_orbit_f64 *orbit;
    const fs_bla_f64 *const *levels;
    uint64_t *stats;
    FsFrame frame;
    double dx, dy, centerX, centerY;
    uint32_t orbit_count;
    uint32_t n_iterations;
    uint32_t n_iterations_hi;
    int32_t lm2;
};

// CalcCpuHDR<.., HDRFloat<F>, F> direct kernels: cx_row[x] = the CPU's accumulated cx (hreal<F>), dy/maxY un-reduced.
template <class F> struct FsDirectHdrArgsT {
    uint32_t *out;
    fs::hreal<F> *cx_row;
    uint64_t *stats;
    FsFrame frame;
    fs::hreal<F> dy;
    fs::hreal<F> maxY;
    uint32_t n_iterations;
    uint32_t n_iterations_hi; // high word of the cap: non-zero selects the 64-bit counting instantiation
};

// LAv2 for HDRFloat<CudaDblflt> (2x32): the reference records are used as they are (24-B orbit entries, 104-B LA
// records, the 184-B ATInfo in the argument block).
struct FsLav2Args2x32 {
    uint32_t *out;
    const fs_orbit_2x32 *orbit;
    const fs_la_2x32_u32 *las;
    const fs_la_stage_u32 *stages;
    uint64_t *stats;
    FsFrame frame;
    fs_real_2x32 coords[4]; // dx, dy, centerX, centerY
    fs_at_2x32_u32 at;
    uint32_t orbit_count;
    uint32_t stage_count;
    uint32_t n_iterations;
    uint32_t n_iterations_hi; // high word of the cap: non-zero selects the 64-bit counting instantiation
    int la_valid;
    int use_at;
    // SimpleCompression orbit kept compressed (fs_set_compressed_orbit_mode 1): the waypoints (fs_orbit_2x32_rc), their
    // number and the decompressor's constant c; `orbit` is NULL then and every entry comes from a per-pixel cursor
    const fs_orbit_2x32_rc *wp;
    uint32_t n_wp;
    fs_real_2x32 cxLow, cyLow;
    const uint32_t *pixel_order; // see FsLav2ArgsT
    uint32_t *pixel_cost;        // see FsLav2ArgsT
    const uint32_t *tile_order;  // see FsLav2ArgsT (round 6: the first frame of a view, kernels_tile_sample.hip)
    uint32_t tiles_x;
};

// Non-HDR LAv2 (Gpu1x32 / Gpu1x64 / Gpu2x32 PerturbedLAv2*): records in the reference layouts of the selected type
// (fs_layout.h "plain" families); `at` and `coords` hold the ATInfo record / the four coordinates of that type.
struct FsLav2ArgsPlain {
    uint32_t *out;
    const void *orbit;
    const void *las;
    const fs_la_stage_u32 *stages;
    uint64_t *stats;
    FsFrame frame;
    alignas(8) uint8_t at[sizeof(fs_at_f64_u32)];
    alignas(8) uint8_t coords[4 * sizeof(double)];
    uint32_t orbit_count;
    uint32_t stage_count;
    uint32_t n_iterations;
    uint32_t n_iterations_hi;
    int la_valid;
    int use_at;
    // SimpleCompression orbit kept compressed: waypoints in the type's reference layout (fs_orbit_f32_rc / _f64_rc / _p2x32_rc),
    // their number, the decompressor's constant c (float / double / fs_real_p2x32, 8 bytes each at most); `orbit` is NULL then
    const void *wp;
    uint32_t n_wp;
    alignas(8) uint8_t c_low[2][8];
};

// Scaled perturbation (GpuHDRx32PerturbedScaled): the HDRFloat<float> orbit with `bad` flags and its binary32 copy,
// both in the reference layouts.
struct FsScaledArgs32 {
    uint32_t *out;
    const fs_orbit_hdr32_bad *orbit_t;
    const fs_orbit_f32_bad *orbit_f;
    uint64_t *stats;
    FsFrame frame;
    FsCoordsT<float> coords;
    uint32_t orbit_count;
    uint32_t n_iterations;
    uint32_t n_iterations_hi; // high word of the cap: non-zero selects the 64-bit counting instantiation (literal kernel)
    float w2threshold; // exp(log(1e30f) / 2), ScaledKernels.cuh:21,66
};

struct FsScaledArgsF64 { // Gpu1x32PerturbedScaled: T = double
    uint32_t *out;
    const fs_orbit_f64_bad *orbit_t;
    const fs_orbit_f32_bad *orbit_f;
    uint64_t *stats;
    FsFrame frame;
    double dx, dy, centerX, centerY;
    uint32_t orbit_count;
    uint32_t n_iterations;
    uint32_t n_iterations_hi;
    float w2threshold;
};

// Gpu1x32 / Gpu2x32 / Gpu2x64 direct kernels: c32 = {cx, cy, dx, dy} (1x32) or {cx.head, cx.tail, cy.., dx.., dy..} (2x32);
// c64 = the same eight values as doubles (2x64).  cx / cy are the view's min corner.
struct FsDirectLpArgs {
    uint32_t *out;
    uint64_t *stats;
    FsFrame frame;
    float c32[16];  // 1x32: 4, 2x32: 8, 4x32: 16 values
    double c64[16]; // 2x64: 8, 4x64: 16 values
    uint32_t n_iterations;
    uint32_t n_iterations_hi;
};

struct FsDirectArgs64 {
    uint32_t *out;
    double *cx_row; // [width] row prefix of cx
    uint64_t *stats;
    FsFrame frame;
    double dy;
    double maxY;
    uint32_t n_iterations;
    uint32_t n_iterations_hi;
};

void fsk_prepare_orbit_hdr32(const fs_orbit_hdr32 *in, float4 *out, uint64_t n, hipStream_t s);
// type_tag = FS_T_F32 | FS_T_F64 | FS_T_2X32 | FS_T_HDR2X32; wp / out / cxLow are the records of that type
// (fs_orbit_*_rc -> fs_orbit_*; float / double / fs_real_p2x32 / fs_real_2x32), kernels_decompress.hip
void fsk_decompress_orbit_plain(int type_tag, const void *wp, uint64_t n_wp, uint64_t n_uncompressed, const void *cxLow,
                                const void *cyLow, void *out, hipStream_t s);
void fsk_decompress_orbit_hdr32(const fs_orbit_hdr32_rc *wp, uint64_t n_wp, uint64_t n_uncompressed, fs_real_hdr32 cxLow,
                                fs_real_hdr32 cyLow, float4 *out, hipStream_t s);
void fsk_decompress_orbit_hdr64(const fs_orbit_hdr64_rc *wp, uint64_t n_wp, uint64_t n_uncompressed, fs_real_hdr64 cxLow,
                                fs_real_hdr64 cyLow, FsZ64 *out, hipStream_t s);
void fsk_prepare_orbit_hdr64(const fs_orbit_hdr64 *in, FsZ64 *out, uint64_t n, hipStream_t s);
void fsk_make_quiet_orbit(const float4 *zref, float4 *zq, float2 *zs2, float4 *zqb, uint64_t n, hipStream_t s);
// Launch order for "long tiles first": order[0 .. n_slots) = the tiles whose probe count (their own centre's or a
// neighbour's) reached `threshold`, in tile order, then the others, then 0xFFFFFFFF; order[n_slots] = the number of long
// tiles.  probe: tiles_y rows of tiles_x counts.
void fsk_tile_order(const uint32_t *probe, uint32_t probe_pitch, uint32_t tiles_x, uint32_t tiles_y, uint32_t threshold,
                    uint32_t *order, uint32_t n_slots, hipStream_t s);
// Launch order "longest tiles first" from recorded costs: order[0 .. n_tiles) = the tile numbers sorted by cost class,
// highest first -- 256 classes, 8 per octave of the cost (exponent and three mantissa bits of the cost as a float) -- tile
// order kept inside a class (neighbours keep starting together: they walk the same stretch of the orbit);
// order[n_tiles .. n_slots) = 0xFFFFFFFF; order[n_slots] = 0.  tmp: fsk_tile_order_work_words() words of work memory.
uint32_t fsk_tile_order_work_words(uint32_t n_tiles);
void fsk_tile_order_by_cost(const uint32_t *cost, uint32_t n_tiles, uint32_t *tmp, uint32_t *order, uint32_t n_slots,
                            hipStream_t s);
// wave slots of the fsk_lav2_hdr32 launch for this frame (>= its number of 8 x 8 tiles), 0 = the launch shape is not the
// default one (FSMI355_BLOCK experiment): no tile order then
uint32_t fsk_lav2_hdr32_slots(const FsFrame &f);
void fsk_lav2_hdr32(const FsLav2Args32 &A, int mode, bool stats, int variant, hipStream_t s);
void fsk_lav2_hdr64(const FsLav2ArgsT<double> &A, int mode, bool stats, hipStream_t s);
void fsk_at_pass64(const FsLav2ArgsT<double> &A, hipStream_t s); // A.at_res (out), A.at_cost (out, optional), A.pixel_order (optional)
// IterType = uint64_t with 64-bit iteration counting (iteration caps of 2^32 and above): the literal kernel instantiated
// with a 64-bit counter; needs a uint64_t iteration buffer (frame.iter_u64)
void fsk_lav2_wide(const FsLav2Args32 *A32, const FsLav2ArgsT<double> *A64, int mode, bool stats, hipStream_t s);
// PerturbExtras::SimpleCompression with the orbit decompressed IN the kernel (GPUPerturbSingleResults::SeqWorkspace /
// GetIterSeq / BinarySearch, Perturb.cuh:146-326): only the waypoints are resident (A.wp); the literal kernel walks them
void fsk_lav2_seq(const FsLav2Args32 *A32, const FsLav2ArgsT<double> *A64, int mode, bool stats, hipStream_t s);
// test hook: one lane's SeqOrbit cursor (32- or 64-bit positions) seeks to `start` and walks n entries on; out[k] = the
// orbit value at start + k as hcplx<float> (12 B) / hcplx<double> (24 B)
void fsk_seq_cursor_probe(bool is64, bool wide_pos, const void *wp, uint32_t n_wp, const void *cx, const void *cy,
                          uint64_t start, uint32_t n, void *out, hipStream_t s);
void fsk_lav2_2x32(const FsLav2Args2x32 &A, int mode, bool stats, hipStream_t s);
// kind: 0 = float, 1 = double, 2 = CudaDblflt
void fsk_lav2_plain(const FsLav2ArgsPlain &A, int kind, int mode, bool stats, hipStream_t s);
// kind: 0 = Gpu1x32, 1 = Gpu2x32, 2 = Gpu2x64; false = iteration_precision the reference does not instantiate
bool fsk_direct_lp(const FsDirectLpArgs &A, int kind, int iteration_precision, bool stats, hipStream_t s);
void fsk_scaled_hdr32(const FsScaledArgs32 &A, bool stats, int variant, hipStream_t s);
void fsk_scaled_bounds(fs_orbit_f32_bad *of, uint64_t n, hipStream_t s); // tuned kernel: per-entry bound into .padding
void fsk_scaled_f64(const FsScaledArgsF64 &A, bool stats, int variant, hipStream_t s);
// BLA table build on the device: levels[l] = device memory for epl[l] records (NULL below the first materialised level 2)
void fsk_bla_build_hdr32(const float4 *zref, void *const *levels, const uint64_t *epl, int n_levels, fs_real_hdr32 bla_size,
                         hipStream_t s);
void fsk_bla_build_hdr64(const FsZ64 *zref, void *const *levels, const uint64_t *epl, int n_levels, fs_real_hdr64 bla_size,
                         hipStream_t s);
// levels: device pointer table of the reference-layout levels; level_off / epl: host arrays [n_levels]; bad: device word that
// is set when an r2 is not a reduced non-negative finite value (the caller then keeps using the reference-layout table)
void fsk_bla_make_native(const fs_bla_hdr32 *const *levels, const uint32_t *level_off, const uint64_t *epl, int n_levels,
                         const float4 *zref, uint32_t orbit_count, FsBlaRec *rec, int4 *lad, uint32_t *bad, int32_t lm2, long long *kmax, uint32_t n_kmax, hipStream_t s);
void fsk_perturb_scalar_hdr32(const FsBlaArgs32 &A, bool use_bla, bool stats, int variant, hipStream_t s);
// the hand-written HDRFloat<float> BLA kernel and its tables (kernels_bla_fast.hip).  fsk_bla_heap_positions: elements of hrec
// (and pairs of hlad) a table with these level sizes needs, 0 = the table cannot be numbered that way.
uint64_t fsk_bla_heap_positions(const uint64_t *epl, int n_levels);
void fsk_bla_make_heap(const FsBlaRec *rec, const int4 *lad, const long long *kmax, uint32_t n_kmax, const uint32_t *level_off,
                       const uint64_t *epl, int n_levels, int32_t lm2, const float4 *zref, uint32_t orbit_count, FsBlaRec *hrec,
                       int4 *hlad, int4 *hq, float4 *zb, hipStream_t s);
void fsk_bla_hdr32_fast(const FsBlaArgs32 &A, bool pool, hipStream_t s);
// pixel order from a frame's counts (kernels_order.hip): n = elements of the iteration buffer; work = 2 n words; order = n words
// An order for a view's FIRST frame (kernels_tile_sample.hip): PerformAT's loop for one pixel per 8 x 8 tile, in binary64.
struct FsTileSampleArgs {
    FsFrame frame;
    FsCoordsT<double> coords;
    fs::hreal<double> ThresholdC, SqrEscapeRadius;
    fs::hcplx<double> RefC, CCoeff;
    uint32_t StepLength;
    uint32_t n_iterations;
    uint32_t tiles_x, tiles_y; // tiles of the local buffer, row-major; n_slots >= tiles_x * tiles_y: waves of the frame's launch
    uint32_t n_slots;
    uint32_t *cost; // [n_slots] out: the sampled pixel's own AT iterations + 1, 0 = no AT step (or no tile)
};
void fsk_at_tile_sample64(const FsTileSampleArgs &A, hipStream_t s);
void fsk_tile_order_finish(uint32_t *order, uint32_t n_slots, uint32_t n_tiles, hipStream_t s);
size_t fsk_pixel_order_temp_bytes(uint32_t n);
hipError_t fsk_pixel_order_build(const uint32_t *counts, uint32_t n, uint32_t *work, uint32_t *order, void *temp, size_t temp_bytes,
                                 hipStream_t s, int key_bits = 32);
void fsk_lav2_lit32(const FsLav2Args32 &A, int mode, bool stats, dim3 g, dim3 b, hipStream_t s); // kernels.hip
void fsk_lav2_hdr64_fast(const FsLav2ArgsT<double> &A, int mode, bool stats, hipStream_t s); // kernels_hdr64.hip
void fsk_perturb_scalar_hdr64(const FsBlaArgsT<double> &A, bool use_bla, bool stats, int variant, hipStream_t s);
void fsk_perturb_bla_f64(const FsBlaArgsF64 &A, bool use_bla, bool stats, hipStream_t s);
void fsk_direct_hdr32(const FsDirectHdrArgsT<float> &A, fs::hreal<float> minX, fs::hreal<float> dx, bool stats, hipStream_t s);
void fsk_direct_hdr64(const FsDirectHdrArgsT<double> &A, fs::hreal<double> minX, fs::hreal<double> dx, bool stats,
                      hipStream_t s);
void fsk_direct_f64(const FsDirectArgs64 &A, double minX, double dx, bool stats, hipStream_t s);
// iter_u64: element type of the iteration buffer (see FsFrame)
void fsk_antialias(const void *iters, int iter_u64, uint32_t rounded_width, fs_color16 *colors, const fs_color16 *pal,
                   uint32_t pal_iters, uint32_t aux_depth, uint32_t aa, uint32_t color_w, uint32_t color_h,
                   uint64_t n_iterations, hipStream_t s);
void fsk_reduce(const void *iters, int iter_u64, uint32_t rounded_width, uint32_t width, uint32_t rows,
                fs_reduction *out, hipStream_t s);
// multi-GPU tiler: out row y = in row index[y] (row_bytes a multiple of 16)
void fsk_gather_rows(const void *in, void *out, const uint32_t *index, uint32_t row_bytes, uint32_t rows, hipStream_t s);

// ---- LAv2 table construction on the device (kernels_la.hip; host orchestration: fs_build_la in renderer.cpp).
// F = float | double.  Work arrays are raw device pointers (hreal<F> / LAInfo<F> of csrc/la_math.hpp).
template <class F> void fsk_la_src_orbit(const void *zref, uint32_t n, void *chebv, hipStream_t s);
template <class F> void fsk_la_src_stage(const void *P, uint32_t n, void *chebv, void *mm, uint32_t *steps, hipStream_t s);
void fsk_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, hipStream_t s); // exclusive; out[n] = total
template <class F> void fsk_la_first(bool stage0, const void *chebv, const void *mm, uint32_t limit, uint32_t *out, hipStream_t s);
// stage 0, multi-threaded variant: first uncapped detection from state (bases[q], 1) for every query q; out[q] = index or ~0u
template <class F>
void fsk_la_first_from(const void *chebv, const uint32_t *bases, uint32_t n_queries, uint32_t limit, uint32_t *out, hipStream_t s);
// stage-0 records from an explicit list of orbit segments seg[2k] .. seg[2k+1] (+ the stage's tail record when tail_out)
template <class F>
void fsk_la_records_list(const void *zref, const uint32_t *seg, uint32_t n, void *out, void *tail_out, uint32_t max_ref,
                         hipStream_t s);
template <class F>
void fsk_la_next(bool stage0, const void *chebv, const void *mm, const uint32_t *pos, uint32_t limit, uint32_t period,
                 uint32_t *next, uint32_t *reach, uint32_t x_start, hipStream_t s); // also writes the chain's start mark into reach
void fsk_la_reach(const uint32_t *jin, uint32_t *jout, uint32_t *reach, uint32_t nstates, hipStream_t s);
void fsk_la_mail(const uint32_t *src, uint32_t n, uint32_t *mail, uint32_t seq, hipStream_t s); // n <= 31 words -> host mailbox, then seq in word 31
void fsk_la_reach_all(const uint32_t *next, uint32_t *bufB, uint32_t *bufC, uint32_t *reach, uint32_t nstates, uint32_t rounds,
                      hipStream_t s); // every round in one launch (one workgroup): stages of at most 2^16 states
template <class F>
void fsk_la_records(bool stage0, const void *zref, const void *P, const uint32_t *pos, const uint32_t *next,
                    const uint32_t *reach, const uint32_t *rank, uint32_t limit, uint32_t rank_offset, void *out, void *tail_out,
                    uint32_t max_ref, hipStream_t s); // tail_out: the stage's tail record in the same launch (or null)
// a higher stage's prologue in one launch: scan of the step lengths, first detection, the words the period decision reads
template <class F>
void fsk_la_stage_prologue(const void *P, const void *chebv, const void *mm, const uint32_t *steps, uint32_t *pos, uint32_t count,
                           uint32_t *out, hipStream_t s);
template <class F>
void fsk_la_one_record(bool stage0, const void *zref, const void *P, uint32_t e, uint32_t step_length, void *out, hipStream_t s);
template <class F> void fsk_la_tail(const void *zref, uint32_t max_ref, void *out, uint32_t *zcoeff_zero, hipStream_t s);
template <class F>
void fsk_la_at(const void *las, const uint32_t *stage_la_index, uint32_t stage_count, const void *radius,
               int use_small_exponents, void *at_out, uint32_t *use_at, hipStream_t s);
void fsk_la_pack(bool is64, const void *in, void *out, uint32_t n, hipStream_t s);
// test hook: FS_FAST_LOOP_FDU's block threshold for n (bound, largest scale shift, largest max|dc|) triples (device arrays)
void fsk_test_block_threshold(const int *bw, const int *eshm, const int *sdc, int *t_out, uint32_t n, hipStream_t s);
