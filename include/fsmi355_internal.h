/* fsmi355_internal.h -- measurement hooks, A/B switches and test read-backs of libfsmi355.so.
 *
 * NOT part of the drop-in boundary: nothing here replaces a GPURenderer member and a FractalShark maintainer never calls any of it
 * (include/fsmi355.h is the interface to bind).  bench.py, tests/ and tools/ use these entry points to time kernels, count executed
 * work, select the in-library A/B variants the defaults were measured against, and read device-built tables back for comparison
 * with the host builders.
 */
#ifndef FSMI355_INTERNAL_H
#define FSMI355_INTERNAL_H

#include "fsmi355.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement hooks (this project's addition; bench.py / profiles).
 * fs_last_kernel_ms: duration of the most recent iteration-kernel launch measured with HIP events on the
 * compute stream (valid after fs_sync_compute).
 * fs_enable_step_count: when on, iteration kernels also accumulate the executed work per launch:
 * counts[0] = AT iterations, [1] = LA steps, [2] = perturbation steps, [3] = pixels,
 * [4] = lane slots occupied in the perturbation loop (64 x longest lane, summed over waves),
 * [5] = perturbation steps that went through the careful (exit-tested) path of the tuned LAv2 loop, [6] = steps taken in
 * its scaled runs, [7] = scaled runs started (both per lane). */
float fs_last_kernel_ms(const fs_renderer *r);
/* The durations of the last n (<= 64) iteration-kernel launches, oldest first (each launch keeps its own pair of HIP
 * events, so frames that were in flight together can be read after the fact); valid after fs_sync_compute. */
uint32_t fs_kernel_ms_history(const fs_renderer *r, float *ms_out, uint32_t n);
/* The same launches split where a frame is made of TWO kernels (HDRFloat<double> LAv2: k_at_pass64, then k_lav2_lit<double>):
 * first_ms[k] = the first kernel (0 for a one-kernel frame), second_ms[k] = the rest; first + second = fs_kernel_ms_history. */
uint32_t fs_kernel_ms_split_history(const fs_renderer *r, float *first_ms, float *second_ms, uint32_t n);
/* Low byte: 0 (default) = tuned iteration loops; 1 = literal operation-by-operation transcription of the CPU function;
 * 2 = tuned loops without the scaled runs of the HDRFloat<float> LAv2 kernel (slower; kept as in-library A/B references
 * for the tuned loops -- results are identical).  ORed with A/B flags, both off by default because they measure slower
 * (DESIGN.md 4.3 / 5.2), both bit-identical to the default and under test (tests/test_gpu_variants.py):
 *   FS_VARIANT_LDS_ORBIT  the scaled runs of the tuned HDRFloat<float> LAv2 kernel take their orbit entries through LDS
 *                         (LDS-DMA double buffer per wave) instead of the scalar cache;
 *   FS_VARIANT_REFILL     the HDRFloat<float|double> BLA kernel runs as a persistent launch whose waves refill finished
 *                         lanes from a frame-wide pixel queue (wave-ballot compaction).
 * Unknown values: hipErrorInvalidValue, the selection stays as it was.
 *   FS_VARIANT_WIDE_COUNTERS  (test switch) every entry point launches the instantiation of its kernel that counts
 *                         iterations in 64 bits -- the ones an iteration cap of 2^32 or above selects -- whatever the cap is:
 *                         lets the 64-bit kernels be compared with the CPU functions at caps a test can afford.
 *   FS_VARIANT_NATURAL_TILE_ORDER  fs_render_bla without BLA (perturbation only, HDRFloat<float>) and the tuned
 *                         HDRFloat<float> fs_render_lav2 (self-recorded order, see fs_forget_tile_costs) launch a frame's 8 x 8
 *                         tiles in their natural order.  Default for frames with an iteration limit of 2^18 or more and
 *                         at least 4096 tiles: the tiles that hold long-running pixels first (a probe launch runs every
 *                         tile's centre pixel for n_iterations / 32 steps), ONE to a workgroup with three
 *                         short tiles beside it (never-escaping waves that share a CU slow each other down) --
 *                         which wave renders which tile changes no pixel.
 *                         The same switch keeps the HDRFloat<double> / HDRFloat<CudaDblflt> fs_render_lav2 frames in the tile
 *                         mapping: by default, from the third frame of a view on (the first runs as it is, the second records and
 *                         sorts -- a view that is shown once pays for no sort; a view = same geometry, row bands, orbit, coordinates,
 *                         iteration limit, mode), lane s of the launch renders the pixel that ranked s-th in the previous
 *                         frame -- by iteration count (HDRFloat<double>) or by the cost that frame recorded per pixel, its own AT
 *                         iterations above its perturbation steps (HDRFloat<CudaDblflt>); a device radix sort, once per view;
 *                         frames of 2^20 elements and more -- so that the lanes of a wave run equally long.  Which lane renders
 *                         which pixel changes no pixel.
 *   FS_VARIANT_BLA_POOL   the hand-written HDRFloat<float> BLA kernel (the default of fs_render_bla with a table) re-packs the
 *                         running pixels of a workgroup's four waves into as few waves as possible every 32 trips (LDS exchange).
 *                         A/B, off by default: measured slower; results identical.
 */
enum { FS_VARIANT_LDS_ORBIT = 0x100, FS_VARIANT_REFILL = 0x200, FS_VARIANT_WIDE_COUNTERS = 0x400,
       FS_VARIANT_NATURAL_TILE_ORDER = 0x800, FS_VARIANT_BLA_POOL = 0x1000 };
uint32_t fs_set_kernel_variant(fs_renderer *r, int variant);
/* Longest tiles first, self-recorded (this project's addition).  Every fs_render_lav2 frame of the tuned
 * HDRFloat<float> kernel records one cost word per 8 x 8 tile (its longest lane's step count); the next frame with the
 * same geometry, row bands and orbit generation is launched in descending cost order -- "warm".  The first frame, a frame
 * after any of those changed or after fs_forget_tile_costs, and every frame under FS_VARIANT_NATURAL_TILE_ORDER run in
 * natural order -- "cold".  The order changes which wave renders which tile, never a pixel.
 * fs_render_bla's probe order (FS_VARIANT_NATURAL_TILE_ORDER above) is kept the same way: a frame with the same geometry, row
 * bands, orbit (generation, or for generation 0 a sampled fingerprint of the entries -- RenderPerturbBLA re-uploads per call),
 * coordinates and iteration limit as the one before reuses the order and skips the probe launch; fs_forget_tile_costs drops it
 * -- and the pixel order of the HDRFloat<double> / HDRFloat<CudaDblflt> LAv2 frames (FS_VARIANT_NATURAL_TILE_ORDER above).
 * fs_last_frame_tile_ordered: 1 when the most recent fs_render_lav2 launch used a recorded order / the most recent
 * perturbation-only fs_render_bla launch reused its probe order.
 * fs_last_frame_sampled_tile_order (round 6): 1 when the most recent fs_render_lav2 launch was a view's FIRST frame of the
 * HDRFloat<double> / HDRFloat<CudaDblflt> kernels with its tiles in the order of a sampled PerformAT count
 * (csrc/kernels_tile_sample.hip; A/B switch: environment FSMI355_COLD_TILE_ORDER=0, FS_VARIANT_NATURAL_TILE_ORDER).
 * fs_read_tile_costs: the costs the last frame recorded (row-major tiles of the LOCAL buffer, (width + 7) / 8 per row);
 * *n_tiles = their number; out may be NULL.  FractalSharkError 10006 when nothing has been recorded. */
uint32_t fs_forget_tile_costs(fs_renderer *r);
int fs_last_frame_tile_ordered(fs_renderer *r);
int fs_last_frame_sampled_tile_order(fs_renderer *r);
uint32_t fs_read_tile_costs(fs_renderer *r, uint32_t *out, uint64_t max_words, uint64_t *n_tiles);
/* Test hook for the waypoint-resident orbit (fs_set_compressed_orbit_mode(1), HDRFloat<float | double>): one lane's
 * decompression cursor -- with 32-bit positions, or the 64-bit ones the wide kernel uses -- seeks to orbit index `start`
 * and walks n entries on; out[k] = the orbit value at start + k as {float re, im; int32 e} (12 B) or
 * {double re, im; int32 e; pad} (24 B).  Indices of 2^32 and above need wide_positions = 1. */
uint32_t fs_seq_cursor_probe(fs_renderer *r, int wide_positions, uint64_t start, uint32_t n, void *out);
/* The launch order of the most recent frame when it was an ordered one (its first n_tiles words: a permutation of the tile
 * numbers, highest cost class first); 10006 otherwise.  For tests. */
uint32_t fs_read_tile_order(fs_renderer *r, uint32_t *out, uint64_t max_words);
uint32_t fs_enable_step_count(fs_renderer *r, int enable);
uint32_t fs_read_step_count(fs_renderer *r, uint64_t counts[8]);
/* The whole statistics buffer (measurement builds append per-wave trace records behind the 8 counters: library built
 * with FS_TRACE_WAVES=1 and FSMI355_TRACE_WAVES=<max waves> in the environment; tools/wave_trace.py). */
uint32_t fs_read_stats_raw(fs_renderer *r, uint64_t *out, uint64_t max_words);
/* Test hook: the wave-uniform block threshold of the tuned HDRFloat<float> LAv2 loop (csrc/kernels.hip, FS_FAST_LOOP_FDU), evaluated on
 * the device by the macro the loop itself uses, for n triples (block bound as a binary32 bit pattern -- 0x80000000 = "never" --, largest
 * scale shift of the running lanes, largest max|dc| as a bit pattern): threshold_out[i] = -1 when dc_bits > bound_bits, else
 * min(bound_bits - scale_shift, 0x46800000) without wrap-around.  Host arrays.  tests/test_gpu_block_threshold.py. */
uint32_t fs_test_block_threshold(fs_renderer *r, const int32_t *bound_bits, const int32_t *scale_shift, const int32_t *dc_bits,
                                 int32_t *threshold_out, uint32_t n);
/* Average duration (HIP events on the compute stream, `repeats` back-to-back launches, no D2H) of the two RenderCurrent
 * kernels over the current iteration buffer: ms_out[0] = antialias + palette, ms_out[1] = min / max / sum.  Needs a
 * palette (fs_init_memory) and the whole frame on this renderer.  tools/bench_render_current.py turns them into GB/s. */
uint32_t fs_time_render_current(fs_renderer *r, uint64_t n_iterations, uint32_t repeats, float ms_out[2]);


/* ---- read-backs of device-built tables (tests, tools) */
/* Table geometry / contents after fs_build_bla or fs_upload_bla (tests, tools): number of levels (m_B.size()), m_LM2,
 * records per level; fs_read_bla_level copies one level to the host (synchronises the compute stream). */
int32_t fs_bla_num_levels(const fs_renderer *r);
int32_t fs_bla_lm2(const fs_renderer *r);
uint64_t fs_bla_level_size(const fs_renderer *r, int32_t level);
uint32_t fs_read_bla_level(fs_renderer *r, int32_t level, void *out, uint64_t max_records);
uint32_t fs_la_counts(const fs_renderer *r, uint32_t *n_las, uint32_t *n_stages, int *use_at, int *is_valid);
uint32_t fs_read_la(fs_renderer *r, void *las_out, uint32_t max_las, void *stages_out, uint32_t max_stages, void *at_out);

/* Duration of the last gather + reassembly on device 0 (HIP events; synchronises device 0's post stream). */
float fs_group_gather_ms(fs_group *g);

#ifdef __cplusplus
}
#endif

#endif /* FSMI355_INTERNAL_H */
