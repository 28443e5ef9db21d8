/* fsmi355.h -- C ABI of libfsmi355.so, the MI355X (gfx950) per-pixel renderer for FractalShark.
 *
 * One opaque fs_renderer corresponds to one reference `GPURenderer` object
 * (FractalSharkLib/GPU_Render.h:20-227, implemented by FractalSharkGpuLib/GPU_Render.cu).  Every entry
 * point below names the GPURenderer member it replaces; fractalshark_amd/csrc/gpu_render_shim.hpp defines
 * those members (same names, argument order and error behaviour) by forwarding here, so Fractal.cpp keeps
 * compiling against GPU_Render.h unchanged.  INTEGRATION.md shows the binding.
 *
 * Conventions (same as the reference):
 *   - every call returns uint32_t; 0 = success; non-zero = hipError_t value or FS_ERR_* (10000+);
 *   - host pointers are borrowed for the duration of the call only; device memory is owned by the renderer;
 *   - one renderer is used by one host thread at a time; different renderers may run concurrently;
 *   - "memory not initialised" makes render calls return 0 without doing anything (GPU_Render.cu:564-566).
 *
 * Record layouts: fs_layout.h.  Type tags select the numeric type T of the reference templates.
 *
 * This header is the whole interface a FractalShark maintainer binds: what gpu_render_shim.hpp forwards to, the device-side
 * table builders, and the multi-GPU group.  Measurement hooks, A/B switches and test read-backs the library also exports
 * (bench.py, tests/, tools/) are declared in fsmi355_internal.h and are not part of the drop-in boundary.
 */
#ifndef FSMI355_H
#define FSMI355_H

#include <stddef.h>
#include <stdint.h>

#include "fs_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_renderer fs_renderer;

/* FractalSharkError, GPU_Render.cu:33-43 (enum starts at 10000). */
enum {
    FS_OK = 0,
    FS_ERR_1 = 10000,
    FS_ERR_2 = 10001,
    FS_ERR_3 = 10002, /* bad antialiasing */
    FS_ERR_4 = 10003, /* width not divisible by antialiasing */
    FS_ERR_5 = 10004, /* height not divisible by antialiasing */
    FS_ERR_6 = 10005, /* no uploaded orbit / table for this type */
    FS_ERR_7 = 10006,
    FS_ERR_UNSUPPORTED = 10100 /* type tag / mode not built into this library (this project's addition) */
};

/* Numeric type T of the reference's kernel templates. */
enum {
    FS_T_F32 = 0,       /* float                      (Gpu1x32*)   */
    FS_T_F64 = 1,       /* double                     (Gpu1x64*)   */
    FS_T_2X32 = 2,      /* CudaDblflt<dblflt>         (Gpu2x32*)   */
    FS_T_HDR32 = 3,     /* HDRFloat<float>            (GpuHDRx32*) */
    FS_T_HDR64 = 4,     /* HDRFloat<double>           (GpuHDRx64*) */
    FS_T_HDR2X32 = 5,   /* HDRFloat<CudaDblflt<..>>   (GpuHDRx2x32*) */
    FS_T_2X64 = 6,      /* MattDbldbl (double-double) (Gpu2x64)      */
    FS_T_4X32 = 7,      /* MattQFltflt (quad-float)   (Gpu4x32)      */
    FS_T_4X64 = 8       /* MattQDbldbl (quad-double)  (Gpu4x64)      */
};

/* LAv2Mode, FractalSharkLib/RenderAlgorithm.h:12-17. */
enum { FS_LAV2_FULL = 0, FS_LAV2_PO = 1, FS_LAV2_LAO = 2 };

/* Which arithmetic the kernels restate.  The reference's CUDA kernels and its CPU RenderAlgorithm functions
 * are not bit-identical to each other; the parity target of this project is the CPU
 * functions, so FS_PARITY_CPU is the default everywhere.
 *   FS_PARITY_CPU      : literal CPU functions, including LAReference::isLAStageInvalid as written
 *                        (LAReference.cpp:1076-1081: stage skipped when cheb(dc) <  LAThresholdC).
 *   FS_PARITY_CPU_GPUSTAGE : CPU arithmetic, but the stage-validity test in the direction the reference's GPU
 *                        twin (and FractalZoomer) uses (GPU_LAReference.h:240-254: skipped when cheb(dc) >= ...).
 */
enum { FS_PARITY_CPU = 0, FS_PARITY_CPU_GPUSTAGE = 1 };

/* GPURenderer::GPURenderer / ~GPURenderer.  `device` = HIP device ordinal (the reference hard-codes 0,
 * GPU_Render.cu:113; the multi-GPU row tiler passes LOCAL_RANK). */
fs_renderer *fs_create(int device);
void fs_destroy(fs_renderer *r);

/* GPURenderer::TestCudaIsWorking (GPU_Render.cu:100-123): NON-ZERO = a usable device exists. */
uint32_t fs_test_device_is_working(void);

/* Number of HIP devices this process can see (0 when there is none or the runtime fails): what a multi-GPU host sizes
 * fs_group_create with.  This project's addition (the reference is single-device). */
int fs_device_count(void);

/* GPURenderer::ConvertErrorToString (GPU_Render.cu:1820-1823). */
const char *fs_error_string(uint32_t err);

/* GPURenderer::InitializeMemory<IterType> (GPU_Render.cu:232-407).  w,h already include antialiasing.
 * iter_bytes = sizeof(IterType) (4 or 8).  The iteration buffer is padded to 16 columns x 8 rows
 * (GPU_Render.cu:334-344).
 * IterType = uint64_t: the iteration buffer, RenderCurrent and ReductionResults work on uint64_t elements and the
 * uint64_t record layouts (fs_la_*_u64, fs_la_stage_u64, fs_at_*_u64) are accepted (table step lengths / indices must
 * fit 32 bits -- a table that large would not fit any device -- else FS_ERR_UNSUPPORTED).  Iteration caps of 2^32 and
 * above are served by every render entry point and numeric type: each kernel is also instantiated with 64-bit counters, as
 * the reference's kernels are templated on IterType (GPU_Render.cu:849-991, 1204-1300, 1380-1436, 1610-1692) -- for the
 * HDRFloat<float> perturbation kernels the literal form (the tuned runs budget their steps in 32 bits).  Such a cap with a
 * 4-byte buffer is a caller error (hipErrorInvalidValue): a uint32_t IterType cannot hold the count (every built-in view
 * selects Bits32, FractalViewPresets.cpp:19). */
uint32_t fs_init_memory(fs_renderer *r, uint32_t w, uint32_t h, uint32_t antialiasing, uint32_t iter_bytes,
                        const fs_color16 *pal_interleaved, uint32_t pal_iters, uint32_t palette_aux_depth,
                        uint64_t palette_generation, int expected_reuse);

/* Row band for multi-GPU tiling (this project's addition; the reference is single-device).  The renderer
 * keeps the full-frame geometry for the pixel -> delta-c mapping (Y stays the global row so
 * DeltaImaginary = -dy*T(Y) - centerY rounds identically) but only computes global rows
 * [band_first_row + k*band_stride_rows, +band_rows) for k = 0,1,...  The local iteration buffer then holds
 * the owned bands back to back.  Defaults: first 0, rows = h, stride = h (whole frame). */
uint32_t fs_set_row_bands(fs_renderer *r, uint32_t band_first_row, uint32_t band_rows, uint32_t band_stride_rows);
/* Number of rows the local buffer holds under the current banding (after padding to 8). */
uint32_t fs_local_rows(const fs_renderer *r);

/* Optional: render into caller-owned DEVICE memory (e.g. a torch tensor handed to RCCL) instead of the
 * internal buffer.  capacity_bytes = size of the caller's allocation: it must hold fs_local_rows() x rounded-width
 * elements of sizeof(IterType), otherwise hipErrorInvalidValue is returned and the internal buffer stays in use.
 * fs_init_memory with a new geometry drops the pointer; fs_set_row_bands re-validates it against the new banding
 * (and falls back to the internal buffer with an error when it is too small).  NULL restores the internal buffer. */
uint32_t fs_set_external_iter_buffer(fs_renderer *r, void *device_ptr, uint64_t capacity_bytes);
void *fs_device_iter_buffer(const fs_renderer *r);
uint32_t fs_rounded_width(const fs_renderer *r);
/* The sharded read-back of a row-tiled frame (this project's addition; the reference copies the whole iteration buffer through
 * its one device, GPURenderer::ExtractItersAndColors, GPU_Render.cu:1760-1805): the renderer's owned bands go straight to THEIR
 * rows of a whole-frame host buffer -- rows in frame order, row pitch = fs_rounded_width() x sizeof(IterType), (height padded to
 * 8) rows -- over this device's own PCIe link: one two-dimensional copy whose destination pitch is the band stride, so nothing has
 * to restore row order afterwards and no frame funnels through device 0.  device_iters = the buffer the frame was rendered into
 * (NULL = the renderer's current iteration buffer); stream = a hipStream_t of this device (NULL = the compute stream: behind the
 * kernel).  host_frame should be page-locked for every device that writes into it (hipHostMalloc with hipHostMallocPortable, or
 * fs_host_register -- e.g. a POSIX shared-memory frame that several rank processes fill); pageable memory works, synchronously.
 * Asynchronous.  Rows at and beyond the frame's height (the padding to 8) are only written without row bands, where the call is
 * the plain copy of the whole padded buffer. */
uint32_t fs_copy_bands_to_host(fs_renderer *r, const void *device_iters, void *host_frame, void *stream);
/* hipHostRegister (portable: usable by every device of the process) / hipHostUnregister of caller-owned host memory. */
uint32_t fs_host_register(void *host_ptr, uint64_t bytes);
uint32_t fs_host_unregister(void *host_ptr);

/* GPURenderer::InitializePerturb<IterType,T1,SubType,PExtras,T2> (GPU_Render.cu:431-501), split in two:
 * orbit upload (GPUPerturbSingleResults ctor, Perturb.cuh:20-80) ... */
uint32_t fs_upload_orbit(fs_renderer *r, uint64_t generation, int type_tag, uint32_t iter_bytes,
                         const void *entries, uint64_t orbit_size, uint64_t uncompressed_size,
                         uint64_t period_maybe_zero);
/* The same for PerturbExtras::SimpleCompression orbits: `entries` = GPUReferenceIter<T, SimpleCompression>[compressed_size]
 * (fs_orbit_hdr32_rc / _hdr64_rc / _f32_rc / _f64_rc / _p2x32_rc / _2x32_rc by type_tag), orbit_x_low / orbit_y_low =
 * GPUPerturbResults::OrbitXLow / OrbitYLow in the same numeric type (fs_real_hdr32 / fs_real_hdr64 / float / double /
 * fs_real_p2x32 / fs_real_2x32): the constant c of the runtime decompressor, Perturb.cuh:272-326.  The orbit is expanded
 * once on the device, in T arithmetic; all render calls then behave exactly as the reference's *RC* algorithms (for
 * HDRFloat<float|double>, float and double bit-identical to its CPU RuntimeDecompressor too).  All six numeric types. */
uint32_t fs_upload_orbit_compressed(fs_renderer *r, uint64_t generation, int type_tag, uint32_t iter_bytes,
                                    const void *entries, uint64_t compressed_size, uint64_t uncompressed_size,
                                    uint64_t period_maybe_zero, const void *orbit_x_low, const void *orbit_y_low);
/* How fs_upload_orbit_compressed keeps an orbit in HBM (set it before the upload; all six numeric types):
 *   0 (default)  expanded once on upload into the full orbit (fastest kernels; costs uncompressed_size entries of HBM);
 *   1            only the waypoints stay resident and every pixel decompresses the orbit as it walks it, with a sequential
 *                cursor -- GPUPerturbSingleResults::SeqWorkspace / GetIterSeq / BinarySearch (Perturb.cuh:146-326), the
 *                reason the format exists: orbits too long to hold expanded.  Same orbit values bit for bit, hence the same
 *                frames.  Served by fs_render_lav2 (all modes except perturbation-only with FS_PARITY_CPU, whose twin is the
 *                scalar kernel; every iteration cap -- 2^32 and above with a uint64_t iteration buffer); fs_render_bla /
 *                fs_build_la / fs_build_bla need the expanded orbit and return FS_ERR_UNSUPPORTED.  float / double / CudaDblflt /
 *                HDRFloat<CudaDblflt> orbits are walked the same way by their LAv2 kernels (GPU_Render.cu:518-523,532-537
 *                instantiates SimpleCompression for them too), with 32-bit positions.  For HDRFloat<float | double> orbit
 *                POSITIONS are IterType-wide like the reference's (Perturb.cuh:21-23,202-203,247-271; LAInfoI.h:5-19): an orbit of 2^32 and more uncompressed entries (only its waypoints are
 *                resident), a period, LA step lengths / next-stage indices and an AT step length beyond 32 bits are accepted
 *                -- fs_upload_la then keeps the uint64_t records as they are -- and the kernel walks them with 64-bit
 *                positions.  (Expanded orbits are limited to 2^32 - 1 entries: FS_ERR_UNSUPPORTED above that.)
 * fs_orbit_device_bytes: HBM bytes the resident orbit occupies in all its device forms (0 without an orbit). */
uint32_t fs_set_compressed_orbit_mode(fs_renderer *r, int mode);
uint64_t fs_orbit_device_bytes(const fs_renderer *r);
/* ... and LA table upload (GPU_LAReference ctor, GPU_LAReference.h:79-160).  at_info may be NULL when
 * use_at == 0.  iter_bytes selects the record family: 4 -> fs_la_*_u32 / fs_la_stage_u32 / fs_at_*_u32,
 * 8 -> fs_la_*_u64 / fs_la_stage_u64 / fs_at_*_u64 (narrowed on upload; kept as they are for HDRFloat<float | double> under
 * fs_set_compressed_orbit_mode(1) when a step length or index does not fit 32 bits). */
uint32_t fs_upload_la(fs_renderer *r, uint64_t generation, int type_tag, uint32_t iter_bytes, const void *las,
                      uint32_t n_las, const void *stages, uint32_t n_stages, int is_valid, int use_at,
                      const void *at_info);

/* BLA table upload (GPU_BLAS ctor, BLA.cuh:123-160); the reference does this inside RenderPerturbBLA. */
uint32_t fs_upload_bla(fs_renderer *r, int type_tag, const void *const *levels, const uint64_t *level_sizes,
                       int32_t n_levels, int32_t lm2);

/* BLAS<IterType,T>::Init(count, blaSize) (FractalSharkLib/BLAS.cpp:212-255) executed on the device instead of the host:
 * builds the BLA table of the orbit last uploaded with fs_upload_orbit (type_tag FS_T_HDR32 / FS_T_HDR64) directly in
 * HBM and installs it as the renderer's table, bit-identical to the host builder.  bla_size = the orbit's max radius
 * (fs_real_hdr32 / fs_real_hdr64, PerturbationResults::GetMaxRadius, Fractal.cpp:2739-2740).  Asynchronous on the compute
 * stream.  The reference rebuilds this table on the CPU and copies it over PCIe on every BLA render. */
uint32_t fs_build_bla(fs_renderer *r, int type_tag, const void *bla_size);

/* LAReference::GenerateApproximationData (FractalSharkLib/LAReference.cpp:971-1013: CreateLAFromOrbit :28-210,
 * CreateNewLAStage :774-966, CreateATFromLA :1050-1074) executed on the DEVICE instead of the host: builds the LAv2 table
 * (all stages + ATInfo) of the orbit last uploaded with fs_upload_orbit / fs_upload_orbit_compressed (type_tag FS_T_HDR32 /
 * FS_T_HDR64) directly in HBM and installs it as the renderer's table, as fs_upload_la would.  max_radius =
 * PerturbationResults::GetMaxRadius (fs_real_hdr32 / fs_real_hdr64); use_small_exponents = the UseSmallExponents flag of
 * CreateATFromLA (RefOrbitCalc.cpp:2346).  The table is bit-identical to the reference's SINGLE-THREADED builder (what
 * LAReference produces when hardware_concurrency() < 2 or the orbit has fewer than 2 * 50000 entries, :236-251; the
 * multi-threaded stage-0 variant yields a thread-count-dependent table: fs_build_la_mt below).  An orbit of at most 64 steps (LowBound,
 * LAReference.h:56) in which no period is found gets the reference's two records and a table that is not valid (:135-140,
 * :1002-1005; fs_la_counts reports is_valid 0 and the kernels ignore the table); one in which periods are found gets its
 * normal small table.  FS_ERR_UNSUPPORTED (use fs_upload_la): an orbit of fewer than three entries, a first step whose ZCoeff
 * is zero.  Synchronous.  (fs_la_counts / fs_read_la of fsmi355_internal.h read the installed table back: tests, tools.) */
uint32_t fs_build_la(fs_renderer *r, int type_tag, const void *max_radius, int use_small_exponents);
/* ... and with stage 0 as LAReference::CreateLAFromOrbitMT (LAReference.cpp:215-770) builds it on a host with `host_threads`
 * hardware threads (std::thread::hardware_concurrency(), :236-240): ThreadCount = min(maxRefIteration / 50000, host_threads)
 * pieces -- the Starter from the prologue, Worker k from the first period boundary it finds after orbit index
 * maxRefIteration * k / ThreadCount (:486-560), each until it meets the next published start (:640-668), stitched (:711-760) --
 * so that the table is, bit for bit, the one FractalShark's own CPU builder makes on THAT host (it differs from the
 * single-threaded table near every piece boundary).  ThreadCount < 2 (host_threads < 2 or an orbit below 100 000 entries) is
 * fs_build_la.  The device computes next() for every state of the scan and the 2 (ThreadCount - 1) uncapped first detections;
 * the host walks the chains (indices only: 8 bytes per orbit entry come back once) and stitches; the device folds the records. */
uint32_t fs_build_la_mt(fs_renderer *r, int type_tag, const void *max_radius, int use_small_exponents, int host_threads);

/* GPURenderer::RenderPerturbLAv2<IterType,T,SubType,Mode,PExtras> (GPU_Render.cu:995-1188) for all six numeric types of
 * its instantiation list (:1204-1300): FS_T_HDR32 / FS_T_HDR64 / FS_T_HDR2X32 (GpuHDRx32 / x64 / x2x32 PerturbedLAv2*)
 * and the non-HDR FS_T_F32 / FS_T_F64 / FS_T_2X32 (Gpu1x32 / Gpu1x64 / Gpu2x32 PerturbedLAv2*; orbit, table and ATInfo
 * in the "plain" record families of fs_layout.h, uploaded with the same type_tag).
 * coords = {dx, dy, centerX, centerY} in the type selected by type_tag (fs_real_hdr32[4], fs_real_hdr64[4],
 * fs_real_2x32[4], float[4], double[4], fs_real_p2x32[4]).  `parity` only applies to FS_T_HDR32 / FS_T_HDR64 (the types
 * with a CPU twin).  Asynchronous on the compute stream. */
uint32_t fs_render_lav2(fs_renderer *r, int type_tag, int mode, int parity, const void *coords,
                        uint64_t n_iterations);

/* GPURenderer::RenderPerturbBLA<IterType,T> (GPU_Render.cu:1440-1607); uses the orbit of fs_upload_orbit and
 * the table of fs_upload_bla (n_levels == 0 there = plain perturbation). */
uint32_t fs_render_bla(fs_renderer *r, int type_tag, const void *coords, uint64_t n_iterations);

/* GPURenderer::Render<IterType,T> (GPU_Render.cu:617-846), direct escape-time kernels.
 * coords = {dx, dy, minX, maxY} (doubles for FS_T_F64), CPU twin Fractal::CalcCpuHDR (Fractal.cpp:2096-2206). */
uint32_t fs_render_direct(fs_renderer *r, int type_tag, const void *coords, uint64_t n_iterations);

/* GPURenderer::RenderPerturbBLAScaled<IterType,T> (GPU_Render.cu:1302-1376) -> mandel_1x_float_perturb_scaled
 * (ScaledKernels.cuh:3-239), T = HDRFloat<float> (FS_T_HDR32, RenderAlgorithm GpuHDRx32PerturbedScaled).
 * fs_upload_orbit_scaled takes the two PerturbExtras::Bad orbits the reference passes into the render call:
 * entries_t = GPUReferenceIter<HDRFloat<float>,Bad>[n] (fs_orbit_hdr32_bad), entries_f32 = GPUReferenceIter<float,Bad>[n]
 * (fs_orbit_f32_bad).  coords = fs_real_hdr32{dx, dy, centerX, centerY}.  No CPU RenderAlgorithm exists for this path:
 * the kernel restates the reference's CUDA kernel with un-contracted IEEE binary32 arithmetic. */
uint32_t fs_upload_orbit_scaled(fs_renderer *r, int type_tag, uint32_t iter_bytes, const void *entries_t,
                                const void *entries_f32, uint64_t orbit_size, uint64_t period_maybe_zero);
uint32_t fs_render_scaled(fs_renderer *r, int type_tag, const void *coords, uint64_t n_iterations);

/* GPURenderer::Render<IterType,T> for the direct kernels that have no CPU RenderAlgorithm twin (LowPrecisionKernels.cuh):
 * FS_T_F32 -> mandel_1x_float (Gpu1x32), coords = float{cx, cy, dx, dy};  FS_T_2X32 -> mandel_2x_float (Gpu2x32),
 * coords = float{cx.head, cx.tail, cy.head, cy.tail, dx.head, dx.tail, dy.head, dy.tail} (MattDblflt);  FS_T_2X64 ->
 * mandel_2x_double (Gpu2x64), the same eight values as doubles (MattDbldbl);  FS_T_4X32 -> mandel_4x_float (Gpu4x32),
 * coords = float{cx.x..w, cy.x..w, dx.x..w, dy.x..w} (MattQFltflt, most significant first);  FS_T_4X64 -> mandel_4x_double
 * (Gpu4x64), the same sixteen values as doubles (MattQDbldbl).  cx / cy = the view's MIN corner
 * (Fractal::FillGpuCoords, Fractal.cpp:1833-1844); rows are written flipped like the reference.  iteration_precision in
 * {1, 4, 8, 16} (ignored by Gpu2x64 / Gpu4x32 / Gpu4x64); other values launch nothing and return 0, like the reference's switch.  The kernels
 * restate the CUDA kernels (no CPU twin exists; un-contracted IEEE arithmetic, __fmaf_rd = fma rounded toward -inf). */
uint32_t fs_render_direct_lp(fs_renderer *r, int type_tag, const void *coords, uint64_t n_iterations,
                             int iteration_precision);

/* GPURenderer::ClearMemory<IterType> (GPU_Render.cu:212-225). */
uint32_t fs_clear(fs_renderer *r);

/* GPURenderer::RenderCurrent<IterType> (GPU_Render.cu:556-581): antialias + palette, min/max/sum, D2H.
 * Any of the three host pointers may be NULL.  progressive != 0 runs on the display stream. */
uint32_t fs_render_current(fs_renderer *r, uint64_t n_iterations, void *iter_buffer, fs_color16 *color_buffer,
                           fs_reduction *reduction, int progressive);

/* RenderCurrent's colour half (RunAntialiasing, GPU_Render.cu:1695-1757) over a whole frame that lies elsewhere on the
 * renderer's device -- the frame an fs_group has gathered and put back in row order: antialias + palette with this renderer's
 * palette and geometry into device_colors (NULL = the renderer's own colour buffer), then, if color_buffer is not NULL, the
 * copy of N_color_cu Color16 to the host, all on `stream` (a hipStream_t of that device).  fs_color_buffer_elements =
 * N_color_cu, the element count of a colour buffer (padded to 16 x 8 blocks like the reference's). */
uint32_t fs_colorize_frame(fs_renderer *r, const void *device_iters, uint64_t n_iterations, fs_color16 *device_colors,
                           fs_color16 *color_buffer, void *stream);
uint64_t fs_color_buffer_elements(const fs_renderer *r);

/* SyncComputeStream / SyncDisplayStream / QueryComputeStream / EnqueueComputeDoneCallback
 * (GPU_Render.cu:596-615).  The callback runs on a runtime thread after all prior compute-stream work. */
uint32_t fs_sync_compute(fs_renderer *r);
uint32_t fs_sync_display(fs_renderer *r);
/* The compute stream itself (a hipStream_t; GPURenderer::m_ComputeStream, GPU_Render.h), NULL before fs_init_memory.  For a
 * host that chains its own device work behind a render without a host round trip (the multi-GPU gather of bench.py waits
 * on it from its own stream).  Owned by the renderer. */
void *fs_compute_stream(const fs_renderer *r);
/* The display stream (GPURenderer::m_DisplayStream: high priority, where a progressive RenderCurrent runs). */
void *fs_display_stream(const fs_renderer *r);
uint32_t fs_query_compute(fs_renderer *r);
typedef void (*fs_done_cb)(void *user);
uint32_t fs_enqueue_done_callback(fs_renderer *r, fs_done_cb cb, void *user);

/* Memory behaviour (GPU_Render.cu:127,142-153,362-395; Perturb.cuh:51-61; GPU_LAReference.h:93-113): device memory is
 * allocated with hipMalloc / hipFree behind a synchronisation of the compute stream (FSMI355_ASYNC_ALLOC=1 in the environment
 * selects hipMallocAsync / hipFreeAsync in compute-stream order instead -- off by default); the work memory of fs_build_la, the installed
 * LA table and the BLA table (one allocation for all levels) are kept and reused when the next one fits.  When the device
 * cannot hold an INPUT table (reference orbit, LA table, BLA table) it is placed in page-locked host memory instead and the
 * kernels read it over the bus, as the reference does -- the frame still renders, slowly.  fs_host_fallback_bytes = bytes
 * ever placed there by this renderer (0 in normal operation).  FSMI355_FAIL_INPUT_ALLOC=1 in the environment when
 * fs_create runs makes every input allocation of that renderer take the fallback (fault injection for tests). */
uint64_t fs_host_fallback_bytes(const fs_renderer *r);
/* Device memory a renderer keeps idle for its next allocation (blocks it has freed: at most 16 blocks / 2 GiB, reused when
 * the next request fits).  A renderer whose allocation fails frees its own idle blocks, then those of EVERY renderer of the
 * process on the same device (FractalShark holds four GPURenderers on one device), and only then takes the page-locked
 * fallback or reports the error.  fs_idle_device_bytes: what `r` holds idle now; fs_release_idle_device_memory: frees the idle
 * blocks of all renderers on `device` (what the out-of-memory path does), returns the bytes freed.  Thread-safe. */
uint64_t fs_idle_device_bytes(fs_renderer *r);
uint64_t fs_release_idle_device_memory(int device);

uint32_t fs_get_width(const fs_renderer *r);
uint32_t fs_get_height(const fs_renderer *r);

/* ---- Multi-GPU: one frame row-tiled over the GPUs of one node, gathered over xGMI with RCCL (this project's addition;
 * the reference is single-device, GPU_Render.cu:113).  A group = one fs_renderer per device inside ONE process (the C++
 * drop-in of INTEGRATION.md is one process), RCCL communicators from ncclCommInitAll.  Rank r renders the 8-row bands
 * k*N + r (24-row bands for antialiasing 3) with the global-row delta-c mapping of fs_set_row_bands; slices are sent to
 * device 0 with ncclSend / ncclRecv in one ncclGroup on the members' compute streams, one kernel restores row order.
 * transport: 0 = RCCL (resolved with dlopen at first use; falls back to 1 with a message on stderr if unavailable; on hosts that only
 * support dmabuf IPC -- this pool -- HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment before the process's FIRST HIP call: the
 * library sets it when it is loaded unless the host chose a value, so load it before anything touches HIP, or export the variable),
 * 1 = hipMemcpyPeerAsync (also chosen automatically when members share a device, e.g. tests on a one-GPU box).
 * Every fs_group_* upload / render call is the fs_* call of the same name applied to every member; renders are
 * asynchronous.  fs_group_render_current = gather + row reassembly + min / max / sum + D2H of the padded iteration buffer
 * (same layout as fs_render_current of a single renderer on the whole frame), asynchronous on member 0's compute stream;
 * fs_group_sync waits for everything.  Colours of the whole frame: fs_group_render_current_colors. */
typedef struct fs_group fs_group;
fs_group *fs_group_create(const int *devices, int n_devices, int transport);
void fs_group_destroy(fs_group *g);
int fs_group_size(const fs_group *g);
int fs_group_transport(const fs_group *g);
fs_renderer *fs_group_renderer(fs_group *g, int rank);
uint32_t fs_group_init_memory(fs_group *g, uint32_t w, uint32_t h, uint32_t antialiasing, uint32_t iter_bytes,
                              const fs_color16 *pal_interleaved, uint32_t pal_iters, uint32_t palette_aux_depth,
                              uint64_t palette_generation);
uint32_t fs_group_upload_orbit(fs_group *g, uint64_t generation, int type_tag, uint32_t iter_bytes, const void *entries,
                               uint64_t orbit_size, uint64_t uncompressed_size, uint64_t period_maybe_zero);
uint32_t fs_group_upload_orbit_compressed(fs_group *g, uint64_t generation, int type_tag, uint32_t iter_bytes,
                                          const void *entries, uint64_t compressed_size, uint64_t uncompressed_size,
                                          uint64_t period_maybe_zero, const void *orbit_x_low, const void *orbit_y_low);
uint32_t fs_group_upload_la(fs_group *g, uint64_t generation, int type_tag, uint32_t iter_bytes, const void *las,
                            uint32_t n_las, const void *stages, uint32_t n_stages, int is_valid, int use_at,
                            const void *at_info);
uint32_t fs_group_upload_bla(fs_group *g, int type_tag, const void *const *levels, const uint64_t *level_sizes,
                             int32_t n_levels, int32_t lm2);
uint32_t fs_group_upload_orbit_scaled(fs_group *g, int type_tag, uint32_t iter_bytes, const void *entries_t,
                                      const void *entries_f32, uint64_t orbit_size, uint64_t period_maybe_zero);
uint32_t fs_group_render_lav2(fs_group *g, int type_tag, int mode, int parity, const void *coords, uint64_t n_iterations);
uint32_t fs_group_render_bla(fs_group *g, int type_tag, const void *coords, uint64_t n_iterations);
uint32_t fs_group_render_scaled(fs_group *g, int type_tag, const void *coords, uint64_t n_iterations);
uint32_t fs_group_render_direct(fs_group *g, int type_tag, const void *coords, uint64_t n_iterations);
uint32_t fs_group_clear(fs_group *g);
uint32_t fs_group_render_current(fs_group *g, uint64_t n_iterations, void *iter_buffer, fs_reduction *reduction);
/* GPURenderer::RenderCurrent<IterType>(n, iters, colors, reduction, progressive) (GPU_Render.cu:556-581) for the group: as
 * fs_group_render_current, plus the Color16 buffer of the whole frame (antialias + palette on device 0 behind the row order,
 * N_color_cu = fs_color_buffer_elements(fs_group_renderer(g, 0)) elements, same layout as fs_render_current's).  Any of the
 * three host pointers may be NULL.  progressive != 0: a SNAPSHOT of the frame while the members' kernels are still writing it
 * -- nothing waits for a kernel; the slices are copied as they are on the members' display streams into buffers of their own
 * (it does not take part in the two-set rotation and does not count for fs_group_wait_current); fs_group_sync_display waits
 * for it. */
uint32_t fs_group_render_current_colors(fs_group *g, uint64_t n_iterations, void *iter_buffer, fs_color16 *color_buffer,
                                        fs_reduction *reduction, int progressive);
/* Where the iteration buffer of fs_group_render_current[_colors] travels (round 6).  0 (default) = gather: the slices go to device 0
 * over xGMI, one kernel restores row order, ONE copy brings the whole frame to the host over device 0's PCIe link.  1 = direct: every
 * member copies its own bands straight to their rows of iter_buffer over ITS OWN link (fs_copy_bands_to_host, on a copy stream of its
 * own behind its kernel, while it renders the next frame into its second slice) -- N links instead of one, no re-order kernel; the
 * gather then only runs when color_buffer or reduction is asked for (they need the frame on one device), and the frame itself no
 * longer crosses device 0's link.  iter_buffer should be page-locked and portable (hipHostMalloc(hipHostMallocPortable) /
 * fs_host_register).  Same frames either way (tests/test_gpu_group.py). */
uint32_t fs_group_set_host_path(fs_group *g, int host_path);
int fs_group_host_path(const fs_group *g);
uint32_t fs_group_sync_display(fs_group *g);
uint32_t fs_group_sync(fs_group *g);
/* Two frames may be in flight: fs_group_render_current runs on a stream of its own on device 0 (receive, row order,
 * reduction, D2H) behind the members' kernels, over one of TWO sets of gather / frame buffers used in rotation, so the
 * members' next frame can be launched right away and renders while this one is delivered.  fs_group_wait_current: the host
 * waits until the fs_group_render_current issued `frames_back` calls ago (0 = the latest, 1 = the one before) has filled
 * the caller's buffers -- the pipelined loop is  render k; render_current k (host buffer k % 2); wait_current(1).
 * The caller owns the host buffers: one per frame in flight.  (After fs_group_render_current member 0's own iteration
 * buffer is the OTHER set's slot: per-renderer colour output of a frame is taken before that call.) */
uint32_t fs_group_wait_current(fs_group *g, uint32_t frames_back);
/* The tiler's plan as a pure host function (tests; equals fractalshark_amd/tiling.py): rows rank `rank` owns, the
 * common padded slice height, and frame_index[y] = row of the gathered buffer (N slices back to back) that holds frame
 * row y.  Any output pointer may be NULL. */
void fs_group_plan(uint32_t height, uint32_t world, uint32_t band, uint32_t rank, uint32_t *local_rows,
                   uint32_t *max_local_rows, uint32_t *frame_index);

#ifdef __cplusplus
}
#endif

#endif /* FSMI355_H */
