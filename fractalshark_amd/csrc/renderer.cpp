// renderer.cpp -- host side of libfsmi355.so: the C ABI of include/fsmi355.h.
//
// State machine of one reference GPURenderer (FractalSharkGpuLib/GPU_Render.cu:92-1823) re-expressed for
// HIP: two non-blocking streams (compute = lowest priority, display = highest, GPU_Render.cu:247-267),
// device buffers owned here, uploads cached by generation number (GPU_Render.cu:440-487), kernels launched
// asynchronously on the compute stream.  There is NO CPU fallback: if no HIP device is usable every entry
// point returns the HIP error.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/fsmi355_internal.h"
#include "kernels.h"
#include "la_math.hpp"

#define FS_TRY(expr)                                                                                                  \
    do {                                                                                                              \
        hipError_t e_ = (expr);                                                                                       \
        if (e_ != hipSuccess)                                                                                         \
            return (uint32_t)e_;                                                                                      \
    } while (0)

struct fs_renderer {
    int device = 0;
    hipStream_t compute = nullptr;
    hipStream_t display = nullptr;
    // HIP events around the iteration-kernel launches, a ring of pairs: fs_last_kernel_ms reads the newest one,
    // fs_kernel_ms_history the last few (frames that are in flight together, e.g. a pipelined bench loop, each keep theirs)
    static constexpr uint32_t kTimingRing = 64;
    hipEvent_t ev_start[kTimingRing] = {}, ev_stop[kTimingRing] = {};
    // a frame made of two kernels (HDRFloat<double> LAv2: the AT pass, then the frame's kernel) also records where the first one
    // ended (fs_kernel_ms_split_history); created on first use
    hipEvent_t ev_mid[kTimingRing] = {};
    bool mid_valid[kTimingRing] = {};
    uint64_t timed_launches = 0; // launches recorded so far; launch i uses pair i % kTimingRing

    // geometry
    uint32_t width = 0, height = 0, aa = 0, iter_bytes = 0;
    uint32_t w_block = 0, h_block = 0;
    uint32_t color_w = 0, color_h = 0;
    size_t n_cu = 0, n_color_cu = 0;
    uint32_t band_first = 0, band_rows = 0, band_stride = 0; // 0 rows = whole frame
    uint32_t local_rows = 0, local_rows_padded = 0;

    // buffers
    void *iters_internal = nullptr;
    size_t iters_internal_bytes = 0;
    void *iters_external = nullptr;
    size_t iters_external_bytes = 0;
    fs_reduction reduce_seed{}; // source of the stream-ordered seed copy in fs_render_current (must outlive the call)
    fs_color16 *colors = nullptr;
    fs_reduction *reduction = nullptr;
    uint64_t *stats = nullptr;
    size_t stats_words = 40;

    uint32_t *queue = nullptr; // pixel counter of the persistent launches (kernels_perturb.hip, k_perturb_scalar)
    uint32_t *tile_probe = nullptr, *tile_order = nullptr; // "long tiles first" (fs_render_bla): probe counts, launch order
    size_t tile_probe_cap = 0, tile_order_cap = 0;           // in elements
    // "longest tiles first" of the tuned LAv2 kernel (fs_render_lav2): the costs the last frame recorded per 8 x 8 tile, the
    // launch order made from them, work memory of the sort; and what the costs belong to (a frame of another geometry, band
    // layout or orbit generation starts cold: natural order, costs recorded)
    uint32_t *lav2_cost = nullptr, *lav2_order = nullptr, *lav2_sort_tmp = nullptr;
    size_t lav2_cost_cap = 0, lav2_order_cap = 0; // in elements
    bool lav2_cost_valid = false;
    struct CostKey {
        uint32_t width, local_rows, band_first, band_rows, band_stride;
        uint64_t orbit_gen;
        bool operator==(const CostKey &o) const
        {
            return width == o.width && local_rows == o.local_rows && band_first == o.band_first && band_rows == o.band_rows &&
                   band_stride == o.band_stride && orbit_gen == o.orbit_gen;
        }
    } lav2_cost_key{};
    // fs_render_bla's probe order is a pure function of (geometry, bands, orbit, coordinates, iteration limit): the next frame
    // with the same inputs reuses it and skips the probe launch (round 4; ~9 ms of C2's frame)
    bool po_order_valid = false;
    CostKey po_order_key{};
    uint64_t po_order_epoch = 0, po_order_iterations = 0;
    unsigned char po_order_coords[32] = {};
    bool last_frame_ordered = false; // the last fs_render_lav2 launch used a recorded order (fs_last_frame_tile_ordered)
    // "pixels in the order of the previous frame's counts" (kernels_order.hip; HDRFloat<double> and HDRFloat<CudaDblflt> LAv2):
    // the order, the sort's work memory, and what the order was made from
    // HDRFloat<double> LAv2: PerformAT in a pass of its own (fsk_at_pass64) with its own pixel order -- its results, the AT
    // iterations every pixel needs by itself (recorded by the first frame of a view), and the order made from them
    FsAtRes *at_res = nullptr;
    uint32_t *at_cost = nullptr, *at_order = nullptr;
    size_t at_cap = 0;
    bool at_order_valid = false; // ... for at_key (set where at_order is built: the order is a permutation of THAT key's buffer)
    uint32_t *pix_cost = nullptr; // per-pixel cost the unordered frame of a view records; what the order is sorted by
    size_t pix_cost_cap = 0;
    uint32_t *pix_order = nullptr, *pix_work = nullptr;
    void *pix_temp = nullptr;
    size_t pix_cap = 0, pix_temp_bytes = 0;
    bool pix_valid = false;
    bool pix_seen = false; // the last unordered frame's key (pix_seen_key): an order is only made for a view that comes twice
    struct PixKey {
        uint32_t rounded_width, local_rows, band_first, band_rows, band_stride;
        int type_tag, mode, parity;
        uint64_t orbit_gen, orbit_epoch, n_iterations;
        unsigned char coords[64];
        bool operator==(const PixKey &o) const { return memcmp(this, &o, sizeof(*this)) == 0; }
    } pix_key{}, pix_seen_key{}, at_key{};
    // (round 6) an order for a view's FIRST frame: tiles by a sampled PerformAT count (kernels_tile_sample.hip)
    uint32_t *cold_cost = nullptr, *cold_order = nullptr, *cold_work = nullptr;
    void *cold_temp = nullptr;
    size_t cold_temp_bytes = 0;
    uint32_t cold_cap = 0;
    bool last_cold_ordered = false; // (fs_last_frame_sampled_tile_order)
    bool lav2_last_ordered = false; // the last launch was an HDRFloat<float> frame in its recorded TILE order (fs_read_tile_order)
    bool last_launch_wide = false;   // the last render launched a 64-bit counting kernel: those carry no step counters
    bool stats_on = false;
    int variant = FS_VARIANT_TUNED;

    // palette (GPU_Render.cu:270-304)
    fs_color16 *pal = nullptr;
    uint32_t pal_iters = 0, pal_aux_depth = 0;
    const fs_color16 *pal_cached_host = nullptr;
    uint64_t pal_cached_gen = 0;

    // orbit (HDRFloat<float>)
    uint64_t orbit_gen = 0;
    // counts orbit uploads whose content differs from the one before (a generation of 0 means "not cached": it does not
    // identify an orbit, and RenderPerturbBLA re-uploads the same orbit on every call as the reference does -- a sampled
    // fingerprint of the entries tells a repeated upload from a new orbit; it only decides whether a recorded tile order
    // is reused, never a pixel)
    uint64_t orbit_epoch = 0, orbit_fp = 0, pending_fp = 0;
    bool orbit_ok = false;
    int orbit_type = -1; // FS_T_HDR32 / FS_T_HDR64 / FS_T_HDR2X32 / FS_T_F64
    fs_orbit_2x32 *orbit_2x32 = nullptr; // HDRFloat<CudaDblflt> orbit (FS_T_HDR2X32), used as uploaded
    int scaled_type = -1;
    void *scaled_t = nullptr; // PerturbExtras::Bad orbits of the scaled kernel (fs_orbit_hdr32_bad[] or fs_orbit_f64_bad[])
    fs_orbit_f32_bad *scaled_f = nullptr;
    uint64_t scaled_count = 0;
    float4 *zref = nullptr;
    float4 *zq = nullptr; // companions of zref for the tuned LAv2 loop (2 x zq_n entries)
    uint64_t zq_n = 0;
    float2 *zs2 = nullptr; // (inside the zq block) compact companions of the 16-step body
    float4 *zqb = nullptr;
    FsZ64 *zref64 = nullptr;
    fs_orbit_f64 *orbit_f64 = nullptr; // plain double orbit (FS_T_F64), used as uploaded
    void *orbit_plain = nullptr;       // plain float / CudaDblflt orbit (FS_T_F32 / FS_T_2X32), used as uploaded
    alignas(8) uint8_t at_plain[sizeof(fs_at_f64_u32)] = {0}; // ATInfo of the plain LA table (type = la_type)
    uint64_t orbit_size = 0, orbit_uncompressed = 0, orbit_period = 0;
    // PerturbExtras::SimpleCompression orbits: 0 = expanded once on upload (default), 1 = kept compressed, decompressed by
    // the kernel as it walks the orbit (fs_set_compressed_orbit_mode)
    int compressed_mode = 0;
    bool orbit_seq = false; // the resident orbit is a compressed one (wp_raw); zref / zref64 are NULL
    void *wp_raw = nullptr; // fs_orbit_hdr32_rc[] / fs_orbit_hdr64_rc[]
    fs_real_hdr32 c_low32[2] = {};
    fs_real_hdr64 c_low64[2] = {};
    alignas(8) uint8_t c_low_plain[2][16] = {}; // ... of a float / double / CudaDblflt / HDRFloat<CudaDblflt> orbit (as uploaded)

    // LA table
    uint64_t la_gen = 0;
    bool la_ok = false;
    int la_type = -1;
    void *las = nullptr; // fs_la_hdr32_u32[] or fs_la_hdr64_u32[]
    fs_la_stage_u32 *stages = nullptr;
    uint32_t n_las = 0, n_stages = 0;
    int la_valid = 0, use_at = 0;
    bool la_u64 = false;     // `las` holds the reference's uint64_t records (only the waypoint-resident wide kernel reads them)
    uint32_t at_step_hi = 0; // high word of the AT step length of a uint64_t table
    fs_at_hdr32_u32 at{};
    fs_at_hdr64_u32 at64{};
    fs_at_2x32_u32 at2x32{};

    // BLA table
    std::vector<void *> bla_level_mem;
    std::vector<uint64_t> bla_level_sizes;
    const void **bla_levels_dev = nullptr;
    int bla_type = -1;
    int32_t bla_n_levels = 0, bla_lm2 = 0;

    // direct kernels
    void *cx_row = nullptr; // double[] / hreal<float>[] / hreal<double>[] (16 B per column is enough for all)
    uint32_t cx_row_cap = 0;

    // memory management (r_alloc / r_free below)
    std::vector<void *> host_allocs; // input tables that live in page-locked HOST memory (device out of memory)
    // device blocks of this renderer (synchronous allocation): every live block with its size, and the released ones that
    // are kept for the next request of a similar size (r_alloc / r_free)
    struct Block {
        void *p;
        size_t bytes;
    };
    std::vector<Block> live_blocks, kept_blocks;
    std::mutex kept_mu; // kept_blocks only: another renderer of the same device may drain them when IT runs out of memory
    size_t host_alloc_bytes = 0;
    bool inject_input_oom = false;   // fault injection: FSMI355_FAIL_INPUT_ALLOC=1 at fs_create time
    void *arena = nullptr;           // work memory of fs_build_la (kept between calls, grown on demand)
    uint32_t *la_mail = nullptr;     // 32 words of coherent page-locked memory the build's kernels report through (k_la_mail)
    uint32_t la_mail_seq = 0;
    size_t arena_cap = 0;
    void *bla_block = nullptr;       // ONE allocation for the BLA table: the level pointer table, then the levels
    size_t bla_block_cap = 0;
    size_t las_cap = 0, stages_cap = 0; // bytes behind `las` / `stages` (reused by the next table when they fit)
    // device-native form of an HDRFloat<float> BLA table (FsBlaRec + ladder, kernels.h): [flag word | records | ladder]
    void *bla_native = nullptr;
    size_t bla_native_cap = 0;
    bool bla_native_ok = false;
    bool bla_native_stale = false; // table or orbit changed since the native form was made: remade by the next BLA render
    uint32_t bla_native_total = 0;
    // the heap-numbered copy the hand-written kernel reads (kernels_bla_fast.hip), made with the native form
    void *bla_heap = nullptr;
    size_t bla_heap_cap = 0;
    bool bla_heap_ok = false;
    uint64_t bla_heap_positions = 0;
    uint32_t bla_heap_nq = 0;
    uint32_t bla_level_off[kBlaMaxLevels] = {0};

    void *iters() const { return iters_external ? iters_external : iters_internal; }
    bool memory_initialized() const { return iters() != nullptr && width != 0; }
};

namespace {

uint32_t use_device(const fs_renderer *r)
{
    FS_TRY(hipSetDevice(r->device));
    return 0;
}

// ---- Device memory of a renderer.
// hipMalloc / hipFree behind a synchronisation of the compute stream (everything that touches such memory is enqueued on
// the compute stream or behind a synchronisation of it); optionally stream-ordered (hipMallocAsync / hipFreeAsync, as the
// reference does, GPU_Render.cu:127,142-153,362-395) -- see async_alloc_enabled() for why that is not the default.
// kInput allocations -- reference orbit, LA table, BLA table, their upload staging -- fall back to page-locked HOST memory
// when the device allocation fails, and the kernels then read them over the bus: slow, but the frame still renders
// (GPUPerturbSingleResults, Perturb.cuh:51-61; GPU_LAReference, GPU_LAReference.h:93-113).  Frame buffers (kFrame) do not.
enum AllocKind { kFrame = 0, kInput = 1 };

// The stream-ordered allocator (round 3: hipMallocAsync / hipFreeAsync on the compute stream) is OFF by default: on this
// ROCm (7.2.0) a block that the pool hands out again is not reliably the memory the next copy and the next kernel agree on
// -- tools/microbench/async_alloc_probe.hip (upload, transform, check, free, four rounds) finds stale or zero data from
// the second round on with blocks of 32 MiB and more, whatever the release threshold, the stream type or the kind of host
// memory, and none with hipMalloc / hipFree (profiles/r03zz_async_alloc_probe.txt).  In this library it showed as wrong
// frames from the SECOND orbit upload of a renderer when the orbit has tens of millions of entries (Views 10, 15, 22).
// FSMI355_ASYNC_ALLOC=1 switches the stream-ordered path back on.  What keeps allocation off the frame path either way:
// buffers are kept and reused when the next table fits (LA table, BLA table block, work arena, iteration buffers).
static bool async_alloc_enabled()
{
    static const bool on = [] {
        const char *e = getenv("FSMI355_ASYNC_ALLOC");
        return e != nullptr && atoi(e) != 0;
    }();
    return on;
}

constexpr size_t kKeptBlocks = 16;
constexpr size_t kKeptBytes = (size_t)2 << 30;

static uint64_t release_kept_blocks(fs_renderer *r)
{
    std::lock_guard<std::mutex> g(r->kept_mu);
    uint64_t bytes = 0;
    for (const auto &k : r->kept_blocks) {
        (void)hipFree(k.p);
        bytes += k.bytes;
    }
    r->kept_blocks.clear();
    return bytes;
}

// Every renderer of the process, so that one that runs out of device memory can take back what the OTHERS of its device keep
// idle (FractalShark holds four GPURenderers on one device; a kept block is idle by construction -- its owner parked it
// after draining its stream -- so any thread may free it).
static std::mutex g_renderers_mu;
static std::vector<fs_renderer *> g_renderers;

static uint64_t release_idle_memory_of_device(int device)
{
    std::lock_guard<std::mutex> g(g_renderers_mu);
    uint64_t bytes = 0;
    for (fs_renderer *o : g_renderers)
        if (o->device == device)
            bytes += release_kept_blocks(o);
    return bytes;
}

hipError_t r_alloc(fs_renderer *r, void **out, size_t bytes, AllocKind kind)
{
    if (bytes == 0)
        bytes = 16;
    *out = nullptr;
    hipError_t e = hipErrorOutOfMemory;
    if (!(kind == kInput && r->inject_input_oom)) {
        if (r->compute && async_alloc_enabled()) {
            e = hipMallocAsync(out, bytes, r->compute);
        } else {
            // a kept block that fits (best fit, at most twice the size asked for) before a new allocation: a host that uploads
            // an orbit and its tables for every frame allocates nothing in the steady state
            {
                std::lock_guard<std::mutex> g(r->kept_mu);
                size_t best = r->kept_blocks.size();
                for (size_t i = 0; i < r->kept_blocks.size(); i++) {
                    const size_t b = r->kept_blocks[i].bytes;
                    if (b >= bytes && b <= 2 * bytes + (1u << 16) &&
                        (best == r->kept_blocks.size() || b < r->kept_blocks[best].bytes))
                        best = i;
                }
                if (best != r->kept_blocks.size()) {
                    *out = r->kept_blocks[best].p;
                    r->live_blocks.push_back(r->kept_blocks[best]);
                    r->kept_blocks.erase(r->kept_blocks.begin() + (long)best);
                    return hipSuccess;
                }
            }
            e = hipMalloc(out, bytes);
            if (e != hipSuccess) { // idle blocks may be what is in the way: this renderer's, then every renderer's of the device
                (void)hipGetLastError();
                if (release_kept_blocks(r) != 0u)
                    e = hipMalloc(out, bytes);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    if (release_idle_memory_of_device(r->device) != 0u)
                        e = hipMalloc(out, bytes);
                }
            }
            if (e == hipSuccess)
                r->live_blocks.push_back(fs_renderer::Block{*out, bytes});
        }
    }
    if (e == hipSuccess || kind != kInput)
        return e;
    (void)hipGetLastError(); // the failed device allocation is handled here, not reported by a later launch check
    e = hipHostMalloc(out, bytes, hipHostMallocDefault); // mapped into the device's address space at the same address
    if (e == hipSuccess) {
        r->host_allocs.push_back(*out);
        r->host_alloc_bytes += bytes;
    }
    return e;
}
template <class T> hipError_t r_alloc(fs_renderer *r, T **out, size_t bytes, AllocKind kind)
{
    return r_alloc(r, (void **)out, bytes, kind);
}

hipError_t r_free(fs_renderer *r, const void *cp)
{
    void *p = const_cast<void *>(cp);
    if (!p)
        return hipSuccess;
    for (size_t i = 0; i < r->host_allocs.size(); i++)
        if (r->host_allocs[i] == p) {
            r->host_allocs.erase(r->host_allocs.begin() + (long)i);
            if (r->compute)
                (void)hipStreamSynchronize(r->compute); // a kernel may still be reading it
            return hipHostFree(p);
        }
    if (r->compute && async_alloc_enabled())
        return hipFreeAsync(p, r->compute);
    if (r->compute)
        (void)hipStreamSynchronize(r->compute); // work that uses the block has been enqueued on this stream only
    for (size_t i = 0; i < r->live_blocks.size(); i++)
        if (r->live_blocks[i].p == p) {
            const fs_renderer::Block b = r->live_blocks[i];
            r->live_blocks.erase(r->live_blocks.begin() + (long)i);
            // kept for the next request -- up to kKeptBlocks of them and kKeptBytes in total (the oldest go first)
            std::lock_guard<std::mutex> g(r->kept_mu);
            r->kept_blocks.push_back(b);
            size_t total = 0;
            for (const auto &k : r->kept_blocks)
                total += k.bytes;
            while (!r->kept_blocks.empty() && (r->kept_blocks.size() > kKeptBlocks || total > kKeptBytes)) {
                total -= r->kept_blocks.front().bytes;
                (void)hipFree(r->kept_blocks.front().p);
                r->kept_blocks.erase(r->kept_blocks.begin());
            }
            return hipSuccess;
        }
    return hipFree(p); // (not one of ours: allocated before the compute stream existed, or by the stream-ordered path)
}

// The installed LA table (records + stages): the buffers of the previous table are kept when the new one fits.
hipError_t la_reserve(fs_renderer *r, size_t las_bytes, size_t stages_bytes)
{
    if (!r->las || r->las_cap < las_bytes) {
        (void)r_free(r, r->las);
        r->las = nullptr;
        r->las_cap = 0;
        hipError_t e = r_alloc(r, &r->las, las_bytes, kInput);
        if (e != hipSuccess)
            return e;
        r->las_cap = las_bytes ? las_bytes : 16;
    }
    if (!r->stages || r->stages_cap < stages_bytes) {
        (void)r_free(r, r->stages);
        r->stages = nullptr;
        r->stages_cap = 0;
        hipError_t e = r_alloc(r, (void **)&r->stages, stages_bytes, kInput);
        if (e != hipSuccess)
            return e;
        r->stages_cap = stages_bytes ? stages_bytes : 16;
    }
    return hipSuccess;
}

// Work memory that outlives a call: grown, never shrunk, handed out from the start on every use.
hipError_t arena_reserve(fs_renderer *r, size_t bytes)
{
    if (r->arena_cap >= bytes)
        return hipSuccess;
    (void)r_free(r, r->arena);
    r->arena = nullptr;
    r->arena_cap = 0;
    const size_t want = bytes + bytes / 4; // some slack: the next orbit of a zoom sequence is usually a little longer
    hipError_t e = r_alloc(r, &r->arena, want, kInput);
    if (e == hipSuccess)
        r->arena_cap = want;
    return e;
}

void compute_local_rows(fs_renderer *r)
{
    if (r->band_rows == 0 || r->band_rows >= r->height) {
        r->band_first = 0;
        r->band_rows = r->height;
        r->band_stride = r->height ? r->height : 1;
    }
    // rows owned = sum over k of |[first + k*stride, +rows) intersect [0,height)|
    uint64_t rows = 0;
    for (uint64_t start = r->band_first; start < r->height; start += r->band_stride) {
        const uint64_t end = start + r->band_rows < r->height ? start + r->band_rows : r->height;
        rows += end - start;
    }
    r->local_rows = (uint32_t)rows;
    r->local_rows_padded = (r->local_rows + 7u) / 8u * 8u;
}

FsFrame make_frame(const fs_renderer *r)
{
    FsFrame f;
    f.width = r->width;
    f.height = r->height;
    f.rounded_width = r->w_block * 16u;
    f.local_rows = r->local_rows;
    f.band_first = r->band_first;
    f.band_rows = r->band_rows;
    f.band_stride = r->band_stride;
    f.iter_u64 = r->iter_bytes == 8 ? 1u : 0u;
    f.wide = (r->variant & FS_VARIANT_FLAG_WIDE) != 0 ? 1u : 0u; // (|= cap >= 2^32 where the cap is known)
    return f;
}

// "long tiles first" (fs_render_bla, perturbation only): the probe runs each tile's centre pixel for n_iterations /
// kTileProbeDivisor steps; on by default for an iteration limit far above the bulk of a frame's pixels and enough tiles
// for an order to matter
constexpr uint64_t kTileProbeDivisor = 32;
constexpr uint64_t kTileOrderMinIterations = 1ull << 18;
constexpr uint32_t kTileOrderMinTiles = 4096;
// the same for the self-recorded order of the tuned LAv2 kernel: below this many tiles the chip is not full anyway
constexpr uint32_t kLav2OrderMinTiles = 2048;

// float4 units of the tuned loops' companion arrays of an n-entry HDRFloat<float> orbit (make_quiet_orbit lays them out;
// fs_orbit_device_bytes reports them)
constexpr uint64_t kQuietSlack = 32;
constexpr uint64_t quiet_orbit_units(uint64_t n)
{
    return 2 * (n + 2) + 16 + ((n + 2) + kQuietSlack + 1) / 2 + ((n + 2) + kQuietSlack);
}

// zq: the tuned LAv2 loop's view of the prepared orbit (same length incl. the two spare entries)
hipError_t make_quiet_orbit(fs_renderer *r, uint64_t n)
{
    if (r->zq) {
        (void)r_free(r, r->zq);
        r->zq = nullptr;
    }
    // two companions back to back; the scaled runs request their entries one 8-entry body ahead, so the second one may be
    // read up to 16 entries past its end (never used)
    // ... followed by the compact form the 16-step body of the untested loop reads: 2Z alone (8 B per entry) and, per entry, the
    // block bounds of the entries 3, 7, 11 and 15 further on (16 B); 32 entries of slack each (the body after the last is
    // requested ahead, never used)
    const uint64_t m = n + 2, slack = kQuietSlack;
    const uint64_t units = quiet_orbit_units(n);
    hipError_t err = r_alloc(r, (void **)&r->zq, units * sizeof(float4), kInput);
    if (err != hipSuccess)
        return err;
    r->zq_n = m;
    r->zs2 = (float2 *)(r->zq + 2 * m + 16);
    r->zqb = r->zq + 2 * m + 16 + (m + slack + 1) / 2;
    err = hipMemsetAsync(r->zs2, 0, ((m + slack + 1) / 2 + (m + slack)) * sizeof(float4), r->compute);
    if (err != hipSuccess)
        return err;
    fsk_make_quiet_orbit(r->zref, r->zq, r->zs2, r->zqb, m, r->compute);
    return hipGetLastError();
}

uint32_t ensure_iter_buffer(fs_renderer *r)
{
    // capacity is tracked in BYTES: the same frame needs twice the memory with IterType = uint64_t
    const size_t need = (size_t)r->w_block * 16u * r->local_rows_padded * r->iter_bytes;
    if (r->iters_external)
        return r->iters_external_bytes >= need ? 0 : (uint32_t)hipErrorInvalidValue; // never write past a caller's buffer
    if (r->iters_internal && r->iters_internal_bytes >= need)
        return 0;
    if (r->iters_internal) {
        if (r->display)
            FS_TRY(hipStreamSynchronize(r->display)); // a progressive RenderCurrent may still be reading it
        FS_TRY(r_free(r, r->iters_internal));
        r->iters_internal = nullptr;
        r->iters_internal_bytes = 0;
    }
    FS_TRY(r_alloc(r, &r->iters_internal, need, kFrame));
    r->iters_internal_bytes = need;
    if (r->compute)
        FS_TRY(hipStreamSynchronize(r->compute)); // usable from any stream from here on
    return 0;
}

// The BLA table lives in ONE allocation: 64 level pointers (the device-side pointer table the kernels index by level), then
// the levels back to back, each 256-byte aligned; kept and reused when the next table fits (the reference re-allocates
// and re-uploads every level on every BLA render, GPU_Render.cu:1464-1479).
constexpr size_t kBlaPtrTableBytes = 64 * sizeof(void *);

void bla_release(fs_renderer *r)
{
    (void)r_free(r, r->bla_native);
    r->bla_native = nullptr;
    r->bla_native_cap = 0;
    r->bla_native_ok = false;
    (void)r_free(r, r->bla_heap);
    r->bla_heap = nullptr;
    r->bla_heap_cap = 0;
    r->bla_heap_ok = false;
    (void)r_free(r, r->bla_block);
    r->bla_block = nullptr;
    r->bla_block_cap = 0;
    r->bla_level_mem.clear();
    r->bla_level_sizes.clear();
    r->bla_levels_dev = nullptr;
    r->bla_n_levels = 0;
}

// Lays out n_levels levels of sizes[l] records of rec_bytes in the block (growing it if needed) and uploads the pointer
// table on the compute stream.  A level of size 0 gets a NULL pointer.
hipError_t bla_layout(fs_renderer *r, const uint64_t *sizes, int32_t n_levels, size_t rec_bytes)
{
    if (n_levels > 64)
        return hipErrorInvalidValue;
    size_t total = kBlaPtrTableBytes;
    for (int32_t l = 0; l < n_levels; l++)
        total += (sizes[l] * rec_bytes + 255u) & ~(size_t)255u;
    r->bla_n_levels = 0;
    if (!r->bla_block || r->bla_block_cap < total) {
        bla_release(r);
        hipError_t e = r_alloc(r, &r->bla_block, total, kInput);
        if (e != hipSuccess)
            return e;
        r->bla_block_cap = total;
    }
    r->bla_level_mem.assign((size_t)n_levels, nullptr);
    r->bla_level_sizes.assign((size_t)n_levels, 0);
    size_t at = kBlaPtrTableBytes;
    for (int32_t l = 0; l < n_levels; l++) {
        if (sizes[l] == 0)
            continue;
        r->bla_level_mem[(size_t)l] = (char *)r->bla_block + at;
        r->bla_level_sizes[(size_t)l] = sizes[l];
        at += (sizes[l] * rec_bytes + 255u) & ~(size_t)255u;
    }
    r->bla_levels_dev = (const void **)r->bla_block;
    return hipMemcpyAsync(r->bla_block, r->bla_level_mem.data(), sizeof(void *) * (size_t)n_levels, hipMemcpyHostToDevice,
                          r->compute);
}

// Device-native form of the HDRFloat<float> table just installed in the block (see FsBlaRec, kernels.h).  Leaves
// bla_native_ok = false -- the kernels then read the reference-layout records -- when the table has more than
// kBlaMaxLevels levels or 2^32 records, when memory for it cannot be had, or when an r2 is not a reduced non-negative finite
// value (the integer-key compare would then differ from the reference's float compare).  Synchronises the compute stream.
uint32_t bla_make_native(fs_renderer *r, int32_t n_levels)
{
    r->bla_native_ok = false;
    r->bla_native_stale = false;
    if (n_levels <= 2 || n_levels > kBlaMaxLevels || r->bla_type != FS_T_HDR32 || !r->orbit_ok ||
        r->orbit_type != FS_T_HDR32 || !r->zref)
        return 0;
    uint64_t total = 0;
    for (int32_t l = 2; l < n_levels; l++) {
        r->bla_level_off[l] = (uint32_t)total;
        total += r->bla_level_sizes[(size_t)l];
    }
    if (total == 0 || total > 0xFFFFFFF0ull)
        return 0;
    // (+ the lookup's pre-test keys, one per orbit index 4 q + 1)
    const uint32_t n_kmax = (uint32_t)(r->orbit_uncompressed / 4u) + 2u;
    const size_t need = 256 + (size_t)total * (sizeof(FsBlaRec) + 2 * sizeof(int4)) + (size_t)n_kmax * sizeof(long long);
    if (!r->bla_native || r->bla_native_cap < need) {
        (void)r_free(r, r->bla_native);
        r->bla_native = nullptr;
        r->bla_native_cap = 0;
        if (r_alloc(r, &r->bla_native, need, kInput) != hipSuccess) {
            (void)hipGetLastError();
            return 0; // not an error: the reference-layout table serves
        }
        r->bla_native_cap = need;
    }
    uint32_t *bad = (uint32_t *)r->bla_native;
    FsBlaRec *rec = (FsBlaRec *)((char *)r->bla_native + 256);
    int4 *lad = (int4 *)((char *)rec + (size_t)total * sizeof(FsBlaRec));
    FS_TRY(hipMemsetAsync(bad, 0, 256, r->compute));
    fsk_bla_make_native((const fs_bla_hdr32 *const *)r->bla_levels_dev, r->bla_level_off, r->bla_level_sizes.data(), n_levels,
                        r->zref, (uint32_t)r->orbit_uncompressed, rec, lad, bad, r->bla_lm2,
                        (long long *)(lad + 2 * (size_t)total), n_kmax, r->compute);
    FS_TRY(hipGetLastError());
    uint32_t flag = 1;
    FS_TRY(hipMemcpyAsync(&flag, bad, 4, hipMemcpyDeviceToHost, r->compute));
    FS_TRY(hipStreamSynchronize(r->compute));
    r->bla_native_total = (uint32_t)total;
    r->bla_native_ok = flag == 0;
    // ... and its heap-numbered copy for the hand-written kernel (not an error when it cannot be had: the compiled kernel serves)
    r->bla_heap_ok = false;
    const uint64_t hn = fsk_bla_heap_positions(r->bla_level_sizes.data(), n_levels);
    // (orbit positions below 2^24: the kernel forms the address of Q[(m - 1) / 4] with one 24-bit multiply-add)
    if (r->bla_native_ok && hn != 0 && r->orbit_uncompressed < 0x00FFFFF0ull) {
        const size_t nz = (size_t)r->orbit_uncompressed + 2u;
        const size_t hneed = (size_t)hn * (sizeof(FsBlaRec) + 2 * sizeof(int4)) + (size_t)n_kmax * 3 * sizeof(int4) + nz * sizeof(float4);
        if (!r->bla_heap || r->bla_heap_cap < hneed) {
            (void)r_free(r, r->bla_heap);
            r->bla_heap = nullptr;
            r->bla_heap_cap = 0;
            if (r_alloc(r, &r->bla_heap, hneed, kInput) != hipSuccess) {
                (void)hipGetLastError();
                return 0;
            }
            r->bla_heap_cap = hneed;
        }
        FS_TRY(hipMemsetAsync(r->bla_heap, 0, hneed, r->compute));
        FsBlaRec *hrec = (FsBlaRec *)r->bla_heap;
        int4 *hlad = (int4 *)(hrec + hn);
        int4 *hq = hlad + 2 * (size_t)hn;
        float4 *zb = (float4 *)(hq + 3 * (size_t)n_kmax);
        fsk_bla_make_heap(rec, lad, (const long long *)(lad + 2 * (size_t)total), n_kmax, r->bla_level_off,
                          r->bla_level_sizes.data(), n_levels, r->bla_lm2, r->zref, (uint32_t)r->orbit_uncompressed, hrec, hlad, hq,
                          zb, r->compute);
        FS_TRY(hipGetLastError());
        FS_TRY(hipStreamSynchronize(r->compute));
        r->bla_heap_positions = hn;
        r->bla_heap_nq = n_kmax;
        r->bla_heap_ok = true;
    }
    return 0;
}

// A new orbit is in place: the native BLA table carries arrival entries of the previous one.
void orbit_changed(fs_renderer *r)
{
    r->bla_native_ok = false;
    r->bla_native_stale = r->bla_n_levels > 0 && r->bla_type == FS_T_HDR32;
}

// The compressed-resident form of the orbit (runtime decompression) goes whenever another orbit is about to come in.
void drop_seq(fs_renderer *r)
{
    (void)r_free(r, r->wp_raw);
    r->wp_raw = nullptr;
    r->orbit_seq = false;
}

void free_perturb(fs_renderer *r)
{
    if (r->zref)
        r_free(r, r->zref);
    if (r->zq)
        r_free(r, r->zq);
    r->zq = nullptr;
    if (r->zref64)
        r_free(r, r->zref64);
    drop_seq(r);
    if (r->orbit_f64)
        r_free(r, r->orbit_f64);
    if (r->orbit_plain)
        r_free(r, r->orbit_plain);
    r->orbit_plain = nullptr;
    if (r->orbit_2x32)
        r_free(r, r->orbit_2x32);
    r->orbit_2x32 = nullptr;
    if (r->scaled_t)
        r_free(r, r->scaled_t);
    if (r->scaled_f)
        r_free(r, r->scaled_f);
    r->scaled_t = nullptr;
    r->scaled_f = nullptr;
    r->scaled_count = 0;
    r->zref = nullptr;
    r->zref64 = nullptr;
    r->orbit_f64 = nullptr;
    r->orbit_ok = false;
    drop_seq(r);
    r->orbit_gen = 0;
    if (r->las)
        r_free(r, r->las);
    if (r->stages)
        r_free(r, r->stages);
    r->las = nullptr;
    r->stages = nullptr;
    r->las_cap = r->stages_cap = 0;
    r->la_ok = false;
    r->la_gen = 0;
    bla_release(r);
}

void free_all(fs_renderer *r)
{
    free_perturb(r);
    if (r->iters_internal)
        r_free(r, r->iters_internal);
    if (r->colors)
        r_free(r, r->colors);
    if (r->reduction)
        r_free(r, r->reduction);
    if (r->stats)
        r_free(r, r->stats);
    if (r->queue)
        r_free(r, r->queue);
    r->queue = nullptr;
    if (r->tile_probe)
        r_free(r, r->tile_probe);
    if (r->tile_order)
        r_free(r, r->tile_order);
    r->tile_probe = r->tile_order = nullptr;
    r->tile_probe_cap = r->tile_order_cap = 0;
    (void)r_free(r, r->lav2_cost);
    (void)r_free(r, r->lav2_order);
    (void)r_free(r, r->lav2_sort_tmp);
    (void)r_free(r, r->pix_cost);
    r->pix_cost = nullptr;
    r->pix_cost_cap = 0;
    (void)r_free(r, r->at_res);
    (void)r_free(r, r->at_cost);
    (void)r_free(r, r->at_order);
    r->at_res = nullptr;
    r->at_cost = r->at_order = nullptr;
    r->at_cap = 0;
    r->at_order_valid = false;
    (void)r_free(r, r->pix_order);
    (void)r_free(r, r->pix_work);
    (void)r_free(r, r->pix_temp);
    r->pix_order = r->pix_work = nullptr;
    r->pix_temp = nullptr;
    r->pix_cap = 0;
    r->pix_valid = false;
    (void)r_free(r, r->cold_cost);
    (void)r_free(r, r->cold_order);
    (void)r_free(r, r->cold_work);
    (void)r_free(r, r->cold_temp);
    r->cold_cost = r->cold_order = r->cold_work = nullptr;
    r->cold_temp = nullptr;
    r->cold_cap = 0;
    r->lav2_cost = r->lav2_order = r->lav2_sort_tmp = nullptr;
    r->lav2_cost_cap = r->lav2_order_cap = 0;
    r->lav2_cost_valid = false;
    r->po_order_valid = false;

    if (r->pal)
        r_free(r, r->pal);
    if (r->cx_row)
        r_free(r, r->cx_row);
    (void)r_free(r, r->arena);
    r->arena = nullptr;
    if (r->la_mail)
        (void)hipHostFree(r->la_mail);
    r->la_mail = nullptr;
    r->arena_cap = 0;
    r->iters_internal = nullptr;
    r->colors = nullptr;
    r->reduction = nullptr;
    r->stats = nullptr;
    r->pal = nullptr;
    r->cx_row = nullptr;
    r->width = r->height = 0;
    release_kept_blocks(r);
}

struct TimedLaunch {
    fs_renderer *r;
    explicit TimedLaunch(fs_renderer *rr) : r(rr)
    {
        if (r->ev_start[0]) {
            hipEventRecord(r->ev_start[r->timed_launches % fs_renderer::kTimingRing], r->compute);
            r->mid_valid[r->timed_launches % fs_renderer::kTimingRing] = false;
        }
        if (r->stats_on && r->stats)
            hipMemsetAsync(r->stats, 0, (r->stats_words == 40 ? 40 : 8) * sizeof(uint64_t), r->compute);
    }
    void mid() // between the two kernels of a two-kernel frame
    {
        if (!r->ev_start[0])
            return;
        const uint32_t i = (uint32_t)(r->timed_launches % fs_renderer::kTimingRing);
        if (!r->ev_mid[i] && hipEventCreate(&r->ev_mid[i]) != hipSuccess) {
            (void)hipGetLastError();
            r->ev_mid[i] = nullptr;
            return;
        }
        if (hipEventRecord(r->ev_mid[i], r->compute) == hipSuccess)
            r->mid_valid[i] = true;
    }
    ~TimedLaunch()
    {
        if (r->ev_stop[0]) {
            hipEventRecord(r->ev_stop[r->timed_launches % fs_renderer::kTimingRing], r->compute);
            r->timed_launches++;
        }
    }
};

} // namespace

static void fill_coords(FsCoordsT<float> &c, const void *coords)
{
    const fs_real_hdr32 *p = (const fs_real_hdr32 *)coords;
    c.dx = fs::hreal32{p[0].m, p[0].e};
    c.dy = fs::hreal32{p[1].m, p[1].e};
    c.centerX = fs::hreal32{p[2].m, p[2].e};
    c.centerY = fs::hreal32{p[3].m, p[3].e};
}
static void fill_coords(FsCoordsT<double> &c, const void *coords)
{
    const fs_real_hdr64 *p = (const fs_real_hdr64 *)coords;
    c.dx = fs::hreal64{p[0].m, p[0].e};
    c.dy = fs::hreal64{p[1].m, p[1].e};
    c.centerX = fs::hreal64{p[2].m, p[2].e};
    c.centerY = fs::hreal64{p[3].m, p[3].e};
}

template <class F> static void fill_lav2(fs_renderer *r, FsLav2ArgsT<F> &A, const void *coords, uint64_t n_iterations, int parity)
{
    memset(&A, 0, sizeof(A));
    A.out = (uint32_t *)r->iters();
    A.las = (const typename FsDev<F>::LA *)r->las;
    A.stages = r->stages;
    A.stats = r->stats;
    A.frame = make_frame(r);
    fill_coords(A.coords, coords);
    A.orbit_count = (uint32_t)r->orbit_uncompressed;
    A.period = (uint32_t)r->orbit_period;
    A.stage_count = r->n_stages;
    A.n_iterations = (uint32_t)n_iterations;
    A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
    A.la_valid = r->la_ok ? r->la_valid : 0;
    A.use_at = r->use_at;
    A.parity = (parity == FS_PARITY_CPU_GPUSTAGE) ? FS_PARITY_GPUSTAGE : FS_PARITY_LITERAL;
    A.orbit_count_hi = (uint32_t)(r->orbit_uncompressed >> 32);
    A.period_hi = (uint32_t)(r->orbit_period >> 32);
    A.at_step_hi = r->at_step_hi;
    A.la_u64 = r->la_u64 ? 1u : 0u;
}


// uint64_t IterType tables (fs_la_*_u64 / fs_la_stage_u64 / fs_at_*_u64) are narrowed to the uint32_t device records.
template <class R64, class R32> static bool narrow_la(const void *in, uint32_t n, std::vector<uint8_t> &out)
{
    out.resize((size_t)n * sizeof(R32));
    const R64 *src = (const R64 *)in;
    R32 *dst = (R32 *)out.data();
    for (uint32_t i = 0; i < n; i++) {
        if (src[i].StepLength > 0xFFFFFFFFull || src[i].NextStageLAIndex > 0xFFFFFFFFull)
            return false;
        memcpy(&dst[i], &src[i], offsetof(R32, StepLength)); // Ref .. MinMag are laid out identically
        dst[i].StepLength = (uint32_t)src[i].StepLength;
        dst[i].NextStageLAIndex = (uint32_t)src[i].NextStageLAIndex;
    }
    return true;
}


extern "C" {

fs_renderer *fs_create(int device)
{
    fs_renderer *r = new (std::nothrow) fs_renderer();
    if (r) {
        r->device = device;
        // fault injection for the out-of-memory path (tests): every input-table allocation of this renderer behaves as if
        // the device were full and lands in page-locked host memory
        const char *e = getenv("FSMI355_FAIL_INPUT_ALLOC");
        r->inject_input_oom = e != nullptr && atoi(e) != 0;
        std::lock_guard<std::mutex> g(g_renderers_mu);
        g_renderers.push_back(r);
    }
    return r;
}

void fs_destroy(fs_renderer *r)
{
    if (!r)
        return;
    {
        std::lock_guard<std::mutex> g(g_renderers_mu);
        g_renderers.erase(std::remove(g_renderers.begin(), g_renderers.end(), r), g_renderers.end());
    }
    if (hipSetDevice(r->device) == hipSuccess) {
        if (r->compute)
            hipStreamSynchronize(r->compute);
        if (r->display)
            hipStreamSynchronize(r->display);
        free_all(r);
        if (r->compute)
            hipStreamSynchronize(r->compute); // the stream-ordered frees have run before their stream goes away
        for (uint32_t i = 0; i < fs_renderer::kTimingRing; i++) {
            if (r->ev_start[i])
                hipEventDestroy(r->ev_start[i]);
            if (r->ev_stop[i])
                hipEventDestroy(r->ev_stop[i]);
            if (r->ev_mid[i])
                hipEventDestroy(r->ev_mid[i]);
        }
        if (r->compute)
            hipStreamDestroy(r->compute);
        if (r->display)
            hipStreamDestroy(r->display);
    }
    delete r;
}

uint32_t fs_test_device_is_working(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return 0;
    // probe the caller's current device and leave it current (the reference hard-codes device 0 because it only ever
    // uses that one, GPU_Render.cu:113; a multi-GPU host has already selected its own)
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess || hipSetDevice(cur) != hipSuccess)
        return 0;
    if (hipFree(nullptr) != hipSuccess)
        return 0;
    return 1;
}

int fs_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

const char *fs_error_string(uint32_t err)
{
    switch (err) {
        case FS_OK:
            return "no error";
        case FS_ERR_1:
        case FS_ERR_2:
            return "FractalSharkError: unused";
        case FS_ERR_3:
            return "FractalSharkError::Error3: antialiasing must be 1..4";
        case FS_ERR_4:
            return "FractalSharkError::Error4: width not divisible by antialiasing";
        case FS_ERR_5:
            return "FractalSharkError::Error5: height not divisible by antialiasing";
        case FS_ERR_6:
            return "FractalSharkError::Error6: no uploaded orbit/table for this type";
        case FS_ERR_7:
            return "FractalSharkError::Error7";
        case FS_ERR_UNSUPPORTED:
            return "fsmi355: numeric type / mode not built into this library";
        default:
            return hipGetErrorString((hipError_t)err);
    }
}

uint32_t fs_init_memory(fs_renderer *r, uint32_t w, uint32_t h, uint32_t antialiasing, uint32_t iter_bytes,
                        const fs_color16 *pal_interleaved, uint32_t pal_iters, uint32_t palette_aux_depth,
                        uint64_t palette_generation, int expected_reuse)
{
    if (uint32_t e = use_device(r))
        return e;
    if (iter_bytes != 4 && iter_bytes != 8)
        return FS_ERR_UNSUPPORTED;
    if (!r->compute) {
        int lo = 0, hi = 0;
        FS_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        FS_TRY(hipStreamCreateWithPriority(&r->compute, hipStreamNonBlocking, lo));
        FS_TRY(hipStreamCreateWithPriority(&r->display, hipStreamNonBlocking, hi));
        for (uint32_t i = 0; i < fs_renderer::kTimingRing; i++) {
            FS_TRY(hipEventCreate(&r->ev_start[i]));
            FS_TRY(hipEventCreate(&r->ev_stop[i]));
        }
        // the stream-ordered allocator keeps freed memory for the next allocation instead of returning it to the driver at
        // every synchronisation (uploads synchronise: their host buffers are borrowed for the call only)
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, r->device) == hipSuccess && pool) {
            uint64_t keep = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    }
    // palette: re-upload when the host pointer or the generation changes (GPU_Render.cu:270-304)
    r->pal_aux_depth = palette_aux_depth;
    if (pal_interleaved && (r->pal_cached_host != pal_interleaved || r->pal_cached_gen != palette_generation ||
                            r->pal_iters != pal_iters)) {
        if (r->pal) {
            // a progressive RenderCurrent on the display stream may still be reading the old palette, and r_free parks the
            // block where the r_alloc below finds it again: the display stream must have drained before the block is reused
            // (hipFree used to synchronise the whole device here)
            FS_TRY(hipStreamSynchronize(r->display));
            FS_TRY(r_free(r, r->pal));
            r->pal = nullptr;
        }
        FS_TRY(r_alloc(r, (void **)&r->pal, sizeof(fs_color16) * (size_t)pal_iters, kFrame));
        FS_TRY(hipMemcpyAsync(r->pal, pal_interleaved, sizeof(fs_color16) * (size_t)pal_iters, hipMemcpyDefault,
                              r->compute));
        FS_TRY(hipStreamSynchronize(r->compute)); // host buffer is borrowed for the call only
        r->pal_iters = pal_iters;
        r->pal_cached_host = pal_interleaved;
        r->pal_cached_gen = palette_generation;
    }
    if (r->width == w && r->height == h && r->aa == antialiasing && r->iter_bytes == iter_bytes && expected_reuse)
        return 0;
    if (antialiasing > 4 || antialiasing < 1)
        return FS_ERR_3;
    if (w % antialiasing != 0)
        return FS_ERR_4;
    if (h % antialiasing != 0)
        return FS_ERR_5;

    r->w_block = w / 16 + (w % 16 != 0);
    r->h_block = h / 8 + (h % 8 != 0);
    r->width = w;
    r->height = h;
    r->aa = antialiasing;
    r->iter_bytes = iter_bytes;
    r->n_cu = (size_t)r->w_block * 16 * r->h_block * 8;
    r->color_w = w / antialiasing;
    r->color_h = h / antialiasing;
    const uint32_t wcb = r->color_w / 16 + (r->color_w % 16 != 0);
    const uint32_t hcb = r->color_h / 8 + (r->color_h % 8 != 0);
    r->n_color_cu = (size_t)wcb * 16 * hcb * 8;
    r->band_rows = 0;
    compute_local_rows(r);
    // a caller-owned iteration buffer was sized for the previous geometry: drop it (fs_set_external_iter_buffer again)
    r->iters_external = nullptr;
    r->iters_external_bytes = 0;

    // ResetMemory(..., ResetPerturb::Yes, ...) -- GPU_Render.cu:346.  Frees are ordered on the compute stream; the display
    // stream (progressive RenderCurrent) may still be reading the buffers that are about to go
    FS_TRY(hipStreamSynchronize(r->display));
    free_perturb(r);
    if (uint32_t e = ensure_iter_buffer(r)) {
        free_all(r);
        return e;
    }
    if (r->colors) {
        r_free(r, r->colors);
        r->colors = nullptr;
    }
    if (!r->reduction)
        FS_TRY(r_alloc(r, (void **)&r->reduction, sizeof(fs_reduction), kFrame));
    if (!r->stats) {
        // 8 counters; a measurement build (FS_TRACE_WAVES) appends four words per wave of the largest frame it will see
        // (words 16..27: per-phase cycle counters of the FS_PROFILE_CYCLES build of the BLA kernel, tools/c5_phase_probe.py;
        // the two measurement builds are not combined)
        r->stats_words = 40; // (words 28..39: the probe build of k_lav2_hdr64's hand-written loops, tools/c4_arm_probe.py)
        if (const char *e = getenv("FSMI355_TRACE_WAVES"))
            r->stats_words = 16 + 4 * (size_t)atoll(e);
        FS_TRY(r_alloc(r, (void **)&r->stats, r->stats_words * sizeof(uint64_t), kFrame));
        FS_TRY(hipMemsetAsync(r->stats, 0, r->stats_words * sizeof(uint64_t), r->compute));
    }
    if (!r->queue)
        FS_TRY(r_alloc(r, (void **)&r->queue, 64, kFrame));
    FS_TRY(r_alloc(r, (void **)&r->colors, r->n_color_cu * sizeof(fs_color16), kFrame));
    if (uint32_t e = fs_clear(r))
        return e;
    // the frame buffers were allocated in compute-stream order; the display stream (and the caller's own streams, through
    // fs_device_iter_buffer) may use them from here on
    FS_TRY(hipStreamSynchronize(r->compute));
    return 0;
}

uint32_t fs_set_row_bands(fs_renderer *r, uint32_t band_first_row, uint32_t band_rows, uint32_t band_stride_rows)
{
    if (uint32_t e = use_device(r))
        return e;
    if (r->width == 0)
        return FS_ERR_6;
    if (band_rows != 0 && (band_stride_rows < band_rows))
        return FS_ERR_7;
    r->band_first = band_first_row;
    r->band_rows = band_rows;
    r->band_stride = band_stride_rows;
    compute_local_rows(r);
    const uint32_t e = ensure_iter_buffer(r);
    if (e && r->iters_external) { // the caller's buffer does not hold the new banding: fall back to the internal one
        r->iters_external = nullptr;
        r->iters_external_bytes = 0;
        (void)ensure_iter_buffer(r);
    }
    return e;
}

uint32_t fs_local_rows(const fs_renderer *r) { return r->local_rows_padded; }

uint32_t fs_set_external_iter_buffer(fs_renderer *r, void *device_ptr, uint64_t capacity_bytes)
{
    if (uint32_t e = use_device(r))
        return e;
    if (r->width == 0)
        return FS_ERR_6;
    r->iters_external = device_ptr;
    r->iters_external_bytes = device_ptr ? (size_t)capacity_bytes : 0;
    const uint32_t e = ensure_iter_buffer(r); // too small for the current geometry: rejected, internal buffer restored
    if (e && device_ptr) {
        r->iters_external = nullptr;
        r->iters_external_bytes = 0;
        (void)ensure_iter_buffer(r);
    }
    return e;
}

void *fs_device_iter_buffer(const fs_renderer *r) { return r->iters(); }

// The renderer's bands -> their rows of a WHOLE-FRAME host buffer, over THIS device's own PCIe link (round 6; the sharded
// read-back of the row-tiled frame: GPURenderer::ExtractItersAndColors, GPU_Render.cu:1760-1805, copies N_cu counts per frame
// through one device).  The local buffer holds the owned bands back to back and band k belongs at frame row
// band_first + k * band_stride: ONE two-dimensional copy whose "row" is a whole band (band_rows x pitch bytes) and whose
// destination pitch is the band stride, so the rows land in frame order and nothing has to restore it; a last, shorter band
// goes by itself.
uint32_t fs_copy_bands_to_host(fs_renderer *r, const void *device_iters, void *host_frame, void *stream)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized() || !host_frame)
        return host_frame ? 0u : (uint32_t)hipErrorInvalidValue;
    if (r->local_rows == 0)
        return 0;
    const char *src = (const char *)(device_iters ? device_iters : r->iters());
    hipStream_t s = stream ? (hipStream_t)stream : r->compute;
    const size_t pitch = (size_t)r->w_block * 16u * r->iter_bytes;
    const uint64_t H = r->height, first = r->band_first, rows = r->band_rows, stride = r->band_stride;
    if (first == 0 && rows >= H) // no banding: the whole padded buffer, as fs_render_current copies it
        return (uint32_t)hipMemcpyAsync(host_frame, src, (size_t)r->local_rows_padded * pitch, hipMemcpyDeviceToHost, s);
    uint64_t full = 0; // bands that lie wholly inside the frame
    if (first + rows <= H)
        full = (H - rows - first) / stride + 1u;
    char *dst = (char *)host_frame + first * pitch;
    if (full == 1u || (full > 1u && stride == rows)) {
        FS_TRY(hipMemcpyAsync(dst, src, full * rows * pitch, hipMemcpyDeviceToHost, s));
    } else if (full > 1u) {
        FS_TRY(hipMemcpy2DAsync(dst, stride * pitch, src, rows * pitch, rows * pitch, full, hipMemcpyDeviceToHost, s));
    }
    const uint64_t tail_start = first + full * stride;
    if (tail_start < H) { // the last band is cut by the frame's edge
        const uint64_t tail_rows = (tail_start + rows < H ? tail_start + rows : H) - tail_start;
        FS_TRY(hipMemcpyAsync((char *)host_frame + tail_start * pitch, src + full * rows * pitch, tail_rows * pitch,
                              hipMemcpyDeviceToHost, s));
    }
    return 0;
}

uint32_t fs_host_register(void *host_ptr, uint64_t bytes)
{
    return (uint32_t)hipHostRegister(host_ptr, (size_t)bytes, hipHostRegisterPortable);
}
uint32_t fs_host_unregister(void *host_ptr) { return (uint32_t)hipHostUnregister(host_ptr); }
uint32_t fs_rounded_width(const fs_renderer *r) { return r->w_block * 16u; }

// FNV-1a over the size, the period and up to 4096 evenly spread 8-byte words of an orbit's entries (never 0)
static uint64_t orbit_fingerprint(const void *entries, uint64_t bytes, uint64_t size, uint64_t period, int type_tag)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](uint64_t v) {
        for (int i = 0; i < 8; i++) {
            h ^= (v >> (8 * i)) & 0xFFu;
            h *= 1099511628211ull;
        }
    };
    mix(size), mix(period), mix((uint64_t)type_tag);
    const uint64_t words = bytes / 8u;
    const uint64_t stride = words > 4096u ? words / 4096u : 1u;
    const unsigned char *p = (const unsigned char *)entries;
    for (uint64_t w = 0; w < words; w += stride) {
        uint64_t v;
        memcpy(&v, p + w * 8u, 8);
        mix(v);
    }
    if (words != 0u) { // the last word, whatever the stride
        uint64_t v;
        memcpy(&v, p + (words - 1u) * 8u, 8);
        mix(v);
    }
    return h != 0ull ? h : 1ull;
}

// called where an upload has replaced the resident orbit
static void bump_orbit_epoch(fs_renderer *r)
{
    if (r->pending_fp == 0ull || r->pending_fp != r->orbit_fp)
        r->orbit_epoch++;
    r->orbit_fp = r->pending_fp;
    r->pending_fp = 0ull;
}

uint32_t fs_upload_orbit(fs_renderer *r, uint64_t generation, int type_tag, uint32_t iter_bytes, const void *entries,
                         uint64_t orbit_size, uint64_t uncompressed_size, uint64_t period_maybe_zero)
{
    if (uint32_t e = use_device(r))
        return e;
    // orbit entries do not depend on IterType (GPU_ReferenceIter.h:52-127); counts must fit the 32-bit device counters
    if ((type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64 && type_tag != FS_T_F64 && type_tag != FS_T_HDR2X32 &&
         type_tag != FS_T_F32 && type_tag != FS_T_2X32) ||
        (iter_bytes != 4 && iter_bytes != 8) || uncompressed_size > 0xFFFFFFFFull)
        return FS_ERR_UNSUPPORTED;
    if (!r->compute)
        return FS_ERR_6;
    if (r->orbit_ok && r->orbit_gen == generation && generation != 0 && r->orbit_type == type_tag)
        return 0; // cached by generation number (GPU_Render.cu:440-487)
    if (type_tag == FS_T_F32 || type_tag == FS_T_2X32) {
        // GPUReferenceIter<float,Disable> (8 B) / GPUReferenceIter<CudaDblflt,Disable> (16 B), used as uploaded
        const size_t eb = type_tag == FS_T_F32 ? sizeof(fs_orbit_f32) : sizeof(fs_orbit_p2x32);
        if (r->orbit_plain) {
            FS_TRY(r_free(r, r->orbit_plain));
            r->orbit_plain = nullptr;
        }
        r->orbit_ok = false;
        drop_seq(r);
        FS_TRY(r_alloc(r, &r->orbit_plain, (orbit_size + 1) * eb, kInput));
        FS_TRY(hipMemcpyAsync(r->orbit_plain, entries, orbit_size * eb, hipMemcpyDefault, r->compute));
        FS_TRY(hipStreamSynchronize(r->compute));
        r->orbit_size = orbit_size;
        r->orbit_uncompressed = uncompressed_size;
        r->orbit_period = period_maybe_zero;
        r->orbit_gen = generation, bump_orbit_epoch(r);
        r->orbit_type = type_tag;
        r->orbit_ok = true;
        orbit_changed(r);
        return 0;
    }
    if (type_tag == FS_T_HDR2X32) {
        if (r->orbit_2x32) {
            FS_TRY(r_free(r, r->orbit_2x32));
            r->orbit_2x32 = nullptr;
        }
        r->orbit_ok = false;
        drop_seq(r);
        FS_TRY(r_alloc(r, (void **)&r->orbit_2x32, orbit_size * sizeof(fs_orbit_2x32), kInput));
        FS_TRY(hipMemcpyAsync(r->orbit_2x32, entries, orbit_size * sizeof(fs_orbit_2x32), hipMemcpyDefault, r->compute));
        FS_TRY(hipStreamSynchronize(r->compute));
        r->orbit_size = orbit_size;
        r->orbit_uncompressed = uncompressed_size;
        r->orbit_period = period_maybe_zero;
        r->orbit_gen = generation, bump_orbit_epoch(r);
        r->orbit_type = type_tag;
        r->orbit_ok = true;
        orbit_changed(r);
        return 0;
    }
    if (type_tag == FS_T_F64) {
        if (r->orbit_f64) {
            FS_TRY(r_free(r, r->orbit_f64));
            r->orbit_f64 = nullptr;
        }
        r->orbit_ok = false;
        drop_seq(r);
        FS_TRY(r_alloc(r, (void **)&r->orbit_f64, orbit_size * sizeof(fs_orbit_f64), kInput));
        FS_TRY(hipMemcpyAsync(r->orbit_f64, entries, orbit_size * sizeof(fs_orbit_f64), hipMemcpyDefault, r->compute));
        FS_TRY(hipStreamSynchronize(r->compute));
        r->orbit_size = orbit_size;
        r->orbit_uncompressed = uncompressed_size;
        r->orbit_period = period_maybe_zero;
        r->orbit_gen = generation, bump_orbit_epoch(r);
        r->orbit_type = type_tag;
        r->orbit_ok = true;
        orbit_changed(r);
        return 0;
    }
    if (r->zref) {
        FS_TRY(r_free(r, r->zref));
        r->zref = nullptr;
    }
    if (r->zref64) {
        FS_TRY(r_free(r, r->zref64));
        r->zref64 = nullptr;
    }
    r->orbit_ok = false;
    drop_seq(r);
    const size_t in_bytes = type_tag == FS_T_HDR32 ? sizeof(fs_orbit_hdr32) : sizeof(fs_orbit_hdr64);
    r->pending_fp = orbit_fingerprint(entries, orbit_size * in_bytes, orbit_size, period_maybe_zero, type_tag);
    void *raw = nullptr;
    FS_TRY(r_alloc(r, &raw, orbit_size * in_bytes, kInput));
    // two spare entries: the tuned loops may prefetch one entry past the end
    hipError_t err = type_tag == FS_T_HDR32 ? r_alloc(r, (void **)&r->zref, (orbit_size + 2) * sizeof(float4), kInput)
                                            : r_alloc(r, (void **)&r->zref64, (orbit_size + 2) * sizeof(FsZ64), kInput);
    if (err == hipSuccess)
        err = hipMemcpyAsync(raw, entries, orbit_size * in_bytes, hipMemcpyDefault, r->compute);
    if (err == hipSuccess) {
        if (type_tag == FS_T_HDR32) {
            err = hipMemsetAsync(r->zref + orbit_size, 0, 2 * sizeof(float4), r->compute);
            fsk_prepare_orbit_hdr32((const fs_orbit_hdr32 *)raw, r->zref, orbit_size, r->compute);
            if (err == hipSuccess)
                err = make_quiet_orbit(r, orbit_size);
        } else {
            err = hipMemsetAsync(r->zref64 + orbit_size, 0, 2 * sizeof(FsZ64), r->compute);
            fsk_prepare_orbit_hdr64((const fs_orbit_hdr64 *)raw, r->zref64, orbit_size, r->compute);
        }
        if (err == hipSuccess)
            err = hipGetLastError();
    }
    if (err == hipSuccess)
        err = hipStreamSynchronize(r->compute);
    (void)r_free(r, raw);
    if (err != hipSuccess)
        return (uint32_t)err;
    r->orbit_size = orbit_size;
    r->orbit_uncompressed = uncompressed_size;
    r->orbit_period = period_maybe_zero;
    r->orbit_gen = generation, bump_orbit_epoch(r);
    r->orbit_type = type_tag;
    r->orbit_ok = true;
    orbit_changed(r);
    return 0;
}

uint32_t fs_upload_orbit_compressed(fs_renderer *r, uint64_t generation, int type_tag, uint32_t iter_bytes,
                                    const void *entries, uint64_t compressed_size, uint64_t uncompressed_size,
                                    uint64_t period_maybe_zero, const void *orbit_x_low, const void *orbit_y_low)
{
    if (uint32_t e = use_device(r))
        return e;
    if ((type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64 && type_tag != FS_T_F32 && type_tag != FS_T_F64 &&
         type_tag != FS_T_2X32 && type_tag != FS_T_HDR2X32) ||
        (iter_bytes != 4 && iter_bytes != 8) || !orbit_x_low || !orbit_y_low)
        return FS_ERR_UNSUPPORTED;
    if (!r->compute)
        return FS_ERR_6;
    const bool want_seq = r->compressed_mode == 1;
    // an EXPANDED orbit must fit the 32-bit positions of the kernels that read it (and the device: 2^32 entries are 64 GiB
    // and more); a waypoint-resident one may be any length -- its positions are 64-bit in the kernel that walks it
    if (!want_seq && uncompressed_size > 0xFFFFFFFFull)
        return FS_ERR_UNSUPPORTED; // (fs_set_compressed_orbit_mode(1) serves such an orbit)
    if (r->orbit_ok && r->orbit_gen == generation && generation != 0 && r->orbit_type == type_tag && r->orbit_seq == want_seq)
        return 0;
    if (type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64) {
        // float / double / CudaDblflt / HDRFloat<CudaDblflt>: expanded into the record array the uncompressed upload
        // of that type fills (kernels_decompress.hip)
        size_t in_b, out_b;
        void **slot;
        switch (type_tag) {
        case FS_T_F32:
            in_b = sizeof(fs_orbit_f32_rc), out_b = sizeof(fs_orbit_f32), slot = &r->orbit_plain;
            break;
        case FS_T_2X32:
            in_b = sizeof(fs_orbit_p2x32_rc), out_b = sizeof(fs_orbit_p2x32), slot = &r->orbit_plain;
            break;
        case FS_T_F64:
            in_b = sizeof(fs_orbit_f64_rc), out_b = sizeof(fs_orbit_f64), slot = (void **)&r->orbit_f64;
            break;
        default:
            in_b = sizeof(fs_orbit_2x32_rc), out_b = sizeof(fs_orbit_2x32), slot = (void **)&r->orbit_2x32;
            break;
        }
        if (*slot) {
            FS_TRY(r_free(r, *slot));
            *slot = nullptr;
        }
        r->orbit_ok = false;
        drop_seq(r);
        if (want_seq) {
            // keep the waypoints, nothing else: k_lav2_plain / k_lav2_2x32 walk them with a cursor per pixel (same values as
            // the expansion below, entry for entry)
            if (compressed_size == 0 || compressed_size > 0xFFFFFFFFull || uncompressed_size > 0xFFFFFFFFull)
                return FS_ERR_UNSUPPORTED; // (these kernels keep 32-bit positions)
            const size_t low_b = type_tag == FS_T_F32 ? sizeof(float) : (type_tag == FS_T_HDR2X32 ? sizeof(fs_real_2x32) : 8u);
            FS_TRY(r_alloc(r, &r->wp_raw, compressed_size * in_b, kInput));
            FS_TRY(hipMemcpyAsync(r->wp_raw, entries, compressed_size * in_b, hipMemcpyDefault, r->compute));
            FS_TRY(hipStreamSynchronize(r->compute));
            memset(r->c_low_plain, 0, sizeof(r->c_low_plain));
            memcpy(r->c_low_plain[0], orbit_x_low, low_b);
            memcpy(r->c_low_plain[1], orbit_y_low, low_b);
            r->orbit_seq = true;
            r->orbit_size = compressed_size;
            r->orbit_uncompressed = uncompressed_size;
            r->orbit_period = period_maybe_zero;
            r->orbit_gen = generation, bump_orbit_epoch(r);
            r->orbit_type = type_tag;
            r->orbit_ok = true;
            orbit_changed(r);
            return 0;
        }
        void *raw = nullptr;
        FS_TRY(r_alloc(r, &raw, compressed_size * in_b, kInput));
        hipError_t err = r_alloc(r, slot, (uncompressed_size + 1) * out_b, kInput);
        if (err == hipSuccess)
            err = hipMemcpyAsync(raw, entries, compressed_size * in_b, hipMemcpyDefault, r->compute);
        if (err == hipSuccess)
            err = hipMemsetAsync((char *)*slot + uncompressed_size * out_b, 0, out_b, r->compute);
        if (err == hipSuccess) {
            fsk_decompress_orbit_plain(type_tag, raw, compressed_size, uncompressed_size, orbit_x_low, orbit_y_low, *slot,
                                       r->compute);
            err = hipGetLastError();
        }
        if (err == hipSuccess)
            err = hipStreamSynchronize(r->compute);
        (void)r_free(r, raw);
        if (err != hipSuccess)
            return (uint32_t)err;
        r->orbit_size = compressed_size;
        r->orbit_uncompressed = uncompressed_size;
        r->orbit_period = period_maybe_zero;
        r->orbit_gen = generation, bump_orbit_epoch(r);
        r->orbit_type = type_tag;
        r->orbit_ok = true;
        orbit_changed(r);
        return 0;
    }
    if (r->zref) {
        FS_TRY(r_free(r, r->zref));
        r->zref = nullptr;
    }
    if (r->zref64) {
        FS_TRY(r_free(r, r->zref64));
        r->zref64 = nullptr;
    }
    r->orbit_ok = false;
    drop_seq(r);
    const size_t in_bytes = type_tag == FS_T_HDR32 ? sizeof(fs_orbit_hdr32_rc) : sizeof(fs_orbit_hdr64_rc);
    if (r->compressed_mode == 1) {
        // keep the waypoints, nothing else: the kernel decompresses as it goes (GPUPerturbSingleResults for
        // PerturbExtras::SimpleCompression uploads exactly this array, Perturb.cuh:51-80)
        if (compressed_size == 0 || compressed_size > 0xFFFFFFFFull)
            return FS_ERR_UNSUPPORTED;
        FS_TRY(r_alloc(r, &r->wp_raw, compressed_size * in_bytes, kInput));
        FS_TRY(hipMemcpyAsync(r->wp_raw, entries, compressed_size * in_bytes, hipMemcpyDefault, r->compute));
        FS_TRY(hipStreamSynchronize(r->compute));
        if (type_tag == FS_T_HDR32) {
            r->c_low32[0] = *(const fs_real_hdr32 *)orbit_x_low;
            r->c_low32[1] = *(const fs_real_hdr32 *)orbit_y_low;
        } else {
            r->c_low64[0] = *(const fs_real_hdr64 *)orbit_x_low;
            r->c_low64[1] = *(const fs_real_hdr64 *)orbit_y_low;
        }
        r->orbit_seq = true;
        r->orbit_size = compressed_size;
        r->orbit_uncompressed = uncompressed_size;
        r->orbit_period = period_maybe_zero;
        r->orbit_gen = generation, bump_orbit_epoch(r);
        r->orbit_type = type_tag;
        r->orbit_ok = true;
        orbit_changed(r);
        return 0;
    }
    void *raw = nullptr;
    FS_TRY(r_alloc(r, &raw, compressed_size * in_bytes, kInput));
    hipError_t err = type_tag == FS_T_HDR32 ? r_alloc(r, (void **)&r->zref, (uncompressed_size + 2) * sizeof(float4), kInput)
                                            : r_alloc(r, (void **)&r->zref64, (uncompressed_size + 2) * sizeof(FsZ64), kInput);
    if (err == hipSuccess)
        err = hipMemcpyAsync(raw, entries, compressed_size * in_bytes, hipMemcpyDefault, r->compute);
    if (err == hipSuccess) {
        if (type_tag == FS_T_HDR32) {
            err = hipMemsetAsync(r->zref + uncompressed_size, 0, 2 * sizeof(float4), r->compute);
            fsk_decompress_orbit_hdr32((const fs_orbit_hdr32_rc *)raw, compressed_size, uncompressed_size,
                                       *(const fs_real_hdr32 *)orbit_x_low, *(const fs_real_hdr32 *)orbit_y_low, r->zref,
                                       r->compute);
            if (err == hipSuccess)
                err = make_quiet_orbit(r, uncompressed_size);
        } else {
            err = hipMemsetAsync(r->zref64 + uncompressed_size, 0, 2 * sizeof(FsZ64), r->compute);
            fsk_decompress_orbit_hdr64((const fs_orbit_hdr64_rc *)raw, compressed_size, uncompressed_size,
                                       *(const fs_real_hdr64 *)orbit_x_low, *(const fs_real_hdr64 *)orbit_y_low, r->zref64,
                                       r->compute);
        }
        if (err == hipSuccess)
            err = hipGetLastError();
    }
    if (err == hipSuccess)
        err = hipStreamSynchronize(r->compute);
    (void)r_free(r, raw);
    if (err != hipSuccess)
        return (uint32_t)err;
    r->orbit_size = compressed_size;
    r->orbit_uncompressed = uncompressed_size;
    r->orbit_period = period_maybe_zero;
    r->orbit_gen = generation, bump_orbit_epoch(r);
    r->orbit_type = type_tag;
    r->orbit_ok = true;
    orbit_changed(r);
    return 0;
}

uint32_t fs_upload_la(fs_renderer *r, uint64_t generation, int type_tag, uint32_t iter_bytes, const void *las,
                      uint32_t n_las, const void *stages, uint32_t n_stages, int is_valid, int use_at,
                      const void *at_info)
{
    if (uint32_t e = use_device(r))
        return e;
    const bool plain = type_tag == FS_T_F32 || type_tag == FS_T_F64 || type_tag == FS_T_2X32;
    if ((type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64 && type_tag != FS_T_HDR2X32 && !plain) ||
        (iter_bytes != 4 && iter_bytes != 8))
        return FS_ERR_UNSUPPORTED;
    if (!r->compute)
        return FS_ERR_6;
    if (r->la_ok && r->la_gen == generation && generation != 0 && r->la_type == type_tag)
        return 0;
    const size_t la_bytes = type_tag == FS_T_HDR32   ? sizeof(fs_la_hdr32_u32)
                            : type_tag == FS_T_HDR64 ? sizeof(fs_la_hdr64_u32)
                            : type_tag == FS_T_F32   ? sizeof(fs_la_f32_u32)
                            : type_tag == FS_T_F64   ? sizeof(fs_la_f64_u32)
                            : type_tag == FS_T_2X32  ? sizeof(fs_la_p2x32_u32)
                                                     : sizeof(fs_la_2x32_u32);
    // size of the uint32_t ATInfo record and the offset of its second field in the uint32_t / uint64_t records
    const size_t at_bytes = type_tag == FS_T_HDR32   ? sizeof(fs_at_hdr32_u32)
                            : type_tag == FS_T_HDR64 ? sizeof(fs_at_hdr64_u32)
                            : type_tag == FS_T_F32   ? sizeof(fs_at_f32_u32)
                            : type_tag == FS_T_F64   ? sizeof(fs_at_f64_u32)
                            : type_tag == FS_T_2X32  ? sizeof(fs_at_p2x32_u32)
                                                     : sizeof(fs_at_2x32_u32);
    const size_t at_rest32 = (type_tag == FS_T_HDR64 || type_tag == FS_T_F64) ? 8 : 4;
    std::vector<uint8_t> las32, stages32;
    uint8_t at32[sizeof(fs_at_hdr64_u32)] = {0};
    bool keep_u64 = false; // the LA records stay in the reference's uint64_t layout
    uint32_t at_step_hi = 0;
    size_t la_bytes_up = la_bytes;
    if (iter_bytes == 8) {
        bool ok = true;
        // HDRFloat<float | double> tables under fs_set_compressed_orbit_mode(1) are read by the waypoint-resident kernel,
        // whose wide instantiation takes the uint64_t records as they are: kept whenever a step length or index does not fit
        // 32 bits (an orbit of 2^32 and more uncompressed entries), and under the FS_VARIANT_WIDE_COUNTERS test switch
        const bool can_keep = (type_tag == FS_T_HDR32 || type_tag == FS_T_HDR64) && r->compressed_mode == 1;
        keep_u64 = can_keep && (r->variant & FS_VARIANT_FLAG_WIDE) != 0;
        if (n_las && !keep_u64) {
            ok = type_tag == FS_T_HDR32   ? narrow_la<fs_la_hdr32_u64, fs_la_hdr32_u32>(las, n_las, las32)
                 : type_tag == FS_T_HDR64 ? narrow_la<fs_la_hdr64_u64, fs_la_hdr64_u32>(las, n_las, las32)
                 : type_tag == FS_T_F32   ? narrow_la<fs_la_f32_u64, fs_la_f32_u32>(las, n_las, las32)
                 : type_tag == FS_T_F64   ? narrow_la<fs_la_f64_u64, fs_la_f64_u32>(las, n_las, las32)
                 : type_tag == FS_T_2X32  ? narrow_la<fs_la_p2x32_u64, fs_la_p2x32_u32>(las, n_las, las32)
                                          : narrow_la<fs_la_2x32_u64, fs_la_2x32_u32>(las, n_las, las32);
            if (!ok && can_keep)
                keep_u64 = ok = true;
        }
        stages32.resize((size_t)n_stages * sizeof(fs_la_stage_u32));
        for (uint32_t i = 0; ok && i < n_stages; i++) {
            // (a stage's first record and its record count index the table itself, whose size is a uint32_t)
            const fs_la_stage_u64 &sg = ((const fs_la_stage_u64 *)stages)[i];
            if (sg.LAIndex > 0xFFFFFFFFull || sg.MacroItCount > 0xFFFFFFFFull)
                return (uint32_t)hipErrorInvalidValue;
            ((fs_la_stage_u32 *)stages32.data())[i] = fs_la_stage_u32{(uint32_t)sg.LAIndex, (uint32_t)sg.MacroItCount};
        }
        if (ok && at_info) {
            uint64_t step;
            memcpy(&step, at_info, 8);
            if (keep_u64)
                at_step_hi = (uint32_t)(step >> 32);
            else
                ok = step <= 0xFFFFFFFFull;
            const uint32_t step32 = (uint32_t)step;
            memcpy(at32, &step32, 4);
            // everything after StepLength is laid out identically; it starts at offset 8 in the uint64_t record
            memcpy(at32 + at_rest32, (const uint8_t *)at_info + 8, at_bytes - at_rest32);
            at_info = at32;
        }
        if (!ok)
            return FS_ERR_UNSUPPORTED; // a step length / index beyond 32 bits for a kernel that reads an EXPANDED orbit
        if (keep_u64)
            la_bytes_up = type_tag == FS_T_HDR32 ? sizeof(fs_la_hdr32_u64) : sizeof(fs_la_hdr64_u64);
        else
            las = las32.data();
        stages = stages32.data();
    }
    r->la_ok = false;
    FS_TRY(la_reserve(r, (size_t)n_las * la_bytes_up, (size_t)n_stages * sizeof(fs_la_stage_u32)));
    if (n_las)
        FS_TRY(hipMemcpyAsync(r->las, las, (size_t)n_las * la_bytes_up, hipMemcpyDefault, r->compute));
    r->la_u64 = keep_u64;
    r->at_step_hi = at_step_hi;
    if (n_stages)
        FS_TRY(hipMemcpyAsync(r->stages, stages, (size_t)n_stages * sizeof(fs_la_stage_u32), hipMemcpyDefault,
                              r->compute));
    FS_TRY(hipStreamSynchronize(r->compute));
    r->n_las = n_las;
    r->n_stages = n_stages;
    r->la_valid = is_valid;
    r->use_at = use_at;
    memset(&r->at, 0, sizeof(r->at));
    memset(&r->at64, 0, sizeof(r->at64));
    memset(&r->at2x32, 0, sizeof(r->at2x32));
    memset(r->at_plain, 0, sizeof(r->at_plain));
    if (at_info && plain)
        memcpy(r->at_plain, at_info, at_bytes);
    else if (at_info && type_tag == FS_T_HDR32)
        memcpy(&r->at, at_info, sizeof(r->at));
    else if (at_info && type_tag == FS_T_HDR2X32)
        memcpy(&r->at2x32, at_info, sizeof(r->at2x32));
    else if (at_info)
        memcpy(&r->at64, at_info, sizeof(r->at64));
    else
        r->use_at = 0;
    r->la_type = type_tag;
    r->la_gen = generation;
    r->la_ok = true;
    return 0;
}

uint32_t fs_upload_bla(fs_renderer *r, int type_tag, const void *const *levels, const uint64_t *level_sizes,
                       int32_t n_levels, int32_t lm2)
{
    if (uint32_t e = use_device(r))
        return e;
    if (type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64 && type_tag != FS_T_F64)
        return FS_ERR_UNSUPPORTED;
    if (!r->compute)
        return FS_ERR_6;
    const size_t rec_bytes = type_tag == FS_T_HDR32 ? sizeof(fs_bla_hdr32)
                                                    : (type_tag == FS_T_HDR64 ? sizeof(fs_bla_hdr64) : sizeof(fs_bla_f64));
    r->bla_type = type_tag;
    r->bla_n_levels = 0;
    if (n_levels <= 0)
        return 0;
    std::vector<uint64_t> sizes((size_t)n_levels, 0);
    for (int32_t l = 0; l < n_levels; l++)
        sizes[(size_t)l] = levels[l] ? level_sizes[l] : 0;
    FS_TRY(bla_layout(r, sizes.data(), n_levels, rec_bytes));
    for (int32_t l = 0; l < n_levels; l++)
        if (sizes[(size_t)l])
            FS_TRY(hipMemcpyAsync(r->bla_level_mem[(size_t)l], levels[l], sizes[(size_t)l] * rec_bytes, hipMemcpyDefault,
                                  r->compute));
    FS_TRY(hipStreamSynchronize(r->compute)); // the host levels are borrowed for the call only
    r->bla_n_levels = n_levels;
    r->bla_lm2 = lm2;
    r->bla_native_ok = false;
    r->bla_native_stale = type_tag == FS_T_HDR32; // made by the next BLA render: it also needs the orbit of that render
    return 0;
}

uint32_t fs_build_bla(fs_renderer *r, int type_tag, const void *bla_size)
{
    if (uint32_t e = use_device(r))
        return e;
    if (type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64)
        return FS_ERR_UNSUPPORTED;
    if (!r->compute || !r->orbit_ok || r->orbit_type != type_tag)
        return FS_ERR_6;
    if (r->orbit_seq)
        return FS_ERR_UNSUPPORTED; // needs the expanded orbit (fs_set_compressed_orbit_mode 0)
    const size_t rec_bytes = type_tag == FS_T_HDR32 ? sizeof(fs_bla_hdr32) : sizeof(fs_bla_hdr64);
    r->bla_n_levels = 0;
    r->bla_type = type_tag;
    // BLAS::Init, BLAS.cpp:218-241: elements per level halve (rounding up) from count-1 down to 1
    const uint64_t InM = r->orbit_uncompressed;
    uint64_t m = InM ? InM - 1 : 0;
    if (InM == 0 || m == 0)
        return 0;
    std::vector<uint64_t> epl;
    for (; m > 1; m = (m + 1) >> 1)
        epl.push_back(m);
    epl.push_back(m);
    const int n_levels = (int)epl.size();
    int32_t lm2 = n_levels - 2;
    if (lm2 < 0)
        lm2 = 0;
    std::vector<uint64_t> materialised(epl); // m_FirstLevel = 2: levels 0 and 1 get no memory (NULL pointers)
    materialised[0] = 0;
    if (n_levels > 1)
        materialised[1] = 0;
    FS_TRY(bla_layout(r, materialised.data(), n_levels, rec_bytes));
    const std::vector<void *> &ptrs = r->bla_level_mem;
    {
        TimedLaunch t(r);
        if (type_tag == FS_T_HDR32)
            fsk_bla_build_hdr32(r->zref, ptrs.data(), epl.data(), n_levels, *(const fs_real_hdr32 *)bla_size, r->compute);
        else
            fsk_bla_build_hdr64(r->zref64, ptrs.data(), epl.data(), n_levels, *(const fs_real_hdr64 *)bla_size, r->compute);
    }
    FS_TRY(hipGetLastError());
    FS_TRY(hipStreamSynchronize(r->compute)); // ptrs / epl are host temporaries of this call
    r->bla_n_levels = n_levels;
    r->bla_lm2 = lm2;
    r->bla_native_ok = false;
    r->bla_native_stale = type_tag == FS_T_HDR32; // made by the next BLA render: it also needs the orbit of that render
    return 0;
}

} // extern "C" (the builder below is a template)

// ---- LAv2 table built on the device (kernels_la.hip): the scalar decisions of LAReference.cpp on the host, everything
// that touches the orbit or a record on the device.  See the header of kernels_la.hip for the algorithm.
namespace {

// Work arrays of one fs_build_la call, carved out of the renderer's arena (no allocation once the arena has grown to the
// largest orbit seen): 256-byte aligned slices handed out front to back.
struct ArenaSlice {
    void *p;
    template <class T> T *as() const { return (T *)p; }
};
struct ArenaCarver {
    char *base;
    size_t used = 0;
    explicit ArenaCarver(void *b) : base((char *)b) {}
    static size_t padded(size_t bytes) { return (bytes + 255u) & ~(size_t)255u; }
    template <class T> T *take(size_t bytes)
    {
        T *p = (T *)(base + used);
        used += padded(bytes ? bytes : 16);
        return p;
    }
};

template <class F> void pack_at(const fs::la::ATInfoT<F> &a, fs_renderer *r);
template <> void pack_at<float>(const fs::la::ATInfoT<float> &a, fs_renderer *r)
{
    auto R = [](fs::hreal<float> h) { return fs_real_hdr32{h.m, h.e}; };
    auto C = [](fs::hcplx<float> c) { return fs_cplx_hdr32{c.re, c.im, c.e}; };
    fs_at_hdr32_u32 &o = r->at;
    memset(&o, 0, sizeof(o));
    o.StepLength = a.StepLength;
    o.ThresholdC = R(a.ThresholdC), o.SqrEscapeRadius = R(a.SqrEscapeRadius);
    o.RefC = C(a.RefC), o.ZCoeff = C(a.ZCoeff), o.CCoeff = C(a.CCoeff), o.InvZCoeff = C(a.InvZCoeff);
    o.CCoeffSqrInvZCoeff = C(a.CCoeffSqrInvZCoeff), o.CCoeffInvZCoeff = C(a.CCoeffInvZCoeff);
    o.CCoeffNormSqr = R(a.CCoeffNormSqr), o.RefCNormSqr = R(a.RefCNormSqr), o.factor = R(a.factor);
}
template <> void pack_at<double>(const fs::la::ATInfoT<double> &a, fs_renderer *r)
{
    auto R = [](fs::hreal<double> h) { return fs_real_hdr64{h.m, h.e, 0}; };
    auto C = [](fs::hcplx<double> c) { return fs_cplx_hdr64{c.re, c.im, c.e, 0}; };
    fs_at_hdr64_u32 &o = r->at64;
    memset(&o, 0, sizeof(o));
    o.StepLength = a.StepLength;
    o.ThresholdC = R(a.ThresholdC), o.SqrEscapeRadius = R(a.SqrEscapeRadius);
    o.RefC = C(a.RefC), o.ZCoeff = C(a.ZCoeff), o.CCoeff = C(a.CCoeff), o.InvZCoeff = C(a.InvZCoeff);
    o.CCoeffSqrInvZCoeff = C(a.CCoeffSqrInvZCoeff), o.CCoeffInvZCoeff = C(a.CCoeffInvZCoeff);
    o.CCoeffNormSqr = R(a.CCoeffNormSqr), o.RefCNormSqr = R(a.RefCNormSqr), o.factor = R(a.factor);
}

constexpr uint32_t kLaLowBound = 64;    // LAReference.h:56
constexpr uint32_t kLaMaxStages = 1024; // LAReference.h
constexpr uint32_t kLaTerm = 0xFFFFFFFFu;

template <class F> uint32_t build_la(fs_renderer *r, const void *max_radius, int use_small_exponents, int host_threads)
{
    using Rec = fs::la::LAInfo<F>;
    using HR = fs::hreal<F>;
    hipStream_t s = r->compute;
    const void *zref = sizeof(F) == 4 ? (const void *)r->zref : (const void *)r->zref64;
    // state numbers (2 per element) and record indices are 32-bit on the device: an orbit of 2^31 entries does not fit
    // (its prepared form alone would be 32 GiB of float4); refuse instead of truncating
    if (r->orbit_uncompressed >= (1ull << 31))
        return FS_ERR_UNSUPPORTED;
    const uint32_t maxRef = (uint32_t)r->orbit_uncompressed - 1u; // entries 0 .. maxRef
    const int periodDivisor = r->orbit_size != r->orbit_uncompressed ? 8 : 2; // LAReference.cpp:12-19
    if (r->orbit_uncompressed < 3)
        return FS_ERR_UNSUPPORTED; // (maxRefIteration == 0: no table, LAReference.cpp:981-984; one step: left to the host builder)
    // capacity: a stage never holds more records than elements it was folded from (+ its tail record)
    const size_t cap_states = 2u * ((size_t)maxRef + 2u);
    // all stages: stage k+1 holds at most half of stage k (+2), so 2 * maxRef + slack bounds the sum
    const size_t cap_recs = 2u * (size_t)maxRef + 64u * kLaLowBound;
    const size_t sizes[13] = {sizeof(HR) * (maxRef + 2u), sizeof(HR) * (maxRef + 2u), 4u * (maxRef + 2u), 4u * (maxRef + 3u),
                              4u * cap_states,            4u * cap_states,            4u * cap_states,     4u * cap_states,
                              4u * (cap_states + 1u),     sizeof(Rec) * cap_recs,     64,                  4u * kLaMaxStages,
                              sizeof(fs::la::ATInfoT<F>)};
    size_t total = 0;
    for (size_t b : sizes)
        total += ArenaCarver::padded(b);
    FS_TRY(arena_reserve(r, total));
    ArenaCarver carve(r->arena);
    ArenaSlice chebv{carve.take<char>(sizes[0])}, mm{carve.take<char>(sizes[1])}, steps{carve.take<char>(sizes[2])},
        pos{carve.take<char>(sizes[3])}, nextA{carve.take<char>(sizes[4])}, nextB{carve.take<char>(sizes[5])},
        nextC{carve.take<char>(sizes[6])}, reach{carve.take<char>(sizes[7])}, rank{carve.take<char>(sizes[8])},
        table{carve.take<char>(sizes[9])}, small{carve.take<char>(sizes[10])}, stage_idx{carve.take<char>(sizes[11])},
        atbuf{carve.take<char>(sizes[12])};
    uint32_t *d_small = small.as<uint32_t>();
    Rec *d_table = table.as<Rec>();

    std::vector<fs_la_stage_u32> stages;
    uint32_t la_size = 0;
    uint32_t h[4];

    // A few words from the device: through the mailbox (a tiny kernel writes them into coherent page-locked memory and then
    // a sequence number; the host spins on that word) -- or, if the mailbox could not be had or stays silent, the plain way
    if (!r->la_mail) {
        if (hipHostMalloc((void **)&r->la_mail, 32 * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            r->la_mail = nullptr;
        } else {
            memset(r->la_mail, 0, 32 * sizeof(uint32_t));
        }
    }
    auto read_words = [&](const uint32_t *src, uint32_t n, uint32_t *out) -> hipError_t {
        if (r->la_mail && n <= 31u) {
            const uint32_t seq = ++r->la_mail_seq ? r->la_mail_seq : ++r->la_mail_seq; // never 0
            fsk_la_mail(src, n, r->la_mail, seq, s);
            volatile uint32_t *m = r->la_mail;
            for (uint64_t spin = 0; spin < 400000000ull; spin++) { // (seconds; a launch error shows below)
                if (m[31] == seq) {
                    __atomic_thread_fence(__ATOMIC_ACQUIRE);
                    for (uint32_t i = 0; i < n; i++)
                        out[i] = m[i];
                    return hipSuccess;
                }
                if ((spin & 0xFFFFFu) == 0xFFFFFu && hipStreamQuery(s) != hipErrorNotReady)
                    break; // the stream has drained (or failed) without the word arriving: read the plain way
            }
        }
        hipError_t e = hipMemcpyAsync(out, src, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        return e != hipSuccess ? e : hipStreamSynchronize(s);
    };

    // isZCoeffZero of the first step (LAReference.cpp:52-56): word 8 of the scratch words, read back with stage 0's first
    // detection below (one round trip less)
    fsk_la_tail<F>(zref, maxRef, nullptr, d_small + 8, s);

    // one stage: elements 0 .. limit-1 (+ the sentinel element `limit`), period / first record decided by the caller
    bool tail_written = false; // run_chain wrote the stage's tail record with its records
    auto run_chain = [&](bool stage0, const Rec *P, uint32_t limit, uint32_t period, bool have_first, uint32_t first_end,
                         uint32_t first_step, uint32_t x_start, uint32_t &n_records) -> uint32_t {
        const uint32_t nstates = 2u * limit; // limit <= maxRef < 2^31 - 1
        tail_written = false;
        uint32_t offset = 0;
        // room for this stage's first record and its tail record before anything is written
        if ((size_t)la_size + 2u > cap_recs)
            return FS_ERR_7;
        if (have_first) {
            fsk_la_one_record<F>(stage0, zref, P, first_end, first_step, d_table + la_size, s);
            offset = 1;
        }
        n_records = offset;
        if (x_start != kLaTerm && (x_start >> 1) < limit) {
            fsk_la_next<F>(stage0, chebv.p, mm.p, pos.as<uint32_t>(), limit, period, nextA.as<uint32_t>(), reach.as<uint32_t>(),
                           x_start, s); // (also zeroes reach and marks the chain's start)
            // jump tables ping-pong between nextB and nextC; the original next stays in nextA for the record kernel
            if (nstates <= (1u << 16)) {
                // a small stage: every round in one launch (the launches were most of the time on a small orbit)
                uint32_t rounds = 0;
                for (uint32_t span = 1; span < limit + 1u; span <<= 1)
                    rounds++;
                fsk_la_reach_all(nextA.as<uint32_t>(), nextB.as<uint32_t>(), nextC.as<uint32_t>(), reach.as<uint32_t>(), nstates,
                                 rounds, s);
            } else {
                uint32_t *jin = nextA.as<uint32_t>(), *jout = nextB.as<uint32_t>();
                for (uint32_t span = 1; span < limit + 1u; span <<= 1) {
                    fsk_la_reach(jin, jout, reach.as<uint32_t>(), nstates, s);
                    jin = jout;
                    jout = jout == nextB.as<uint32_t>() ? nextC.as<uint32_t>() : nextB.as<uint32_t>();
                }
            }
            fsk_scan_u32(reach.as<uint32_t>(), rank.as<uint32_t>(), nstates, s);
            FS_TRY(read_words(rank.as<uint32_t>() + nstates, 1, h));
            if ((size_t)la_size + offset + h[0] + 2u > cap_recs)
                return FS_ERR_7;
            // (the stage's tail record goes out with the same launch)
            fsk_la_records<F>(stage0, zref, P, pos.as<uint32_t>(), nextA.as<uint32_t>(), reach.as<uint32_t>(),
                              rank.as<uint32_t>(), limit, offset, d_table + la_size, d_table + la_size + offset + h[0], maxRef, s);
            n_records = offset + h[0];
            tail_written = true;
        }
        return (uint32_t)hipGetLastError();
    };

    // ---------------- stage 0: CreateLAFromOrbit, LAReference.cpp:28-210
    bool no_table = false; // CreateLAFromOrbit returned false: the records stay, the table is not valid
    {
        const uint32_t limit = maxRef;
        fsk_la_src_orbit<F>(zref, maxRef + 1u, chebv.p, s);
        fsk_la_first<F>(true, chebv.p, mm.p, limit, d_small, s);
        uint32_t h0[9];
        FS_TRY(read_words(d_small, 9, h0));
        if (h0[8])
            return FS_ERR_UNSUPPORTED; // the first step's ZCoeff is zero
        h[0] = h0[0], h[1] = h0[1];
        uint32_t Period = h[0] == kLaTerm ? 0u : h[0];
        bool have_first = false;
        uint32_t x_start;
        const double NthRoot = std::round(std::log2((double)maxRef) / periodDivisor);
        if (Period == 0 && maxRef <= kLaLowBound) {
            // :135-140: no period in an orbit of at most 64 steps -- one record over the whole orbit and the closing one,
            // CreateLAFromOrbit returns false and the table stays invalid (GenerateApproximationData, :1002-1005)
            no_table = true;
            x_start = kLaTerm;
        } else if (Period == 0 || Period > kLaLowBound) {
            Period = (uint32_t)std::round(std::pow((double)maxRef, 1.0 / NthRoot)); // :128-134 / :141-147
            x_start = 1u;                                                            // (0, flavour 1)
        } else {
            have_first = true; // the record that ended at the first detection stays (:97-101)
            const uint32_t i = Period;
            x_start = i + 1u < maxRef ? 2u * i + 1u : 2u * i; // :105-111: step z[i+1] at once unless that is the end
        }
        stages.push_back(fs_la_stage_u32{0u, 0u});
        uint32_t n = 0;
        // CreateLAFromOrbitMT (:215-770) is what the reference runs when the orbit has two or more 50 000-entry chunks and the
        // host two or more hardware threads (:236-251): the same prologue, then the scan in pieces
        size_t thread_count = maxRef / 50000u;
        if (thread_count > (size_t)(host_threads > 0 ? host_threads : 1))
            thread_count = (size_t)(host_threads > 0 ? host_threads : 1);
        if (no_table) {
            fsk_la_one_record<F>(true, zref, nullptr, maxRef, maxRef, d_table, s);
            n = 1;
            tail_written = false;
        } else if (thread_count > 1) {
            // Every piece of the reference's multi-threaded scan is a stretch of one of the chains x -> next(x) of the
            // single-threaded state machine: the Starter's from the prologue's state, Worker k's from the state its first
            // period detection leaves (two uncapped trackers begun one element apart at maxRef * k / N, :486-560); a piece ends
            // where its scan meets the start the next worker has published (:640-668, :440-470), and Stitch (:711-760) lines the
            // pieces up.  All cross-thread values are futures in the reference, so none of this depends on timing.  On the
            // device: next() for every state and the 2 (N - 1) first detections; the host walks the chains (indices only) and
            // stitches; the device folds the records of the segments that came out.
            const size_t TC = thread_count;
            const uint32_t nstates = 2u * limit;
            tail_written = false;
            if (have_first)
                fsk_la_one_record<F>(true, zref, nullptr, Period, Period, d_table, s);
            const uint32_t offset = have_first ? 1u : 0u;
            fsk_la_next<F>(true, chebv.p, mm.p, pos.as<uint32_t>(), limit, Period, nextA.as<uint32_t>(), reach.as<uint32_t>(), x_start, s);
            std::vector<uint32_t> bases(2u * (TC - 1u)), firsts(2u * (TC - 1u));
            for (size_t k = 1; k < TC; k++) {
                const uint32_t Begin = (uint32_t)((uint64_t)maxRef * k / TC);
                bases[2u * (k - 1u)] = Begin - 1u; // LA: z[Begin-1] stepped with z[Begin], tests from Begin + 1
                bases[2u * (k - 1u) + 1u] = Begin; // LA2: z[Begin] stepped with z[Begin+1], tests from Begin + 2
            }
            uint32_t *d_bases = nextB.as<uint32_t>(), *d_firsts = nextC.as<uint32_t>();
            FS_TRY(hipMemcpyAsync(d_bases, bases.data(), 4u * bases.size(), hipMemcpyHostToDevice, s));
            fsk_la_first_from<F>(chebv.p, d_bases, (uint32_t)bases.size(), limit, d_firsts, s);
            std::vector<uint32_t> hnext(nstates);
            FS_TRY(hipMemcpyAsync(firsts.data(), d_firsts, 4u * firsts.size(), hipMemcpyDeviceToHost, s));
            FS_TRY(hipMemcpyAsync(hnext.data(), nextA.p, 4u * (size_t)nstates, hipMemcpyDeviceToHost, s));
            FS_TRY(hipStreamSynchronize(s));
            FS_TRY(hipGetLastError());

            struct Piece {
                int64_t start = 0, finish = 0;
                std::vector<uint32_t> states; // the records this piece pushed: segment of state x = [x >> 1, next(x) >> 1)
                uint32_t last_b = 0, last_e = 0; // the record it was still accumulating when it stopped
            };
            std::vector<Piece> piece(TC);
            // the main scan of a piece (:392-484 Starter, :600-690 Worker): from state x; once past `end`, each boundary is
            // compared with the published start of the next piece
            auto walk = [&](uint32_t x, uint32_t end, size_t next_thread, Piece &pc) {
                for (;;) {
                    const uint32_t nx = hnext[x];
                    if (nx == kLaTerm) { // the scan ran to the end of the orbit: its open record covers the rest
                        pc.finish = maxRef;
                        pc.last_b = x >> 1, pc.last_e = maxRef;
                        return;
                    }
                    pc.states.push_back(x);
                    x = nx;
                    const uint32_t c = (x >> 1) + (x & 1u); // the scan index when the reference tests `j > End`
                    if (c > end && next_thread < TC) {
                        const int64_t ns = piece[next_thread].start;
                        if ((int64_t)c == ns - 1) { // joined: the open record is what the new state has taken so far
                            pc.finish = (int64_t)c + 1;
                            pc.last_b = x >> 1, pc.last_e = (x >> 1) + (x & 1u) + 1u;
                            return;
                        }
                        if ((int64_t)c >= ns)
                            next_thread++;
                    }
                }
            };
            for (size_t k = TC - 1u; k >= 1u; k--) {
                const uint32_t Begin = (uint32_t)((uint64_t)maxRef * k / TC), End = (uint32_t)((uint64_t)maxRef * (k + 1u) / TC);
                const uint32_t dA = firsts[2u * (k - 1u)], dB = firsts[2u * (k - 1u) + 1u];
                // the loop tests LA at Begin + 1 + t, then LA2 at Begin + 2 + t: the first to fire wins, LA on a tie
                uint32_t d = kLaTerm;
                if (dA != kLaTerm && (dB == kLaTerm || (uint64_t)dA - (Begin + 1u) <= (uint64_t)dB - (Begin + 2u)))
                    d = dA;
                else if (dB != kLaTerm)
                    d = dB;
                uint32_t x = kLaTerm;
                int64_t j = maxRef;
                if (d != kLaTerm) {
                    const uint32_t f = d + 1u < maxRef ? 1u : 0u; // :520-527, :541-549
                    x = 2u * d + f;
                    j = (int64_t)d + 1 + f;
                }
                Piece &pc = piece[k];
                if (k == TC - 1u || (j >= (int64_t)Begin && j < (int64_t)End)) {
                    pc.start = j;
                } else { // no period boundary inside its own chunk: the worker adopts the next one's start and contributes nothing
                    pc.start = piece[k + 1u].start;
                    pc.finish = -1;
                    continue;
                }
                if (x == kLaTerm) { // (last worker, nothing detected: no records, finish == start)
                    pc.finish = maxRef;
                    pc.last_b = Begin - 1u, pc.last_e = maxRef;
                    continue;
                }
                walk(x, End, k + 1u, pc);
            }
            walk(x_start, maxRef / (uint32_t)TC, 1u, piece[0]);

            // Stitch, :711-760
            std::vector<uint32_t> seg;
            auto append = [&](const Piece &pc) {
                for (uint32_t x : pc.states) {
                    seg.push_back(x >> 1);
                    seg.push_back(hnext[x] >> 1);
                }
            };
            append(piece[0]);
            size_t last_to_add = 0, index = 0, jj = 0;
            while (index < TC - 1u && piece[jj].finish > piece[index + 1u].start)
                index++;
            index++;
            for (; index < TC; index++) {
                append(piece[index]);
                if (piece[index].finish > piece[index].start)
                    last_to_add = index;
                jj = index;
                while (index < TC - 1u && piece[jj].finish > piece[index + 1u].start)
                    index++;
            }
            seg.push_back(piece[last_to_add].last_b);
            seg.push_back(piece[last_to_add].last_e);
            const uint32_t nseg = (uint32_t)(seg.size() / 2u);
            if ((size_t)offset + nseg + 2u > cap_recs || seg.size() > cap_states)
                return FS_ERR_7;
            FS_TRY(hipMemcpyAsync(nextB.p, seg.data(), 4u * seg.size(), hipMemcpyHostToDevice, s));
            fsk_la_records_list<F>(zref, nextB.as<uint32_t>(), nseg, d_table + offset, d_table + offset + nseg, maxRef, s);
            FS_TRY(hipStreamSynchronize(s)); // (seg lives on this stack frame)
            FS_TRY(hipGetLastError());
            n = offset + nseg;
            tail_written = true;
        } else if (uint32_t e = run_chain(true, nullptr, limit, Period, have_first, have_first ? Period : 0u,
                                          have_first ? Period : 0u, x_start, n))
            return e;
        stages[0].MacroItCount = n;
        la_size = n;
        if (!tail_written)
            fsk_la_tail<F>(zref, maxRef, d_table + la_size, nullptr, s);
        la_size++;
    }

    // ---------------- higher stages: CreateNewLAStage, LAReference.cpp:774-966
    while (!no_table) {
        const uint32_t PrevStage = (uint32_t)stages.size() - 1u, CurrentStage = (uint32_t)stages.size();
        if (CurrentStage >= kLaMaxStages)
            break;
        const uint32_t PrevIdx = stages[PrevStage].LAIndex, Count = stages[PrevStage].MacroItCount;
        const Rec *P = d_table + PrevIdx;
        fsk_la_src_stage<F>(P, Count + 1u, chebv.p, mm.p, steps.as<uint32_t>(), s);
        // scan of the step lengths, first detection and everything the period decision reads: one launch, one read-back
        // (round 4: three launches and three round trips per stage before)
        fsk_la_stage_prologue<F>(P, chebv.p, mm.p, steps.as<uint32_t>(), pos.as<uint32_t>(), Count, d_small, s);
        uint32_t hs[5];
        FS_TRY(read_words(d_small, 5, hs));
        uint32_t jd = hs[0], fd = hs[1];
        const uint32_t step0 = hs[2];
        uint32_t Period = 0;
        if (jd != kLaTerm) {
            if (hs[4]) // isLAThresholdZero: the prologue breaks without a period (:815-817)
                jd = kLaTerm;
            else
                Period = hs[3];
        }
        stages.push_back(fs_la_stage_u32{la_size, 0u});
        const double NthRoot = std::round(std::log2((double)maxRef) / periodDivisor);
        bool have_first = false, last_stage = false;
        uint32_t x_start = 1u, first_end = 0, first_step = 0;
        if (Period == 0) {
            if ((uint64_t)maxRef > (uint64_t)step0 * kLaLowBound) {
                const double Ratio = ((double)maxRef) / step0;
                Period = step0 * (uint32_t)std::round(std::pow(Ratio, 1.0 / NthRoot)); // :861-869
            } else {
                // :870-881: one record over the whole previous stage, and this is the last stage
                last_stage = true;
                have_first = true;
                first_end = Count;
                first_step = maxRef;
                x_start = kLaTerm;
            }
        } else if ((uint64_t)Period > (uint64_t)step0 * kLaLowBound) {
            const double Ratio = ((double)Period) / step0;
            Period = step0 * ((uint32_t)std::round(std::pow(Ratio, 1.0 / NthRoot))); // :882-893
        } else {
            have_first = true;
            first_end = jd;
            first_step = Period;
            x_start = 2u * jd + fd;
        }
        uint32_t n = 0;
        if (uint32_t e = run_chain(false, P, Count, Period, have_first, first_end, first_step, x_start, n))
            return e;
        stages[CurrentStage].MacroItCount = last_stage ? 1u : n;
        la_size += n;
        if (!tail_written)
            fsk_la_tail<F>(zref, maxRef, d_table + la_size, nullptr, s);
        la_size++;
        if (last_stage)
            break;
    }

    // ---------------- CreateATFromLA + install
    const uint32_t stage_count = (uint32_t)stages.size();
    std::vector<uint32_t> idx(stage_count);
    for (uint32_t k = 0; k < stage_count; k++)
        idx[k] = stages[k].LAIndex;
    FS_TRY(hipMemcpyAsync(stage_idx.p, idx.data(), 4u * stage_count, hipMemcpyHostToDevice, s));
    if (!no_table)
        fsk_la_at<F>(d_table, stage_idx.as<uint32_t>(), stage_count, max_radius, use_small_exponents, atbuf.p, d_small, s);
    else { // (no CreateATFromLA: the ATInfo stays as constructed and is never used)
        FS_TRY(hipMemsetAsync(atbuf.p, 0, sizeof(fs::la::ATInfoT<F>), s));
        FS_TRY(hipMemsetAsync(d_small, 0, 4, s));
    }
    fs::la::ATInfoT<F> at;
    r->la_ok = false;
    const size_t rec_bytes = sizeof(F) == 4 ? sizeof(fs_la_hdr32_u32) : sizeof(fs_la_hdr64_u32);
    FS_TRY(la_reserve(r, rec_bytes * la_size, sizeof(fs_la_stage_u32) * stage_count));
    fsk_la_pack(sizeof(F) == 8, d_table, r->las, la_size, s);
    FS_TRY(hipMemcpyAsync(r->stages, stages.data(), sizeof(fs_la_stage_u32) * stage_count, hipMemcpyHostToDevice, s));
    FS_TRY(hipMemcpyAsync(&at, atbuf.p, sizeof(at), hipMemcpyDeviceToHost, s));
    FS_TRY(hipMemcpyAsync(h, d_small, 4, hipMemcpyDeviceToHost, s));
    FS_TRY(hipStreamSynchronize(s)); // (one round trip for the AT record, its flag, and the host temporaries above)
    FS_TRY(hipGetLastError());
    r->n_las = la_size;
    r->n_stages = stage_count;
    r->la_valid = no_table ? 0 : 1;
    r->use_at = !no_table && h[0] ? 1 : 0;
    memset(&r->at, 0, sizeof(r->at));
    memset(&r->at64, 0, sizeof(r->at64));
    if (!no_table)
        pack_at<F>(at, r);
    r->la_type = sizeof(F) == 4 ? FS_T_HDR32 : FS_T_HDR64;
    r->la_gen = 0;
    r->la_u64 = false; // the table just installed has uint32 fields, whatever an earlier fs_upload_la left behind
    r->at_step_hi = 0;
    r->la_ok = true;
    return 0;
}

} // namespace

extern "C" {

uint32_t fs_build_la(fs_renderer *r, int type_tag, const void *max_radius, int use_small_exponents)
{
    return fs_build_la_mt(r, type_tag, max_radius, use_small_exponents, 1);
}

uint32_t fs_build_la_mt(fs_renderer *r, int type_tag, const void *max_radius, int use_small_exponents, int host_threads)
{
    if (uint32_t e = use_device(r))
        return e;
    if (type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64)
        return FS_ERR_UNSUPPORTED;
    if (!r->compute || !r->orbit_ok || r->orbit_type != type_tag || !max_radius)
        return FS_ERR_6;
    if (r->orbit_seq)
        return FS_ERR_UNSUPPORTED; // needs the expanded orbit (fs_set_compressed_orbit_mode 0)
    TimedLaunch t(r);
    return type_tag == FS_T_HDR32 ? build_la<float>(r, max_radius, use_small_exponents, host_threads)
                                  : build_la<double>(r, max_radius, use_small_exponents, host_threads);
}

uint32_t fs_la_counts(const fs_renderer *r, uint32_t *n_las, uint32_t *n_stages, int *use_at, int *is_valid)
{
    if (!r->la_ok)
        return FS_ERR_6;
    if (n_las)
        *n_las = r->n_las;
    if (n_stages)
        *n_stages = r->n_stages;
    if (use_at)
        *use_at = r->use_at;
    if (is_valid)
        *is_valid = r->la_valid;
    return 0;
}

uint32_t fs_read_la(fs_renderer *r, void *las_out, uint32_t max_las, void *stages_out, uint32_t max_stages, void *at_out)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->la_ok || (r->la_type != FS_T_HDR32 && r->la_type != FS_T_HDR64))
        return FS_ERR_6;
    const size_t rec_bytes = r->la_type == FS_T_HDR32 ? sizeof(fs_la_hdr32_u32) : sizeof(fs_la_hdr64_u32);
    const uint32_t nl = r->n_las < max_las ? r->n_las : max_las, ns = r->n_stages < max_stages ? r->n_stages : max_stages;
    if (las_out && nl)
        FS_TRY(hipMemcpyAsync(las_out, r->las, rec_bytes * nl, hipMemcpyDeviceToHost, r->compute));
    if (stages_out && ns)
        FS_TRY(hipMemcpyAsync(stages_out, r->stages, sizeof(fs_la_stage_u32) * ns, hipMemcpyDeviceToHost, r->compute));
    FS_TRY(hipStreamSynchronize(r->compute));
    if (at_out) {
        if (r->la_type == FS_T_HDR32)
            memcpy(at_out, &r->at, sizeof(r->at));
        else
            memcpy(at_out, &r->at64, sizeof(r->at64));
    }
    return 0;
}

int32_t fs_bla_num_levels(const fs_renderer *r) { return r->bla_n_levels; }
int32_t fs_bla_lm2(const fs_renderer *r) { return r->bla_lm2; }
uint64_t fs_bla_level_size(const fs_renderer *r, int32_t level)
{
    return level >= 0 && (size_t)level < r->bla_level_sizes.size() ? r->bla_level_sizes[(size_t)level] : 0;
}
uint32_t fs_read_bla_level(fs_renderer *r, int32_t level, void *out, uint64_t max_records)
{
    if (uint32_t e = use_device(r))
        return e;
    if (level < 0 || (size_t)level >= r->bla_level_mem.size())
        return FS_ERR_7;
    const size_t rec_bytes = r->bla_type == FS_T_HDR32 ? sizeof(fs_bla_hdr32)
                                                       : (r->bla_type == FS_T_HDR64 ? sizeof(fs_bla_hdr64) : sizeof(fs_bla_f64));
    const uint64_t n = r->bla_level_sizes[(size_t)level] < max_records ? r->bla_level_sizes[(size_t)level] : max_records;
    if (n && r->bla_level_mem[(size_t)level])
        FS_TRY(hipMemcpyAsync(out, r->bla_level_mem[(size_t)level], n * rec_bytes, hipMemcpyDefault, r->compute));
    return (uint32_t)hipStreamSynchronize(r->compute);
}

uint32_t fs_render_bla(fs_renderer *r, int type_tag, const void *coords, uint64_t n_iterations);

// Pixel order for the LAv2 kernels that wait for their slowest lane (see kernels_order.hip).  pix_order_for: the order to launch
// this frame with, or nullptr (first frame of a view, small frames, 64-bit buffers, A/B switch, no memory); pix_order_after: called
// behind the frame's kernel when it ran WITHOUT an order -- sorts the buffer it has just written and keeps the result for the next
// frame with the same key.  Frames of fewer than kPixOrderMinPixels elements are not worth the sort.
constexpr uint64_t kPixOrderMinPixels = 1u << 20;

static fs_renderer::PixKey pix_key_of(fs_renderer *r, const FsFrame &f, int type_tag, int mode, int parity, const void *coords,
                                      size_t coords_bytes, uint64_t n_iterations)
{
    fs_renderer::PixKey k;
    memset(&k, 0, sizeof(k));
    k.rounded_width = f.rounded_width, k.local_rows = f.local_rows, k.band_first = f.band_first, k.band_rows = f.band_rows;
    k.band_stride = f.band_stride, k.type_tag = type_tag, k.mode = mode, k.parity = parity;
    k.orbit_gen = r->orbit_gen, k.orbit_epoch = r->orbit_epoch, k.n_iterations = n_iterations;
    memcpy(k.coords, coords, coords_bytes < sizeof(k.coords) ? coords_bytes : sizeof(k.coords));
    return k;
}

static bool pix_order_wanted(fs_renderer *r, const FsFrame &f)
{
    const uint64_t n = (uint64_t)f.rounded_width * ((f.local_rows + 7u) & ~7u);
    // (FSMI355_STATS_KEEP_ORDER=1: a counting launch keeps the recorded order -- tools/c4_arm_probe.py counts what the ORDERED waves do)
    static const bool stats_keep = [] { const char *e = getenv("FSMI355_STATS_KEEP_ORDER"); return e && e[0] == '1'; }();
    return r->iter_bytes == 4 && f.wide == 0u && (!r->stats_on || stats_keep) && n >= kPixOrderMinPixels && n < 0x7FFFFFFFull &&
           (r->variant & FS_VARIANT_FLAG_NATURAL_ORDER) == 0 && (r->variant & FS_VARIANT_BASE_MASK) == FS_VARIANT_TUNED;
}

static const uint32_t *pix_order_for(fs_renderer *r, const FsFrame &f, const fs_renderer::PixKey &key)
{
    r->last_frame_ordered = false;
    if (!pix_order_wanted(r, f) || !r->pix_valid || !(r->pix_key == key))
        return nullptr;
    r->last_frame_ordered = true;
    return r->pix_order;
}

// An order costs a sort (two for HDRFloat<double>) and is worth it only for a view that is rendered again: a viewer that zooms
// changes the coordinates with every frame and would pay for sorts it never uses.  So the first unordered frame of a key only
// leaves its key behind; the second one records and sorts; the third and later ones run ordered.  Returns whether THIS unordered
// frame is such a second one.
static bool pix_second_sighting(fs_renderer *r, const FsFrame &f, const fs_renderer::PixKey &key)
{
    const bool wanted = pix_order_wanted(r, f);
    const bool again = wanted && r->pix_seen && r->pix_seen_key == key;
    r->pix_seen = wanted;
    r->pix_seen_key = key;
    return again;
}

// The cost record of a frame that runs WITHOUT an order (the first of a view): a zeroed buffer in the iteration buffer's geometry
// that the kernel fills pixel by pixel (padding stays 0 and sorts last), or nullptr (no order wanted, no memory).
static uint32_t *pix_cost_for(fs_renderer *r, const FsFrame &f, bool frame_is_ordered)
{
    if (frame_is_ordered || !pix_order_wanted(r, f))
        return nullptr;
    const size_t n = (size_t)f.rounded_width * ((f.local_rows + 7u) & ~7u);
    if (r->pix_cost_cap < n) {
        (void)r_free(r, r->pix_cost);
        r->pix_cost = nullptr;
        r->pix_cost_cap = 0;
        if (r_alloc(r, (void **)&r->pix_cost, n * sizeof(uint32_t), kFrame) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        r->pix_cost_cap = n;
    }
    if (hipMemsetAsync(r->pix_cost, 0, n * sizeof(uint32_t), r->compute) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return r->pix_cost;
}

// order / work / temp buffers of the pixel sort for n elements; false = no memory (frames keep the tile mapping)
static bool pix_buffers(fs_renderer *r, uint32_t n)
{
    if (r->pix_cap >= n)
        return true;
    (void)r_free(r, r->pix_order);
    (void)r_free(r, r->pix_work);
    (void)r_free(r, r->pix_temp);
    r->pix_order = r->pix_work = nullptr;
    r->pix_temp = nullptr;
    r->pix_cap = 0;
    r->pix_valid = false;
    const size_t tb = fsk_pixel_order_temp_bytes(n);
    if (r_alloc(r, (void **)&r->pix_order, (size_t)n * sizeof(uint32_t), kFrame) != hipSuccess ||
        r_alloc(r, (void **)&r->pix_work, (size_t)n * 2 * sizeof(uint32_t), kFrame) != hipSuccess ||
        r_alloc(r, &r->pix_temp, tb ? tb : 16, kFrame) != hipSuccess) {
        (void)hipGetLastError();
        (void)r_free(r, r->pix_order);
        (void)r_free(r, r->pix_work);
        (void)r_free(r, r->pix_temp);
        r->pix_order = r->pix_work = nullptr;
        r->pix_temp = nullptr;
        return false;
    }
    r->pix_cap = n;
    r->pix_temp_bytes = tb;
    return true;
}

static void pix_order_after(fs_renderer *r, const FsFrame &f, const fs_renderer::PixKey &key, bool frame_was_ordered,
                            const uint32_t *cost = nullptr, int key_bits = 32)
{
    if (frame_was_ordered || !pix_order_wanted(r, f))
        return; // (an ordered frame's buffer equals the one the order was made from: nothing new to learn)
    const uint32_t n = f.rounded_width * ((f.local_rows + 7u) & ~7u);
    r->pix_valid = false;
    if (!pix_buffers(r, n))
        return;
    // sorted by the cost the frame recorded (round 5) -- or, without a record, by the counts as before
    if (fsk_pixel_order_build(cost ? cost : (const uint32_t *)r->iters(), n, r->pix_work, r->pix_order, r->pix_temp, r->pix_temp_bytes,
                              r->compute, key_bits) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    r->pix_key = key;
    r->pix_valid = true;
}

// The tile order of a view's first frame (kernels_tile_sample.hip): S carries the frame, the coordinates and the AT record's values in
// binary64; -> order[n_slots] on the device (wave w of the frame's launch renders tile order[w]), or nullptr: not wanted (small frames,
// 64-bit buffers, A/B switch FSMI355_COLD_TILE_ORDER=0, FS_VARIANT_NATURAL_TILE_ORDER), no memory.  Queued on the compute stream.
static const uint32_t *cold_tile_order(fs_renderer *r, FsTileSampleArgs &S)
{
    static const bool off = [] { const char *e = getenv("FSMI355_COLD_TILE_ORDER"); return e && e[0] == '0'; }();
    if (off || !pix_order_wanted(r, S.frame) || S.StepLength == 0u)
        return nullptr;
    S.tiles_x = (S.frame.width + 7u) / 8u, S.tiles_y = (S.frame.local_rows + 7u) / 8u;
    S.n_slots = ((S.frame.width + 31u) / 32u) * S.tiles_y * 4u; // waves of the frame's launch (tile_grid: 4 tiles per workgroup)
    if (r->cold_cap < S.n_slots) {
        (void)r_free(r, r->cold_cost);
        (void)r_free(r, r->cold_order);
        (void)r_free(r, r->cold_work);
        (void)r_free(r, r->cold_temp);
        r->cold_cost = r->cold_order = r->cold_work = nullptr;
        r->cold_temp = nullptr;
        r->cold_cap = 0;
        const size_t tb = fsk_pixel_order_temp_bytes(S.n_slots);
        if (r_alloc(r, (void **)&r->cold_cost, (size_t)S.n_slots * sizeof(uint32_t), kFrame) != hipSuccess ||
            r_alloc(r, (void **)&r->cold_order, (size_t)S.n_slots * sizeof(uint32_t), kFrame) != hipSuccess ||
            r_alloc(r, (void **)&r->cold_work, (size_t)S.n_slots * 2 * sizeof(uint32_t), kFrame) != hipSuccess ||
            r_alloc(r, &r->cold_temp, tb ? tb : 16, kFrame) != hipSuccess) {
            (void)hipGetLastError();
            (void)r_free(r, r->cold_cost);
            (void)r_free(r, r->cold_order);
            (void)r_free(r, r->cold_work);
            (void)r_free(r, r->cold_temp);
            r->cold_cost = r->cold_order = r->cold_work = nullptr;
            r->cold_temp = nullptr;
            return nullptr;
        }
        r->cold_cap = S.n_slots;
        r->cold_temp_bytes = tb;
    }
    S.cost = r->cold_cost;
    fsk_at_tile_sample64(S, r->compute);
    if (fsk_pixel_order_build(r->cold_cost, S.n_slots, r->cold_work, r->cold_order, r->cold_temp, r->cold_temp_bytes, r->compute, 32) !=
        hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    fsk_tile_order_finish(r->cold_order, S.n_slots, S.tiles_x * S.tiles_y, r->compute);
    r->last_cold_ordered = true;
    return r->cold_order;
}

uint32_t fs_render_lav2(fs_renderer *r, int type_tag, int mode, int parity, const void *coords, uint64_t n_iterations)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized())
        return 0; // GPU_Render.cu:1007-1009
    if (r->local_rows == 0)
        return 0; // this renderer owns no row of the frame (a rank beyond the last band)
    r->last_frame_ordered = false; // (every path below that uses a recorded order says so itself)
    r->lav2_last_ordered = false;
    r->last_cold_ordered = false;
    const bool plain = type_tag == FS_T_F32 || type_tag == FS_T_F64 || type_tag == FS_T_2X32;
    // iteration caps of 2^32 and above need IterType = uint64_t (an 8-byte buffer): every type then runs an instantiation
    // of its kernel that counts in 64 bits (the literal one for HDRFloat<float|double>)
    if (type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64 && type_tag != FS_T_HDR2X32 && !plain)
        return FS_ERR_UNSUPPORTED;
    if (n_iterations > 0xFFFFFFFFull && r->iter_bytes != 8)
        return (uint32_t)hipErrorInvalidValue; // a 4-byte IterType cannot hold such a count
    // (the 64-bit counting kernels can also be forced at small caps: FS_VARIANT_WIDE_COUNTERS, a test switch)
    const bool wide = n_iterations > 0xFFFFFFFFull || (r->variant & FS_VARIANT_FLAG_WIDE) != 0;
    if (!r->orbit_ok || r->orbit_type != type_tag)
        return FS_ERR_6; // GPU_Render.cu:1015-1022
    if (r->orbit_seq && (type_tag == FS_T_HDR32 || type_tag == FS_T_HDR64)) {
        // the orbit is resident as waypoints only (fs_set_compressed_orbit_mode 1): the literal kernel with a sequential
        // decompression cursor per pixel.  Perturbation-only with CPU parity has its twin in the scalar kernel, which reads
        // an expanded orbit: not served in this mode.
        if (mode == FS_LAV2_PO && parity == FS_PARITY_CPU)
            return FS_ERR_UNSUPPORTED;
        if (mode != FS_LAV2_PO && (!r->la_ok || r->la_type != type_tag))
            return FS_ERR_6;
        const int kmode = mode == FS_LAV2_FULL ? FS_MODE_FULL : (mode == FS_LAV2_PO ? FS_MODE_PO : FS_MODE_LAO);
        // 64-bit POSITIONS (and counters) whenever something does not fit 32 bits: the orbit's uncompressed length or period,
        // a table kept in the uint64_t layout -- besides the iteration cap and the test switch
        const bool wide_pos = wide || r->la_u64 || r->orbit_uncompressed > 0xFFFFFFFFull || r->orbit_period > 0xFFFFFFFFull;
        TimedLaunch t(r);
        if (type_tag == FS_T_HDR32) {
            FsLav2ArgsT<float> A;
            fill_lav2<float>(r, A, coords, n_iterations, parity);
            A.frame.wide |= wide_pos ? 1u : 0u;
            r->last_launch_wide = A.frame.wide != 0u;
            A.at = r->at;
            A.wp = r->wp_raw, A.n_wp = (uint32_t)r->orbit_size;
            A.cxLow = r->c_low32[0], A.cyLow = r->c_low32[1];
            fsk_lav2_seq(&A, nullptr, kmode, r->stats_on, r->compute);
        } else {
            FsLav2ArgsT<double> A;
            fill_lav2<double>(r, A, coords, n_iterations, parity);
            A.frame.wide |= wide_pos ? 1u : 0u;
            r->last_launch_wide = A.frame.wide != 0u;
            A.at = r->at64;
            A.wp = r->wp_raw, A.n_wp = (uint32_t)r->orbit_size;
            A.cxLow = r->c_low64[0], A.cyLow = r->c_low64[1];
            fsk_lav2_seq(nullptr, &A, kmode, r->stats_on, r->compute);
        }
        return (uint32_t)hipGetLastError();
    }
    if (r->la_u64 && mode != FS_LAV2_PO)
        return FS_ERR_UNSUPPORTED; // the table is in the uint64_t layout: only the waypoint-resident kernel reads it
    if (wide && (type_tag == FS_T_HDR32 || type_tag == FS_T_HDR64) && !(mode == FS_LAV2_PO && parity == FS_PARITY_CPU)) {
        // GPURenderer::RenderPerturbLAv2<uint64_t, ...> with a cap the 32-bit counters cannot hold: the literal kernel
        // instantiated with 64-bit counters (all three modes; the reference's arithmetic, operation by operation)
        if (mode != FS_LAV2_PO && (!r->la_ok || r->la_type != type_tag))
            return FS_ERR_6;
        const int kmode = mode == FS_LAV2_FULL ? FS_MODE_FULL : (mode == FS_LAV2_PO ? FS_MODE_PO : FS_MODE_LAO);
        TimedLaunch t(r);
        if (type_tag == FS_T_HDR32) {
            FsLav2ArgsT<float> A;
            fill_lav2<float>(r, A, coords, n_iterations, parity);
            A.zref = r->zref;
            A.zq = r->zq;
            A.zs = r->zq + r->zq_n;
        A.zs2 = r->zs2;
        A.zqb = r->zqb;
            A.at = r->at;
            fsk_lav2_wide(&A, nullptr, kmode, r->stats_on, r->compute);
        } else {
            FsLav2ArgsT<double> A;
            fill_lav2<double>(r, A, coords, n_iterations, parity);
            A.zref = r->zref64;
            A.at = r->at64;
            fsk_lav2_wide(nullptr, &A, kmode, r->stats_on, r->compute);
        }
        return (uint32_t)hipGetLastError();
    }
    if (plain) {
        // Gpu1x32 / Gpu1x64 / Gpu2x32 PerturbedLAv2*: no CPU RenderAlgorithm exists for LAv2 on a plain type, the kernel
        // restates the reference's CUDA kernel and ignores `parity`.  coords = float[4] / double[4] / fs_real_p2x32[4].
        if (mode != FS_LAV2_PO && (!r->la_ok || r->la_type != type_tag))
            return FS_ERR_6;
        FsLav2ArgsPlain A;
        memset(&A, 0, sizeof(A));
        A.out = (uint32_t *)r->iters();
        A.orbit = type_tag == FS_T_F64 ? (const void *)r->orbit_f64 : (const void *)r->orbit_plain;
        A.las = r->las;
        A.stages = r->stages;
        A.stats = r->stats;
        A.frame = make_frame(r);
        memcpy(A.coords, coords, type_tag == FS_T_F32 ? 4 * sizeof(float) : 4 * sizeof(double));
        memcpy(A.at, r->at_plain, sizeof(A.at));
        A.orbit_count = (uint32_t)r->orbit_uncompressed;
        A.stage_count = r->n_stages;
        A.n_iterations = (uint32_t)n_iterations;
        A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        // (the waypoint-resident instantiations are built without the step counters too: fs_read_step_count must refuse, not
        // report zeros)
        r->last_launch_wide = A.frame.wide != 0u || r->orbit_seq;
        A.la_valid = (r->la_ok && r->la_type == type_tag) ? r->la_valid : 0;
        A.use_at = r->use_at;
        if (r->orbit_seq) { // waypoint-resident orbit: a cursor per pixel (k_lav2_plain<.., kSeq>)
            A.orbit = nullptr;
            A.wp = r->wp_raw;
            A.n_wp = (uint32_t)r->orbit_size;
            memcpy(A.c_low[0], r->c_low_plain[0], 8);
            memcpy(A.c_low[1], r->c_low_plain[1], 8);
        }
        TimedLaunch t(r);
        fsk_lav2_plain(A, type_tag == FS_T_F32 ? 0 : (type_tag == FS_T_F64 ? 1 : 2),
                       mode == FS_LAV2_FULL ? FS_MODE_FULL : (mode == FS_LAV2_PO ? FS_MODE_PO : FS_MODE_LAO), r->stats_on,
                       r->compute);
        return (uint32_t)hipGetLastError();
    }
    if (type_tag == FS_T_HDR2X32) {
        // No CPU RenderAlgorithm exists for this type: the kernel restates the reference's CUDA kernel and ignores
        // `parity` (coords are fs_real_2x32[4]).
        if (mode != FS_LAV2_PO && (!r->la_ok || r->la_type != type_tag))
            return FS_ERR_6;
        FsLav2Args2x32 A;
        memset(&A, 0, sizeof(A));
        A.out = (uint32_t *)r->iters();
        A.orbit = r->orbit_2x32;
        A.las = (const fs_la_2x32_u32 *)r->las;
        A.stages = r->stages;
        A.stats = r->stats;
        A.frame = make_frame(r);
        memcpy(A.coords, coords, sizeof(A.coords));
        A.at = r->at2x32;
        A.orbit_count = (uint32_t)r->orbit_uncompressed;
        A.stage_count = r->n_stages;
        A.n_iterations = (uint32_t)n_iterations;
        A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u || r->orbit_seq; // (kSeq: no counters either)
        A.la_valid = (r->la_ok && r->la_type == type_tag) ? r->la_valid : 0;
        A.use_at = r->use_at;
        if (r->orbit_seq) { // waypoint-resident orbit: a cursor per pixel (k_lav2_2x32<.., kSeq>)
            A.orbit = nullptr;
            A.wp = (const fs_orbit_2x32_rc *)r->wp_raw;
            A.n_wp = (uint32_t)r->orbit_size;
            memcpy(&A.cxLow, r->c_low_plain[0], sizeof(fs_real_2x32));
            memcpy(&A.cyLow, r->c_low_plain[1], sizeof(fs_real_2x32));
        }
        const fs_renderer::PixKey pk = pix_key_of(r, A.frame, type_tag, mode, 0, coords, sizeof(A.coords), n_iterations);
        A.pixel_order = r->orbit_seq ? nullptr : pix_order_for(r, A.frame, pk);
        const bool second = !r->orbit_seq && A.pixel_order == nullptr && pix_second_sighting(r, A.frame, pk);
        A.pixel_cost = second ? pix_cost_for(r, A.frame, false) : nullptr;
        {
            TimedLaunch t(r);
            if (!r->orbit_seq && A.pixel_order == nullptr && !second && mode != FS_LAV2_PO && A.use_at && A.la_valid) {
                // a view's first frame: tiles in the order of a sampled PerformAT count (the record's values in binary64: head + tail, exact)
                auto R = [](const fs_real_2x32 &x) { return fs::hreal<double>{(double)x.head + (double)x.tail, x.e}; };
                auto Cx = [](const fs_cplx_2x32 &c) {
                    return fs::hcplx<double>{(double)c.re_head + (double)c.re_tail, (double)c.im_head + (double)c.im_tail, c.e};
                };
                FsTileSampleArgs S;
                memset(&S, 0, sizeof(S));
                S.frame = A.frame;
                S.coords = FsCoordsT<double>{R(A.coords[0]), R(A.coords[1]), R(A.coords[2]), R(A.coords[3])};
                S.ThresholdC = R(A.at.ThresholdC), S.SqrEscapeRadius = R(A.at.SqrEscapeRadius);
                S.RefC = Cx(A.at.RefC), S.CCoeff = Cx(A.at.CCoeff);
                S.StepLength = A.at.StepLength, S.n_iterations = A.n_iterations;
                A.tile_order = cold_tile_order(r, S);
                A.tiles_x = S.tiles_x;
            }
            fsk_lav2_2x32(A, mode == FS_LAV2_FULL ? FS_MODE_FULL : (mode == FS_LAV2_PO ? FS_MODE_PO : FS_MODE_LAO),
                          r->stats_on, r->compute);
        }
        if (second)
            pix_order_after(r, A.frame, pk, false, A.pixel_cost);
        return (uint32_t)hipGetLastError();
    }
    if (mode == FS_LAV2_PO && parity == FS_PARITY_CPU) {
        // No dispatched CPU RenderAlgorithm is perturbation-only in HDRFloatComplex arithmetic; the CPU parity
        // target for PO is the single-step branch of CalcCpuPerturbationFractalBLA.
        const int32_t saved = r->bla_n_levels;
        r->bla_n_levels = 0;
        const uint32_t e = fs_render_bla(r, type_tag, coords, n_iterations);
        r->bla_n_levels = saved;
        return e;
    }
    if (mode != FS_LAV2_PO && (!r->la_ok || r->la_type != type_tag))
        return FS_ERR_6;
    const int kmode = mode == FS_LAV2_FULL ? FS_MODE_FULL : (mode == FS_LAV2_PO ? FS_MODE_PO : FS_MODE_LAO);
    if (type_tag == FS_T_HDR32) {
        FsLav2ArgsT<float> A;
        fill_lav2<float>(r, A, coords, n_iterations, parity);
        A.zref = r->zref;
        A.zq = r->zq;
        A.zs = r->zq + r->zq_n;
        A.zs2 = r->zs2;
        A.zqb = r->zqb;
        A.at = r->at;
        // Longest tiles first, self-recorded.  Every frame of the tuned kernel stores one cost word per 8 x 8 tile (its
        // longest lane's step count); the NEXT frame of the same geometry, band layout and orbit generation is launched in
        // descending cost order (64 classes, raster order inside a class).  A frame ends one long wave after its last wave
        // was dispatched and the waves differ 2.5x in length, so the drain at the end of the launch shrinks from the longest
        // wave's duration towards the shortest's.  Which wave renders which tile changes no pixel; the first frame (and
        // every frame after fs_forget_tile_costs, or with FS_VARIANT_NATURAL_TILE_ORDER) runs in natural order.
        const uint32_t tiles_x = (r->width + 7u) / 8u, tiles_y = (r->local_rows + 7u) / 8u;
        const uint32_t n_tiles = tiles_x * tiles_y;
        const uint32_t n_slots = fsk_lav2_hdr32_slots(A.frame);
        const bool tuned = (r->variant & FS_VARIANT_BASE_MASK) != FS_VARIANT_LITERAL;
        const bool record = tuned && n_slots != 0u && n_tiles >= kLav2OrderMinTiles &&
                            (r->variant & FS_VARIANT_FLAG_NATURAL_ORDER) == 0;
        r->last_frame_ordered = false;
        if (record) {
            if (r->lav2_cost_cap < n_tiles) {
                (void)r_free(r, r->lav2_cost);
                (void)r_free(r, r->lav2_sort_tmp);
                r->lav2_cost = r->lav2_sort_tmp = nullptr;
                r->lav2_cost_cap = 0;
                r->lav2_cost_valid = false;
                FS_TRY(r_alloc(r, (void **)&r->lav2_cost, (size_t)n_tiles * sizeof(uint32_t), kFrame));
                FS_TRY(r_alloc(r, (void **)&r->lav2_sort_tmp, (size_t)fsk_tile_order_work_words(n_tiles) * sizeof(uint32_t), kFrame));
                r->lav2_cost_cap = n_tiles;
            }
            if (r->lav2_order_cap < n_slots) {
                (void)r_free(r, r->lav2_order);
                r->lav2_order = nullptr;
                r->lav2_order_cap = 0;
                FS_TRY(r_alloc(r, (void **)&r->lav2_order, ((size_t)n_slots + 1) * sizeof(uint32_t), kFrame));
                r->lav2_order_cap = n_slots;
            }
            const fs_renderer::CostKey key{r->width, r->local_rows, A.frame.band_first, A.frame.band_rows,
                                           A.frame.band_stride, r->orbit_gen};
            A.tile_cost = r->lav2_cost;
            A.tiles_x = tiles_x;
            if (r->lav2_cost_valid && r->lav2_cost_key == key)
                A.tile_order = r->lav2_order;
            r->lav2_cost_key = key;
        }
        if (A.tile_order) {
            fsk_tile_order_by_cost(r->lav2_cost, n_tiles, r->lav2_sort_tmp, r->lav2_order, n_slots, r->compute);
            r->last_frame_ordered = true;
            r->lav2_last_ordered = true;
        }
        TimedLaunch t(r);
        fsk_lav2_hdr32(A, kmode, r->stats_on, r->variant, r->compute);
        r->lav2_cost_valid = record;
    } else {
        FsLav2ArgsT<double> A;
        fill_lav2<double>(r, A, coords, n_iterations, parity);
        A.zref = r->zref64;
        A.at = r->at64;
        const fs_renderer::PixKey pk = pix_key_of(r, A.frame, type_tag, mode, parity, coords, 4 * sizeof(fs_real_hdr64), n_iterations);
        A.pixel_order = pix_order_for(r, A.frame, pk);
        // PerformAT in a pass of its own, in the order of the AT iterations every pixel needs by itself (recorded by the view's
        // first frame): the AT loop reads no memory, so its waves can be made of pixels from anywhere -- equal work per wave --
        // while the frame's kernel keeps the order that keeps neighbours together (below).
        static const bool at_split_off = [] { const char *e = getenv("FSMI355_AT_IN_KERNEL"); return e && e[0] == '1'; }();
        // (A/B, off: FSMI355_AT_SPLIT_COLD=1 runs the pass of its own in the first frame of a view too -- natural order, nothing
        // recorded.  Measured in round 6: AT pass 13.5 ms + the frame's kernel in the tile mapping 48.7 = 62.3 ms against 55.9 ms for the
        // ONE kernel that iterates PerformAT itself: without an order the pass's waves wait for their slowest pixel just as the
        // kernel's do, and the frame's kernel gains nothing from lanes that arrive together)
        static const bool at_split_cold = [] { const char *e = getenv("FSMI355_AT_SPLIT_COLD"); return e && e[0] == '1'; }();
        const bool second = A.pixel_order == nullptr && pix_second_sighting(r, A.frame, pk);
        bool at_split = !at_split_off && mode != FS_LAV2_PO && A.use_at && A.la_valid && pix_order_wanted(r, A.frame) &&
                        (at_split_cold || A.pixel_order != nullptr || second);
        if (at_split) {
            const size_t n = (size_t)A.frame.rounded_width * ((A.frame.local_rows + 7u) & ~7u);
            if (r->at_cap < n) {
                (void)r_free(r, r->at_res);
                (void)r_free(r, r->at_cost);
                (void)r_free(r, r->at_order);
                r->at_res = nullptr;
                r->at_cost = r->at_order = nullptr;
                r->at_cap = 0;
                r->at_order_valid = false;
                if (r_alloc(r, (void **)&r->at_res, n * sizeof(FsAtRes), kFrame) != hipSuccess ||
                    r_alloc(r, (void **)&r->at_cost, n * sizeof(uint32_t), kFrame) != hipSuccess ||
                    r_alloc(r, (void **)&r->at_order, n * sizeof(uint32_t), kFrame) != hipSuccess) {
                    (void)hipGetLastError(); // no memory for it: PerformAT stays inside the frame's kernel
                    (void)r_free(r, r->at_res);
                    (void)r_free(r, r->at_cost);
                    (void)r_free(r, r->at_order);
                    r->at_res = nullptr;
                    r->at_cost = r->at_order = nullptr;
                    at_split = false;
                } else {
                    r->at_cap = n;
                }
            }
        }
        // (the AT order has a key of its own: it is a permutation of the buffer it was recorded on, and pix_order can be rebuilt
        // -- other row bands, a table without AT in between -- without it)
        const bool at_warm = at_split && r->at_order_valid && r->at_key == pk;
        const bool at_record = at_split && !at_warm && (second || A.pixel_order != nullptr);
        // (A/B, off: FSMI355_C4_INFRAME_ORDER=1 sorts the first frame of a view by its own AT iteration counts -- measured in round 6:
        // the frame's kernel then takes 54 ms against 49.6 in the tile mapping and 35 in the order of the previous frame's COUNTS;
        // what the count order knows and the AT count does not is how long a pixel's LA and perturbation phases are)
        static const bool inframe_on = [] { const char *e = getenv("FSMI355_C4_INFRAME_ORDER"); return e && e[0] == '1'; }();
        const bool inframe = inframe_on && at_split && A.pixel_order == nullptr;
        bool inframe_done = false;
        // (sorted by COUNT, not by a recorded cost as the 2x32 frames are: this kernel's steps are cheap enough for the loads of
        // a wave whose lanes are scattered over the frame to cost more than the idle lanes they save -- 81 ms with the cost as
        // the key, 68 with its binades, 53 with the counts, which keep the pixels inside the set side by side: DESIGN.md 7)
        // The order's key (A/B, FSMI355_C4_ORDER_KEY=cost): the previous frame's COUNTS (default), or what the pixels cost the frame's
        // kernel in that frame -- LA steps + perturbation steps, recorded by the second frame of the view
        static const bool key_cost = [] { const char *e = getenv("FSMI355_C4_ORDER_KEY"); return e && e[0] == 'c' && e[1] == 'o' && e[2] == 's'; }();
        uint32_t *cost_key = nullptr;
        if (key_cost && second && at_split) {
            cost_key = pix_cost_for(r, A.frame, false);
            A.pixel_cost = cost_key;
        }
        {
            TimedLaunch t(r);
            if (at_split) {
                const uint32_t n = A.frame.rounded_width * ((A.frame.local_rows + 7u) & ~7u);
                FsLav2ArgsT<double> P = A;
                P.at_res = r->at_res;
                if (at_warm) {
                    P.pixel_order = r->at_order;
                } else {
                    P.pixel_order = nullptr;
                    if (at_record) { // (a view's first frame records nothing: a viewer that zooms never uses it)
                        P.at_cost = r->at_cost;
                        FS_TRY(hipMemsetAsync(r->at_cost, 0, (size_t)n * sizeof(uint32_t), r->compute));
                    }
                    r->at_order_valid = false;
                }
                // In-frame order (round 6): a frame without an order makes its own from the AT pass it has just run -- the AT
                // iteration count is the leading part of a pixel's final count (count = AT iterations x step length + LA steps +
                // perturbation steps), so sorting by it groups the pixels as the previous frame's counts would, with nothing
                // needed from an earlier frame.  key_bits: ATMaxIt bounds the key.
                uint32_t *key = inframe && pix_buffers(r, n) ? pix_cost_for(r, A.frame, false) : nullptr;
                P.pixel_cost = key;
                fsk_at_pass64(P, r->compute);
                t.mid();
                A.at_res = r->at_res;
                if (key) {
                    uint64_t at_max = A.at.StepLength ? n_iterations / A.at.StepLength : 0;
                    int bits = 1;
                    while (bits < 32 && (at_max >> bits) != 0)
                        bits++;
                    if (fsk_pixel_order_build(key, n, r->pix_work, r->pix_order, r->pix_temp, r->pix_temp_bytes, r->compute,
                                              bits) == hipSuccess) {
                        r->pix_key = pk;
                        r->pix_valid = true;
                        A.pixel_order = r->pix_order;
                        inframe_done = true;
                        r->last_frame_ordered = true;
                    } else {
                        (void)hipGetLastError();
                    }
                }
            }
            if (A.pixel_order == nullptr && !second && !at_split && mode != FS_LAV2_PO && A.use_at && A.la_valid) {
                // a view's first frame: tiles in the order of a sampled PerformAT count (kernels_tile_sample.hip)
                FsTileSampleArgs S;
                memset(&S, 0, sizeof(S));
                S.frame = A.frame;
                S.coords = A.coords;
                S.ThresholdC = fs::hreal<double>{A.at.ThresholdC.m, A.at.ThresholdC.e};
                S.SqrEscapeRadius = fs::hreal<double>{A.at.SqrEscapeRadius.m, A.at.SqrEscapeRadius.e};
                S.RefC = fs::hcplx<double>{A.at.RefC.re, A.at.RefC.im, A.at.RefC.e};
                S.CCoeff = fs::hcplx<double>{A.at.CCoeff.re, A.at.CCoeff.im, A.at.CCoeff.e};
                S.StepLength = A.at.StepLength, S.n_iterations = A.n_iterations;
                A.tile_order = cold_tile_order(r, S);
                A.tiles_x = S.tiles_x;
            }
            // the production kernel (kernels_hdr64.hip); FS_VARIANT_LITERAL keeps the operation-by-operation one for A/B
            // (FSMI355_HDR64_LITERAL=1: the literal kernel with the same orders and the same AT pass -- the A/B of the kernel alone)
            static const bool lit_env = [] { const char *e = getenv("FSMI355_HDR64_LITERAL"); return e && e[0] == '1'; }();
            // (k_lav2_hdr64 addresses its records with 32-bit byte offsets: an orbit or a table of 4 GB and more stays with the literal kernel)
            const bool small = (uint64_t)A.orbit_count * sizeof(FsZ64) < 0xFFFFFF00ull &&
                               (uint64_t)r->n_las * sizeof(fs_la_hdr64_u32) < 0xFFFFFF00ull;
            if (lit_env || !small || (r->variant & FS_VARIANT_BASE_MASK) == FS_VARIANT_LITERAL)
                fsk_lav2_hdr64(A, kmode, r->stats_on, r->compute);
            else
                fsk_lav2_hdr64_fast(A, kmode, r->stats_on, r->compute);
        }
        if (second && !inframe_done)
            pix_order_after(r, A.frame, pk, false, cost_key, cost_key ? 16 : 32);
        const uint32_t n_buf = A.frame.rounded_width * ((A.frame.local_rows + 7u) & ~7u);
        if (at_split && at_record && r->pix_valid && r->pix_work && r->pix_temp && r->pix_cap >= n_buf) {
            // the AT pass's own order, from the costs it has just recorded (the sort's work memory is the pixel order's)
            const uint32_t n = n_buf;
            if (fsk_pixel_order_build(r->at_cost, n, r->pix_work, r->at_order, r->pix_temp, r->pix_temp_bytes, r->compute) == hipSuccess) {
                r->at_order_valid = true;
                r->at_key = pk;
            } else
                (void)hipGetLastError();
        }
    }
    return (uint32_t)hipGetLastError();
}

uint32_t fs_render_bla(fs_renderer *r, int type_tag, const void *coords, uint64_t n_iterations)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized())
        return 0;
    if (r->local_rows == 0)
        return 0; // this renderer owns no row of the frame (a rank beyond the last band)
    if (type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64 && type_tag != FS_T_F64)
        return FS_ERR_UNSUPPORTED;
    if (n_iterations > 0xFFFFFFFFull && r->iter_bytes != 8)
        return (uint32_t)hipErrorInvalidValue; // a 4-byte IterType cannot hold such a count
    if (!r->orbit_ok || r->orbit_type != type_tag)
        return FS_ERR_6;
    if (r->orbit_seq)
        return FS_ERR_UNSUPPORTED; // needs the expanded orbit (fs_set_compressed_orbit_mode 0)
    const bool use_bla = r->bla_n_levels > 2 && r->bla_levels_dev != nullptr && r->bla_type == type_tag;
    if (type_tag == FS_T_F64) {
        const double *c = (const double *)coords;
        FsBlaArgsF64 A;
        memset(&A, 0, sizeof(A));
        A.out = (uint32_t *)r->iters();
        A.orbit = r->orbit_f64;
        A.levels = (const fs_bla_f64 *const *)r->bla_levels_dev;
        A.stats = r->stats;
        A.frame = make_frame(r);
        A.dx = c[0];
        A.dy = c[1];
        A.centerX = c[2];
        A.centerY = c[3];
        A.orbit_count = (uint32_t)r->orbit_uncompressed;
        A.n_iterations = (uint32_t)n_iterations;
        A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
        A.lm2 = r->bla_lm2;
        TimedLaunch t(r);
        fsk_perturb_bla_f64(A, use_bla, r->stats_on, r->compute);
    } else if (type_tag == FS_T_HDR32) {
        FsBlaArgsT<float> A;
        memset(&A, 0, sizeof(A));
        A.out = (uint32_t *)r->iters();
        A.zref = r->zref;
        A.zq = r->zq;
        A.zs = r->zq + r->zq_n;
        A.zs2 = r->zs2;
        A.zqb = r->zqb;
        A.levels = (const fs_bla_hdr32 *const *)r->bla_levels_dev;
        A.stats = r->stats;
        A.queue = r->queue;
        A.frame = make_frame(r);
        fill_coords(A.coords, coords);
        A.orbit_count = (uint32_t)r->orbit_uncompressed;
        A.n_iterations = (uint32_t)n_iterations;
        A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
        A.lm2 = r->bla_lm2;
        if (use_bla && r->bla_native_stale)
            if (uint32_t e = bla_make_native(r, r->bla_n_levels))
                return e;
        if (use_bla && r->bla_native_ok) {
            A.nrec = (const FsBlaRec *)((const char *)r->bla_native + 256);
            A.nlad = (const int4 *)((const char *)A.nrec + (size_t)r->bla_native_total * sizeof(FsBlaRec));
            A.nkmax = (const long long *)(A.nlad + 2 * (size_t)r->bla_native_total);
            memcpy(A.level_off, r->bla_level_off, sizeof(A.level_off));
            if (r->bla_heap_ok) {
                A.hrec = (const FsBlaRec *)r->bla_heap;
                A.hlad = (const int4 *)(A.hrec + r->bla_heap_positions);
                A.hq = A.hlad + 2 * (size_t)r->bla_heap_positions;
                A.zb = (const float4 *)(A.hq + 3 * (size_t)r->bla_heap_nq);
            }
        }
        // Long tiles first.  A perturbation-only frame with a high iteration limit is bounded by the few waves that hold
        // never-escaping pixels: each runs its millions of steps at the pace of a wave that is alone on its SIMD, and the
        // frame ends that long after the LAST of them was dispatched -- later still where two of them share a SIMD.  A
        // probe launch runs the centre pixel of every 8 x 8 tile for n_iterations / 32 steps (one lane per tile), the
        // tiles whose centre (or a neighbour's) is still running then are launched first -- one per SIMD while there are
        // no more of them than SIMDs -- the rest in their natural order.  Which wave renders which tile changes no pixel.
        const uint32_t tiles_x = (r->width + 7u) / 8u, tiles_y = (r->local_rows + 7u) / 8u;
        const uint32_t n_slots = ((tiles_x + 3u) / 4u) * 4u * tiles_y; // waves of the launch (tile_grid: 4 tiles per block)
        const bool reorder = !use_bla && !r->stats_on && A.frame.wide == 0u && n_iterations >= kTileOrderMinIterations &&
                             n_slots >= kTileOrderMinTiles && (r->variant & FS_VARIANT_FLAG_NATURAL_ORDER) == 0 &&
                             (r->variant & FS_VARIANT_BASE_MASK) == FS_VARIANT_TUNED;
        if (reorder) {
            if (r->tile_probe_cap < (size_t)tiles_x * tiles_y) {
                if (r->tile_probe)
                    FS_TRY(r_free(r, r->tile_probe));
                r->tile_probe = nullptr;
                r->tile_probe_cap = 0;
                FS_TRY(r_alloc(r, (void **)&r->tile_probe, (size_t)tiles_x * tiles_y * sizeof(uint32_t), kFrame));
                r->tile_probe_cap = (size_t)tiles_x * tiles_y;
            }
            if (r->tile_order_cap < n_slots) {
                if (r->tile_order)
                    FS_TRY(r_free(r, r->tile_order));
                r->tile_order = nullptr;
                r->tile_order_cap = 0;
                FS_TRY(r_alloc(r, (void **)&r->tile_order, ((size_t)n_slots + 1) * sizeof(uint32_t), kFrame));
                r->tile_order_cap = n_slots;
                r->po_order_valid = false;
            }
        }
        TimedLaunch t(r);
        r->last_frame_ordered = false;
        if (reorder) {
            // the order in r->tile_order is the probe's answer for exactly these inputs: a repeated frame (a viewer redraws a
            // view; every bench step) reuses it and the probe launch is skipped
            const fs_renderer::CostKey key{r->width, r->local_rows, A.frame.band_first, A.frame.band_rows,
                                           A.frame.band_stride, r->orbit_gen};
            const bool warm = r->po_order_valid && r->po_order_key == key && r->po_order_epoch == r->orbit_epoch &&
                              r->po_order_iterations == n_iterations && memcmp(r->po_order_coords, coords, 32) == 0;
            if (!warm) {
                FsBlaArgsT<float> P = A;
                P.probe_out = r->tile_probe;
                P.probe_pitch = tiles_x;
                P.n_iterations = (uint32_t)(n_iterations / kTileProbeDivisor);
                fsk_perturb_scalar_hdr32(P, use_bla, false, r->variant, r->compute);
                fsk_tile_order(r->tile_probe, tiles_x, tiles_x, tiles_y, P.n_iterations, r->tile_order, n_slots, r->compute);
                r->po_order_key = key;
                r->po_order_epoch = r->orbit_epoch;
                r->po_order_iterations = n_iterations;
                memcpy(r->po_order_coords, coords, 32);
                r->po_order_valid = true;
            }
            r->last_frame_ordered = warm;
            A.tile_order = r->tile_order;
        }
        fsk_perturb_scalar_hdr32(A, use_bla, r->stats_on, r->variant, r->compute);
    } else {
        FsBlaArgsT<double> A;
        memset(&A, 0, sizeof(A));
        A.out = (uint32_t *)r->iters();
        A.zref = r->zref64;
        A.levels = (const fs_bla_hdr64 *const *)r->bla_levels_dev;
        A.stats = r->stats;
        A.queue = r->queue;
        A.frame = make_frame(r);
        fill_coords(A.coords, coords);
        A.orbit_count = (uint32_t)r->orbit_uncompressed;
        A.n_iterations = (uint32_t)n_iterations;
        A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
        A.lm2 = r->bla_lm2;
        TimedLaunch t(r);
        fsk_perturb_scalar_hdr64(A, use_bla, r->stats_on, r->variant, r->compute);
    }
    return (uint32_t)hipGetLastError();
}

uint32_t fs_render_direct(fs_renderer *r, int type_tag, const void *coords, uint64_t n_iterations)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized())
        return 0; // GPU_Render.cu:626-628
    if (r->local_rows == 0)
        return 0; // this renderer owns no row of the frame (a rank beyond the last band)
    if (type_tag != FS_T_F64 && type_tag != FS_T_HDR32 && type_tag != FS_T_HDR64)
        return FS_ERR_UNSUPPORTED;
    if (n_iterations > 0xFFFFFFFFull && r->iter_bytes != 8)
        return (uint32_t)hipErrorInvalidValue;
    if (r->cx_row_cap < r->width) {
        if (r->cx_row)
            FS_TRY(r_free(r, r->cx_row));
        r->cx_row = nullptr;
        FS_TRY(r_alloc(r, &r->cx_row, (size_t)16 * r->width, kFrame));
        r->cx_row_cap = r->width;
    }
    if (type_tag == FS_T_F64) {
        const double *c = (const double *)coords;
        FsDirectArgs64 A;
        memset(&A, 0, sizeof(A));
        A.out = (uint32_t *)r->iters();
        A.cx_row = (double *)r->cx_row;
        A.stats = r->stats;
        A.frame = make_frame(r);
        A.dy = c[1];
        A.maxY = c[3];
        A.n_iterations = (uint32_t)n_iterations;
        A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
        TimedLaunch t(r);
        fsk_direct_f64(A, c[2], c[0], r->stats_on, r->compute);
    } else if (type_tag == FS_T_HDR32) {
        const fs_real_hdr32 *c = (const fs_real_hdr32 *)coords;
        FsDirectHdrArgsT<float> A;
        memset(&A, 0, sizeof(A));
        A.out = (uint32_t *)r->iters();
        A.cx_row = (fs::hreal<float> *)r->cx_row;
        A.stats = r->stats;
        A.frame = make_frame(r);
        A.dy = fs::hreal32{c[1].m, c[1].e};
        A.maxY = fs::hreal32{c[3].m, c[3].e};
        A.n_iterations = (uint32_t)n_iterations;
        A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
        TimedLaunch t(r);
        fsk_direct_hdr32(A, fs::hreal32{c[2].m, c[2].e}, fs::hreal32{c[0].m, c[0].e}, r->stats_on, r->compute);
    } else {
        const fs_real_hdr64 *c = (const fs_real_hdr64 *)coords;
        FsDirectHdrArgsT<double> A;
        memset(&A, 0, sizeof(A));
        A.out = (uint32_t *)r->iters();
        A.cx_row = (fs::hreal<double> *)r->cx_row;
        A.stats = r->stats;
        A.frame = make_frame(r);
        A.dy = fs::hreal64{c[1].m, c[1].e};
        A.maxY = fs::hreal64{c[3].m, c[3].e};
        A.n_iterations = (uint32_t)n_iterations;
        A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
        TimedLaunch t(r);
        fsk_direct_hdr64(A, fs::hreal64{c[2].m, c[2].e}, fs::hreal64{c[0].m, c[0].e}, r->stats_on, r->compute);
    }
    return (uint32_t)hipGetLastError();
}

uint32_t fs_upload_orbit_scaled(fs_renderer *r, int type_tag, uint32_t iter_bytes, const void *entries_t,
                                const void *entries_f32, uint64_t orbit_size, uint64_t period_maybe_zero)
{
    (void)period_maybe_zero;
    if (uint32_t e = use_device(r))
        return e;
    if ((type_tag != FS_T_HDR32 && type_tag != FS_T_F64) || (iter_bytes != 4 && iter_bytes != 8) ||
        orbit_size > 0xFFFFFFFFull || orbit_size < 2)
        return FS_ERR_UNSUPPORTED;
    if (!r->compute)
        return FS_ERR_6;
    if (r->scaled_t) {
        FS_TRY(r_free(r, r->scaled_t));
        r->scaled_t = nullptr;
    }
    if (r->scaled_f) {
        FS_TRY(r_free(r, r->scaled_f));
        r->scaled_f = nullptr;
    }
    r->scaled_count = 0;
    const size_t t_bytes = type_tag == FS_T_HDR32 ? sizeof(fs_orbit_hdr32_bad) : sizeof(fs_orbit_f64_bad);
    FS_TRY(r_alloc(r, &r->scaled_t, orbit_size * t_bytes, kInput));
    // (the tuned kernel requests its binary32 entries four steps ahead: up to three entries past the end are read, never used)
    FS_TRY(r_alloc(r, (void **)&r->scaled_f, (orbit_size + 8) * sizeof(fs_orbit_f32_bad), kInput));
    FS_TRY(hipMemsetAsync(r->scaled_f + orbit_size, 0, 8 * sizeof(fs_orbit_f32_bad), r->compute));
    FS_TRY(hipMemcpyAsync(r->scaled_t, entries_t, orbit_size * t_bytes, hipMemcpyDefault, r->compute));
    FS_TRY(hipMemcpyAsync(r->scaled_f, entries_f32, orbit_size * sizeof(fs_orbit_f32_bad), hipMemcpyDefault, r->compute));
    fsk_scaled_bounds(r->scaled_f, orbit_size, r->compute); // the tuned kernel's per-entry bound, in the padding word
    FS_TRY(hipStreamSynchronize(r->compute)); // host buffers are borrowed for the call only
    r->scaled_count = orbit_size;
    r->scaled_type = type_tag;
    return 0;
}

uint32_t fs_render_scaled(fs_renderer *r, int type_tag, const void *coords, uint64_t n_iterations)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized())
        return 0; // GPU_Render.cu:1317-1319
    if (r->local_rows == 0)
        return 0; // this renderer owns no row of the frame (a rank beyond the last band)
    if (type_tag != FS_T_HDR32 && type_tag != FS_T_F64)
        return FS_ERR_UNSUPPORTED;
    if (n_iterations > 0xFFFFFFFFull && r->iter_bytes != 8)
        return (uint32_t)hipErrorInvalidValue;
    if (!r->scaled_t || !r->scaled_f || r->scaled_count < 2 || r->scaled_type != type_tag)
        return FS_ERR_6;
    const float w2threshold = (float)exp(log((double)1e30f) / 2.0);
    if (type_tag == FS_T_F64) {
        FsScaledArgsF64 A;
        memset(&A, 0, sizeof(A));
        A.out = (uint32_t *)r->iters();
        A.orbit_t = (const fs_orbit_f64_bad *)r->scaled_t;
        A.orbit_f = r->scaled_f;
        A.stats = r->stats;
        A.frame = make_frame(r);
        const double *c = (const double *)coords;
        A.dx = c[0], A.dy = c[1], A.centerX = c[2], A.centerY = c[3];
        A.orbit_count = (uint32_t)r->scaled_count;
        A.n_iterations = (uint32_t)n_iterations;
        A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
        A.w2threshold = w2threshold;
        TimedLaunch t(r);
        fsk_scaled_f64(A, r->stats_on, r->variant & FS_VARIANT_BASE_MASK, r->compute);
        return (uint32_t)hipGetLastError();
    }
    FsScaledArgs32 A;
    memset(&A, 0, sizeof(A));
    A.out = (uint32_t *)r->iters();
    A.orbit_t = (const fs_orbit_hdr32_bad *)r->scaled_t;
    A.orbit_f = r->scaled_f;
    A.stats = r->stats;
    A.frame = make_frame(r);
    fill_coords(A.coords, coords);
    A.orbit_count = (uint32_t)r->scaled_count;
    A.n_iterations = (uint32_t)n_iterations;
    A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
    A.w2threshold = w2threshold;
    TimedLaunch t(r);
    fsk_scaled_hdr32(A, r->stats_on, r->variant & FS_VARIANT_BASE_MASK, r->compute);
    return (uint32_t)hipGetLastError();
}

uint32_t fs_render_direct_lp(fs_renderer *r, int type_tag, const void *coords, uint64_t n_iterations,
                             int iteration_precision)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized())
        return 0; // GPU_Render.cu:626-628
    if (r->local_rows == 0)
        return 0; // this renderer owns no row of the frame (a rank beyond the last band)
    if (type_tag != FS_T_F32 && type_tag != FS_T_2X32 && type_tag != FS_T_2X64 && type_tag != FS_T_4X32 &&
        type_tag != FS_T_4X64)
        return FS_ERR_UNSUPPORTED;
    if (n_iterations > 0xFFFFFFFFull && r->iter_bytes != 8)
        return (uint32_t)hipErrorInvalidValue;
    FsDirectLpArgs A;
    memset(&A, 0, sizeof(A));
    A.out = (uint32_t *)r->iters();
    A.stats = r->stats;
    A.frame = make_frame(r);
    A.n_iterations = (uint32_t)n_iterations;
    A.n_iterations_hi = (uint32_t)(n_iterations >> 32);
        A.frame.wide |= A.n_iterations_hi != 0u ? 1u : 0u;
        r->last_launch_wide = A.frame.wide != 0u;
    if (type_tag == FS_T_F32)
        memcpy(A.c32, coords, 4 * sizeof(float));
    else if (type_tag == FS_T_2X32)
        memcpy(A.c32, coords, 8 * sizeof(float));
    else if (type_tag == FS_T_4X32)
        memcpy(A.c32, coords, 16 * sizeof(float));
    else if (type_tag == FS_T_4X64)
        memcpy(A.c64, coords, 16 * sizeof(double));
    else
        memcpy(A.c64, coords, 8 * sizeof(double));
    const int kind = type_tag == FS_T_F32    ? 0
                     : type_tag == FS_T_2X32 ? 1
                     : type_tag == FS_T_2X64 ? 2
                     : type_tag == FS_T_4X32 ? 3
                                             : 4;
    TimedLaunch t(r);
    (void)fsk_direct_lp(A, kind, iteration_precision, r->stats_on, r->compute);
    return (uint32_t)hipGetLastError();
}

uint32_t fs_clear(fs_renderer *r)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized())
        return 0;
    const size_t elems = (size_t)r->w_block * 16u * r->local_rows_padded;
    FS_TRY(hipMemsetAsync(r->iters(), 0, elems * r->iter_bytes, r->compute));
    if (r->colors)
        FS_TRY(hipMemsetAsync(r->colors, 0, r->n_color_cu * sizeof(fs_color16), r->compute));
    return 0;
}

uint32_t fs_render_current(fs_renderer *r, uint64_t n_iterations, void *iter_buffer, fs_color16 *color_buffer,
                           fs_reduction *reduction, int progressive)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized())
        return 0; // GPU_Render.cu:564-566
    hipStream_t s = progressive ? r->display : r->compute;
    const uint32_t rw = r->w_block * 16u;
    const bool whole_frame = r->local_rows == r->height;
    if (color_buffer && r->pal && whole_frame) {
        fsk_antialias(r->iters(), r->iter_bytes == 8, rw, r->colors, r->pal, r->pal_iters, r->pal_aux_depth, r->aa,
                      r->color_w, r->color_h, n_iterations, s);
        FS_TRY(hipGetLastError());
    }
    if (reduction) {
        r->reduce_seed = fs_reduction{r->iter_bytes == 8 ? ~0ull : 0xFFFFFFFFull, 0, 0}; // ReductionKernels.cuh:99-104
        FS_TRY(hipMemcpyAsync(r->reduction, &r->reduce_seed, sizeof(fs_reduction), hipMemcpyHostToDevice, s));
        fsk_reduce(r->iters(), r->iter_bytes == 8, rw, r->width, r->local_rows, r->reduction, s);
        FS_TRY(hipGetLastError());
    }
    // ExtractItersAndColors, GPU_Render.cu:1759-1805: padding included.
    if (iter_buffer)
        FS_TRY(hipMemcpyAsync(iter_buffer, r->iters(), (size_t)rw * r->local_rows_padded * r->iter_bytes,
                              hipMemcpyDefault, s));
    if (color_buffer && whole_frame)
        FS_TRY(hipMemcpyAsync(color_buffer, r->colors, r->n_color_cu * sizeof(fs_color16), hipMemcpyDefault, s));
    if (reduction)
        FS_TRY(hipMemcpyAsync(reduction, r->reduction, sizeof(fs_reduction), hipMemcpyDefault, s));
    return 0;
}

uint32_t fs_time_render_current(fs_renderer *r, uint64_t n_iterations, uint32_t repeats, float ms_out[2])
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized() || !r->pal || r->local_rows != r->height || !repeats)
        return FS_ERR_6;
    const uint32_t rw = r->w_block * 16u;
    hipEvent_t a, b;
    FS_TRY(hipEventCreate(&a));
    FS_TRY(hipEventCreate(&b));
    // the kernels only (the 24-byte seed copy of fs_render_current is not part of what is measured; min / max are
    // idempotent and the accumulated sum of the repeats is discarded)
    r->reduce_seed = fs_reduction{r->iter_bytes == 8 ? ~0ull : 0xFFFFFFFFull, 0, 0};
    FS_TRY(hipMemcpyAsync(r->reduction, &r->reduce_seed, sizeof(fs_reduction), hipMemcpyHostToDevice, r->compute));
    for (int which = 0; which < 2; which++) {
        FS_TRY(hipEventRecord(a, r->compute));
        for (uint32_t i = 0; i < repeats; i++) {
            if (which == 0)
                fsk_antialias(r->iters(), r->iter_bytes == 8, rw, r->colors, r->pal, r->pal_iters, r->pal_aux_depth, r->aa,
                              r->color_w, r->color_h, n_iterations, r->compute);
            else
                fsk_reduce(r->iters(), r->iter_bytes == 8, rw, r->width, r->local_rows, r->reduction, r->compute);
        }
        FS_TRY(hipEventRecord(b, r->compute));
        FS_TRY(hipEventSynchronize(b));
        FS_TRY(hipGetLastError());
        float ms = 0;
        FS_TRY(hipEventElapsedTime(&ms, a, b));
        ms_out[which] = ms / (float)repeats;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return 0;
}

uint32_t fs_sync_compute(fs_renderer *r)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->compute)
        return 0;
    return (uint32_t)hipStreamSynchronize(r->compute);
}

void *fs_compute_stream(const fs_renderer *r) { return (void *)r->compute; }
void *fs_display_stream(const fs_renderer *r) { return (void *)r->display; }

// RunAntialiasing (GPU_Render.cu:1695-1757) over a whole frame that lies somewhere else on this renderer's device (the frame
// an fs_group has put back in row order), with this renderer's palette and geometry, on the caller's stream.
uint32_t fs_colorize_frame(fs_renderer *r, const void *device_iters, uint64_t n_iterations, fs_color16 *device_colors,
                           fs_color16 *color_buffer, void *stream)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->memory_initialized() || !device_iters)
        return 0;
    if (!r->pal)
        return 0; // no palette was ever uploaded: RenderCurrent leaves the colour buffer alone
    hipStream_t s = (hipStream_t)stream;
    fs_color16 *dst = device_colors ? device_colors : r->colors;
    fsk_antialias(device_iters, r->iter_bytes == 8, r->w_block * 16u, dst, r->pal, r->pal_iters, r->pal_aux_depth, r->aa,
                  r->color_w, r->color_h, n_iterations, s);
    FS_TRY(hipGetLastError());
    if (color_buffer)
        FS_TRY(hipMemcpyAsync(color_buffer, dst, r->n_color_cu * sizeof(fs_color16), hipMemcpyDefault, s));
    return 0;
}
uint64_t fs_color_buffer_elements(const fs_renderer *r) { return r->n_color_cu; }

uint32_t fs_sync_display(fs_renderer *r)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->display)
        return 0;
    return (uint32_t)hipStreamSynchronize(r->display);
}

uint32_t fs_query_compute(fs_renderer *r)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->compute)
        return 0;
    return (uint32_t)hipStreamQuery(r->compute);
}

struct DoneThunk {
    fs_done_cb cb;
    void *user;
};

static void done_trampoline(void *p)
{
    DoneThunk *t = (DoneThunk *)p;
    t->cb(t->user);
    delete t;
}

uint32_t fs_enqueue_done_callback(fs_renderer *r, fs_done_cb cb, void *user)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!cb)
        return FS_ERR_6;
    DoneThunk *t = new DoneThunk{cb, user};
    // before InitializeMemory the reference's m_ComputeStream is the null stream and the callback still fires
    // (GPU_Render.cu:613-615); r->compute == nullptr behaves the same way
    const hipError_t e = hipLaunchHostFunc(r->compute, done_trampoline, t);
    if (e != hipSuccess)
        delete t;
    return (uint32_t)e;
}

uint64_t fs_host_fallback_bytes(const fs_renderer *r) { return r->host_alloc_bytes; }

uint64_t fs_idle_device_bytes(fs_renderer *r)
{
    std::lock_guard<std::mutex> g(r->kept_mu);
    uint64_t b = 0;
    for (const auto &k : r->kept_blocks)
        b += k.bytes;
    return b;
}

uint64_t fs_release_idle_device_memory(int device) { return release_idle_memory_of_device(device); }

uint32_t fs_set_compressed_orbit_mode(fs_renderer *r, int mode)
{
    if (mode != 0 && mode != 1)
        return hipErrorInvalidValue;
    r->compressed_mode = mode;
    return 0;
}

uint64_t fs_orbit_device_bytes(const fs_renderer *r)
{
    if (!r->orbit_ok)
        return 0;
    const uint64_t n = r->orbit_uncompressed;
    if (r->orbit_seq) {
        switch (r->orbit_type) {
            case FS_T_HDR32: return r->orbit_size * sizeof(fs_orbit_hdr32_rc);
            case FS_T_HDR64: return r->orbit_size * sizeof(fs_orbit_hdr64_rc);
            case FS_T_F32: return r->orbit_size * sizeof(fs_orbit_f32_rc);
            case FS_T_F64: return r->orbit_size * sizeof(fs_orbit_f64_rc);
            case FS_T_2X32: return r->orbit_size * sizeof(fs_orbit_p2x32_rc);
            default: return r->orbit_size * sizeof(fs_orbit_2x32_rc);
        }
    }
    switch (r->orbit_type) {
        case FS_T_HDR32: // prepared entries + the two companion arrays of the tuned loops
            return (n + 2) * sizeof(float4) + quiet_orbit_units(n) * sizeof(float4);
        case FS_T_HDR64:
            return (n + 2) * sizeof(FsZ64);
        case FS_T_F64:
            return n * sizeof(fs_orbit_f64);
        case FS_T_HDR2X32:
            return n * sizeof(fs_orbit_2x32);
        case FS_T_F32:
            return (n + 1) * sizeof(fs_orbit_f32);
        case FS_T_2X32:
            return (n + 1) * sizeof(fs_orbit_p2x32);
        default:
            return 0;
    }
}

uint32_t fs_get_width(const fs_renderer *r) { return r->width; }
uint32_t fs_get_height(const fs_renderer *r) { return r->height; }

float fs_last_kernel_ms(const fs_renderer *r)
{
    if (r->timed_launches == 0)
        return -1.0f;
    float ms = -1.0f;
    const uint32_t i = (uint32_t)((r->timed_launches - 1) % fs_renderer::kTimingRing);
    if (hipEventElapsedTime(&ms, r->ev_start[i], r->ev_stop[i]) != hipSuccess)
        return -1.0f;
    return ms;
}

uint32_t fs_kernel_ms_history(const fs_renderer *r, float *ms_out, uint32_t n)
{
    // the last n launches, oldest first; they must have completed (fs_sync_compute)
    if (n > fs_renderer::kTimingRing || n > r->timed_launches)
        return (uint32_t)hipErrorInvalidValue;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t i = (uint32_t)((r->timed_launches - n + k) % fs_renderer::kTimingRing);
        FS_TRY(hipEventElapsedTime(&ms_out[k], r->ev_start[i], r->ev_stop[i]));
    }
    return 0;
}

uint32_t fs_kernel_ms_split_history(const fs_renderer *r, float *first_ms, float *second_ms, uint32_t n)
{
    if (n > fs_renderer::kTimingRing || n > r->timed_launches)
        return (uint32_t)hipErrorInvalidValue;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t i = (uint32_t)((r->timed_launches - n + k) % fs_renderer::kTimingRing);
        if (r->mid_valid[i] && r->ev_mid[i]) {
            FS_TRY(hipEventElapsedTime(&first_ms[k], r->ev_start[i], r->ev_mid[i]));
            FS_TRY(hipEventElapsedTime(&second_ms[k], r->ev_mid[i], r->ev_stop[i]));
        } else {
            first_ms[k] = 0.0f;
            FS_TRY(hipEventElapsedTime(&second_ms[k], r->ev_start[i], r->ev_stop[i]));
        }
    }
    return 0;
}

uint32_t fs_set_kernel_variant(fs_renderer *r, int variant)
{
    const int base = variant & FS_VARIANT_BASE_MASK, flags = variant & ~FS_VARIANT_BASE_MASK;
    if (base > FS_VARIANT_TUNED_NOSCALE ||
        (flags & ~(FS_VARIANT_FLAG_LDS_ORBIT | FS_VARIANT_FLAG_REFILL | FS_VARIANT_FLAG_WIDE | FS_VARIANT_FLAG_NATURAL_ORDER |
                   FS_VARIANT_FLAG_BLA_POOL)) != 0)
        return hipErrorInvalidValue;
    r->variant = base | flags;
    return 0;
}

uint32_t fs_forget_tile_costs(fs_renderer *r)
{
    r->lav2_cost_valid = false;
    r->po_order_valid = false;
    r->pix_valid = false;
    r->pix_seen = false;
    r->at_order_valid = false;
    return 0;
}

int fs_last_frame_tile_ordered(fs_renderer *r) { return r->last_frame_ordered ? 1 : 0; }
int fs_last_frame_sampled_tile_order(fs_renderer *r) { return r->last_cold_ordered ? 1 : 0; }

uint32_t fs_read_tile_costs(fs_renderer *r, uint32_t *out, uint64_t max_words, uint64_t *n_tiles)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->lav2_cost || !r->lav2_cost_valid)
        return FS_ERR_6;
    const uint64_t n = (uint64_t)((r->lav2_cost_key.width + 7u) / 8u) * ((r->lav2_cost_key.local_rows + 7u) / 8u);
    if (n_tiles)
        *n_tiles = n;
    const uint64_t m = n < max_words ? n : max_words;
    if (out && m) {
        FS_TRY(hipMemcpyAsync(out, r->lav2_cost, m * sizeof(uint32_t), hipMemcpyDeviceToHost, r->compute));
        FS_TRY(hipStreamSynchronize(r->compute));
    }
    return 0;
}

uint32_t fs_seq_cursor_probe(fs_renderer *r, int wide_positions, uint64_t start, uint32_t n, void *out)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->orbit_ok || !r->orbit_seq || !r->wp_raw || !out)
        return FS_ERR_6;
    const bool is64 = r->orbit_type == FS_T_HDR64;
    const size_t rec = is64 ? sizeof(fs::hcplx<double>) : sizeof(fs::hcplx<float>);
    void *dev = nullptr;
    FS_TRY(r_alloc(r, &dev, (size_t)n * rec, kFrame));
    fsk_seq_cursor_probe(is64, wide_positions != 0, r->wp_raw, (uint32_t)r->orbit_size,
                         is64 ? (const void *)&r->c_low64[0] : (const void *)&r->c_low32[0],
                         is64 ? (const void *)&r->c_low64[1] : (const void *)&r->c_low32[1], start, n, dev, r->compute);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess)
        err = hipMemcpyAsync(out, dev, (size_t)n * rec, hipMemcpyDeviceToHost, r->compute);
    if (err == hipSuccess)
        err = hipStreamSynchronize(r->compute);
    (void)r_free(r, dev);
    return (uint32_t)err;
}

uint32_t fs_read_tile_order(fs_renderer *r, uint32_t *out, uint64_t max_words)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->lav2_order || !r->lav2_last_ordered)
        return FS_ERR_6;
    const uint64_t n = (uint64_t)((r->lav2_cost_key.width + 7u) / 8u) * ((r->lav2_cost_key.local_rows + 7u) / 8u);
    const uint64_t m = n < max_words ? n : max_words;
    FS_TRY(hipMemcpyAsync(out, r->lav2_order, m * sizeof(uint32_t), hipMemcpyDeviceToHost, r->compute));
    FS_TRY(hipStreamSynchronize(r->compute));
    return 0;
}

uint32_t fs_enable_step_count(fs_renderer *r, int enable)
{
    r->stats_on = enable != 0;
    return 0;
}

uint32_t fs_test_block_threshold(fs_renderer *r, const int32_t *bound_bits, const int32_t *scale_shift, const int32_t *dc_bits,
                                 int32_t *threshold_out, uint32_t n)
{
    if (!r || !bound_bits || !scale_shift || !dc_bits || !threshold_out)
        return (uint32_t)hipErrorInvalidValue;
    if (n == 0)
        return 0;
    FS_TRY(hipSetDevice(r->device));
    int *d = nullptr;
    FS_TRY(hipMalloc((void **)&d, (size_t)n * 4 * sizeof(int)));
    uint32_t rc = (uint32_t)hipMemcpy(d, bound_bits, n * sizeof(int), hipMemcpyHostToDevice);
    if (!rc)
        rc = (uint32_t)hipMemcpy(d + n, scale_shift, n * sizeof(int), hipMemcpyHostToDevice);
    if (!rc)
        rc = (uint32_t)hipMemcpy(d + 2 * (size_t)n, dc_bits, n * sizeof(int), hipMemcpyHostToDevice);
    if (!rc) {
        fsk_test_block_threshold(d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, n, r->compute);
        rc = (uint32_t)hipGetLastError();
    }
    if (!rc)
        rc = (uint32_t)hipStreamSynchronize(r->compute);
    if (!rc)
        rc = (uint32_t)hipMemcpy(threshold_out, d + 3 * (size_t)n, n * sizeof(int), hipMemcpyDeviceToHost);
    hipFree(d);
    return rc;
}

uint32_t fs_read_stats_raw(fs_renderer *r, uint64_t *out, uint64_t max_words)
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->stats)
        return FS_ERR_6;
    const size_t n = r->stats_words < max_words ? r->stats_words : (size_t)max_words;
    FS_TRY(hipMemcpyAsync(out, r->stats, n * sizeof(uint64_t), hipMemcpyDeviceToHost, r->compute));
    FS_TRY(hipStreamSynchronize(r->compute));
    return 0;
}

uint32_t fs_read_step_count(fs_renderer *r, uint64_t counts[8])
{
    if (uint32_t e = use_device(r))
        return e;
    if (!r->stats)
        return FS_ERR_6;
    // the 64-bit counting instantiations are not built with the step counters: zeros would read as "no work was done"
    if (r->last_launch_wide)
        return FS_ERR_UNSUPPORTED;
    // ordered behind the kernels of the (non-blocking) compute stream, which the null stream is not
    FS_TRY(hipMemcpyAsync(counts, r->stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, r->compute));
    FS_TRY(hipStreamSynchronize(r->compute));
    return 0;
}

} // extern "C"
