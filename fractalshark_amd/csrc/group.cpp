// group.cpp -- fs_group_*: one frame row-tiled over the GPUs of one node behind the C ABI (include/fsmi355.h).
//
// The reference is single-device (GPU_Render.cu:113 hard-codes device 0); north_star asks for "frames row-tiled across
// the 8 GPUs of one node with a RCCL gather over xGMI of the iteration buffer".  A C++ host that links libfsmi355.so
// (the FractalShark drop-in of INTEGRATION.md) is ONE process, so the group is single-process multi-GPU: one fs_renderer
// per device, one RCCL communicator per device from ncclCommInitAll, and the gather is N-1 ncclSend / one rank-0
// ncclRecv per peer inside a single ncclGroupStart / ncclGroupEnd, enqueued on the members' own compute streams (so it
// is ordered behind each member's kernel without a host round trip).  Only rank 0 needs the frame (it is what
// RenderCurrent hands to the host), so this moves (N-1)/N of the frame once over the point-to-point xGMI links into
// device 0 -- an all-gather would move N times as much for nothing.
//
// Partitioning = fractalshark_amd/tiling.py: rank r owns the 8-row bands k*N + r (fine interleaving: slow pixels cluster
// spatially), the kernels keep the GLOBAL row in the pixel -> delta-c mapping (fs_set_row_bands), slices are padded to
// equal size, and one HBM-bound kernel on device 0 restores row order (k_gather_rows).
//
// RCCL is resolved at run time (dlopen "librccl.so.1"): the library has no link-time dependency on it and a one-GPU
// host never loads it.  transport: 0 = RCCL (default for distinct devices), 1 = hipMemcpyPeerAsync (members that share
// a device -- the one-GPU test box -- and hosts without RCCL).
#include <cstdlib>
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/fsmi355_internal.h"
#include "kernels.h"

namespace {

// the few RCCL entry points used, with the signatures of rccl.h (ncclResult_t and ncclDataType_t are ints there)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

// RCCL's intra-node transport shares buffers between devices through IPC handles, and the hosts of this pool only support the
// dmabuf form (hipIpcGetMemHandle: invalid argument otherwise).  The ROCm runtime reads the switch once, when it comes up
// (the first HIP call of the process), so it is set when THIS library is loaded -- before any call of ours can be that first
// call -- and never over a value the host application has chosen itself.
__attribute__((constructor)) void fs_env_defaults() { (void)setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", /*overwrite=*/0); }

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl q;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            q.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (q.lib)
                break;
        }
        if (!q.lib)
            return q;
        q.CommInitAll = (decltype(q.CommInitAll))dlsym(q.lib, "ncclCommInitAll");
        q.CommDestroy = (decltype(q.CommDestroy))dlsym(q.lib, "ncclCommDestroy");
        q.GroupStart = (decltype(q.GroupStart))dlsym(q.lib, "ncclGroupStart");
        q.GroupEnd = (decltype(q.GroupEnd))dlsym(q.lib, "ncclGroupEnd");
        q.Send = (decltype(q.Send))dlsym(q.lib, "ncclSend");
        q.Recv = (decltype(q.Recv))dlsym(q.lib, "ncclRecv");
        q.GetErrorString = (decltype(q.GetErrorString))dlsym(q.lib, "ncclGetErrorString");
        q.ok = q.CommInitAll && q.CommDestroy && q.GroupStart && q.GroupEnd && q.Send && q.Recv;
        return q;
    }();
    return r;
}

constexpr int kNcclUint8 = 1; // ncclUint8, rccl.h

} // namespace

struct fs_group {
    std::vector<int> devices;
    std::vector<fs_renderer *> members;
    std::vector<void *> comms;     // ncclComm_t per member (RCCL transport)
    // per member r >= 1: its slice buffers on its own device (the external iteration buffer of the frame in set b is
    // slices[b][r]; member 0 renders straight into gather slot 0 of the set).  Two, in rotation with the sets: under the direct host
    // path a slice is still being copied to the host over the member's own link while the member renders its next frame.
    std::vector<void *> slices[2];
    int transport = 0;
    int host_path = 0;                      // 0 = whole frame through device 0 (gather), 1 = every member copies its own bands
    std::vector<hipStream_t> copy_streams;  // direct host path: one per member, on its device
    std::vector<hipEvent_t> ev_kernel;      // per member: its kernel of the frame being delivered has finished
    std::vector<hipEvent_t> ev_copied[2];   // per set, per member: its bands of the set's frame are in the caller's host buffer
    std::vector<char> copied_recorded[2];
    uint32_t width = 0, height = 0, band = 8, iter_bytes = 4, rounded_width = 0, max_rows = 0;
    // Two frames can be in flight: while frame k is received, put in row order, reduced and copied to the
    // host on device 0's POST stream, the members render frame k+1.  Everything frame k's post-processing reads or writes
    // on device 0 therefore exists twice, used in rotation (`cur` = the set the frame being rendered lands in):
    static constexpr int kSets = 2;
    void *gathered[kSets] = {};    // device 0: N slices back to back (member 0 renders straight into slot 0)
    void *frame[kSets] = {};       // device 0: rows in order, padded to 8
    uint32_t *index = nullptr;     // device 0: frame row -> gathered row
    fs_reduction *reduction = nullptr;
    fs_reduction reduce_seed{};
    float last_gather_ms = -1.0f;
    int cur = 0;
    uint64_t frames_posted = 0;         // fs_group_render_current calls so far
    hipStream_t post = nullptr;         // device 0: receive + row order + reduction + D2H of a finished frame
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    hipEvent_t ev_rendered[kSets] = {}; // member 0's kernel of the frame in set b has finished (recorded on its compute stream)
    hipEvent_t ev_consumed[kSets] = {}; // k_gather_rows has read gathered[b]: its slots may be written again
    hipEvent_t ev_done[kSets] = {};     // the frame of set b is in the caller's host buffer
    // progressive RenderCurrent (a snapshot of the frame while the members' kernels are still writing it, GPU_Render.cu:556-581
    // with progressive = true): a third set of device-0 buffers of its own, filled on the members' DISPLAY streams, so that
    // it neither waits for a kernel nor touches what the two frames in flight use.  Allocated by the first progressive call.
    void *prog_gathered = nullptr, *prog_frame = nullptr;
    fs_color16 *prog_colors = nullptr;
    fs_reduction *prog_reduction = nullptr;
    fs_reduction prog_seed{};
    size_t slice_bytes() const { return (size_t)max_rows * rounded_width * iter_bytes; }
};

extern "C" {

void fs_group_plan(uint32_t height, uint32_t world, uint32_t band, uint32_t rank, uint32_t *local_rows,
                   uint32_t *max_local_rows, uint32_t *frame_index)
{
    // rows of `rank` in local-buffer order, and the slot every rank's slice is padded to (rank 0 owns the most)
    auto rows_of = [&](uint32_t r) {
        uint32_t n = 0;
        for (uint64_t s = (uint64_t)r * band; s < height; s += (uint64_t)world * band)
            n += (uint32_t)((s + band < height ? s + band : height) - s);
        return n;
    };
    const uint32_t m = (rows_of(0) + 7u) / 8u * 8u;
    if (local_rows)
        *local_rows = rows_of(rank);
    if (max_local_rows)
        *max_local_rows = m;
    if (frame_index)
        for (uint32_t r = 0; r < world; r++) {
            uint32_t k = 0;
            for (uint64_t s = (uint64_t)r * band; s < height; s += (uint64_t)world * band)
                for (uint64_t y = s; y < s + band && y < height; y++)
                    frame_index[y] = r * m + k++;
        }
}

fs_group *fs_group_create(const int *devices, int n_devices, int transport)
{
    if (!devices || n_devices < 1 || n_devices > 64)
        return nullptr;
    fs_group *g = new (std::nothrow) fs_group();
    if (!g)
        return nullptr;
    g->devices.assign(devices, devices + n_devices);
    bool distinct = true;
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < i; j++)
            distinct = distinct && devices[i] != devices[j];
    g->transport = (transport == 1 || !distinct || n_devices == 1) ? 1 : 0;
    for (int i = 0; i < n_devices; i++) {
        fs_renderer *r = fs_create(devices[i]);
        if (!r) {
            fs_group_destroy(g);
            return nullptr;
        }
        g->members.push_back(r);
    }
    if (g->transport == 0) {
        Rccl &q = rccl();
        g->comms.assign((size_t)n_devices, nullptr);
        // (advisor, round 5) The dmabuf-IPC switch only works when it was in the environment BEFORE the ROCm runtime came up: the
        // constructor above sets it when this library is loaded, which is too late in a process that made a HIP call first (or
        // imported torch first).  Say so instead of failing later with "invalid argument".
        const char *ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
        if (!ipc || strcmp(ipc, "0") != 0)
            fprintf(stderr, "fsmi355: HSA_ENABLE_IPC_MODE_LEGACY is %s: RCCL between the group's devices needs it to be 0 in the environment "
                            "before the first HIP call of the process (load libfsmi355 first, or export it)\n", ipc ? ipc : "unset");
        if (!q.ok || q.CommInitAll(g->comms.data(), n_devices, devices) != 0) {
            fprintf(stderr, "fsmi355: RCCL unavailable or ncclCommInitAll failed (if the log above it says hipIpcGetMemHandle: invalid argument, "
                            "HSA_ENABLE_IPC_MODE_LEGACY=0 was not in the environment when the ROCm runtime came up); the group falls back to peer copies\n");
            g->comms.clear();
            g->transport = 1;
        }
    }
    return g;
}

static void group_free_buffers(fs_group *g)
{
    if (g->prog_gathered) // a progressive snapshot may still be copying out of the slices on the display streams
        for (fs_renderer *m : g->members)
            (void)fs_sync_display(m);
    for (size_t i = 0; i < g->copy_streams.size(); i++) // (a direct copy may still be reading a slice)
        if (g->copy_streams[i] && hipSetDevice(g->devices[i]) == hipSuccess)
            (void)hipStreamSynchronize(g->copy_streams[i]);
    for (int b = 0; b < 2; b++) {
        for (size_t i = 0; i < g->slices[b].size(); i++)
            if (g->slices[b][i] && hipSetDevice(g->devices[i]) == hipSuccess) {
                (void)fs_set_external_iter_buffer(g->members[i], nullptr, 0);
                (void)hipFree(g->slices[b][i]);
            }
        g->slices[b].clear();
        std::fill(g->copied_recorded[b].begin(), g->copied_recorded[b].end(), 0);
    }
    // member 0 renders straight into its gather slot (no slice of its own): detach it before that memory goes away
    if (g->gathered[0] && !g->members.empty())
        (void)fs_set_external_iter_buffer(g->members[0], nullptr, 0);
    if (!g->devices.empty() && hipSetDevice(g->devices[0]) == hipSuccess) {
        if (g->post)
            (void)hipStreamSynchronize(g->post);
        for (int b = 0; b < fs_group::kSets; b++) {
            if (g->gathered[b])
                (void)hipFree(g->gathered[b]);
            if (g->frame[b])
                (void)hipFree(g->frame[b]);
            g->gathered[b] = g->frame[b] = nullptr;
        }
        if (g->index)
            (void)hipFree(g->index);
        if (g->reduction)
            (void)hipFree(g->reduction);
        if (!g->members.empty() && fs_display_stream(g->members[0]))
            (void)hipStreamSynchronize((hipStream_t)fs_display_stream(g->members[0]));
        for (void *p : {g->prog_gathered, g->prog_frame, (void *)g->prog_colors, (void *)g->prog_reduction})
            if (p)
                (void)hipFree(p);
    }
    g->prog_gathered = g->prog_frame = nullptr;
    g->prog_colors = nullptr;
    g->prog_reduction = nullptr;
    g->index = nullptr;
    g->reduction = nullptr;
    g->cur = 0;
    g->frames_posted = 0; // (fs_group_wait_current counts back from here: no frame of the new geometry has been posted)
}

void fs_group_destroy(fs_group *g)
{
    if (!g)
        return;
    (void)fs_group_sync(g);
    group_free_buffers(g);
    if (g->ev_a)
        (void)hipEventDestroy(g->ev_a);
    if (g->ev_b)
        (void)hipEventDestroy(g->ev_b);
    for (int b = 0; b < fs_group::kSets; b++) {
        if (g->ev_rendered[b])
            (void)hipEventDestroy(g->ev_rendered[b]);
        if (g->ev_consumed[b])
            (void)hipEventDestroy(g->ev_consumed[b]);
        if (g->ev_done[b])
            (void)hipEventDestroy(g->ev_done[b]);
    }
    if (g->post && !g->devices.empty() && hipSetDevice(g->devices[0]) == hipSuccess)
        (void)hipStreamDestroy(g->post);
    for (size_t i = 0; i < g->copy_streams.size(); i++)
        if (hipSetDevice(g->devices[i]) == hipSuccess) {
            if (g->copy_streams[i])
                (void)hipStreamDestroy(g->copy_streams[i]);
            if (i < g->ev_kernel.size() && g->ev_kernel[i])
                (void)hipEventDestroy(g->ev_kernel[i]);
            for (int b = 0; b < 2; b++)
                if (i < g->ev_copied[b].size() && g->ev_copied[b][i])
                    (void)hipEventDestroy(g->ev_copied[b][i]);
        }
    for (void *c : g->comms)
        if (c)
            rccl().CommDestroy(c);
    for (fs_renderer *r : g->members)
        fs_destroy(r);
    delete g;
}

int fs_group_size(const fs_group *g) { return (int)g->members.size(); }
fs_renderer *fs_group_renderer(fs_group *g, int rank)
{
    return rank >= 0 && (size_t)rank < g->members.size() ? g->members[(size_t)rank] : nullptr;
}
int fs_group_transport(const fs_group *g) { return g->transport; }

#define FSG_TRY(expr)                                                                                                 \
    do {                                                                                                              \
        const uint32_t e_ = (uint32_t)(expr);                                                                         \
        if (e_ != 0)                                                                                                  \
            return e_;                                                                                                \
    } while (0)

uint32_t fs_group_init_memory(fs_group *g, uint32_t w, uint32_t h, uint32_t antialiasing, uint32_t iter_bytes,
                              const fs_color16 *pal_interleaved, uint32_t pal_iters, uint32_t palette_aux_depth,
                              uint64_t palette_generation)
{
    const uint32_t world = (uint32_t)g->members.size();
    group_free_buffers(g);
    g->band = antialiasing == 3 ? 24u : 8u; // a band never splits an antialiasing row group (tiling.py)
    for (uint32_t r = 0; r < world; r++) {
        FSG_TRY(fs_init_memory(g->members[r], w, h, antialiasing, iter_bytes, pal_interleaved, pal_iters, palette_aux_depth,
                               palette_generation, 0));
        FSG_TRY(fs_set_row_bands(g->members[r], r * g->band, g->band, world * g->band));
    }
    g->width = w;
    g->height = h;
    g->iter_bytes = iter_bytes;
    g->rounded_width = fs_rounded_width(g->members[0]);
    std::vector<uint32_t> idx(h);
    fs_group_plan(h, world, g->band, 0, nullptr, &g->max_rows, idx.data());
    // device 0: the gather target (N padded slices), the ordered frame, the row index, the reduction cell
    FSG_TRY(hipSetDevice(g->devices[0]));
    const size_t sb = g->slice_bytes();
    const size_t frame_rows = ((size_t)h + 7u) / 8u * 8u;
    for (int b = 0; b < fs_group::kSets; b++) {
        FSG_TRY(hipMalloc(&g->gathered[b], sb * world));
        FSG_TRY(hipMemset(g->gathered[b], 0, sb * world));
        FSG_TRY(hipMalloc(&g->frame[b], frame_rows * g->rounded_width * iter_bytes));
        FSG_TRY(hipMemset(g->frame[b], 0, frame_rows * g->rounded_width * iter_bytes));
    }
    FSG_TRY(hipMalloc((void **)&g->index, sizeof(uint32_t) * h));
    FSG_TRY(hipMemcpy(g->index, idx.data(), sizeof(uint32_t) * h, hipMemcpyHostToDevice));
    FSG_TRY(hipMalloc((void **)&g->reduction, sizeof(fs_reduction)));
    if (!g->ev_a) {
        FSG_TRY(hipEventCreate(&g->ev_a));
        FSG_TRY(hipEventCreate(&g->ev_b));
        for (int b = 0; b < fs_group::kSets; b++) {
            FSG_TRY(hipEventCreateWithFlags(&g->ev_rendered[b], hipEventDisableTiming));
            FSG_TRY(hipEventCreateWithFlags(&g->ev_consumed[b], hipEventDisableTiming));
            FSG_TRY(hipEventCreateWithFlags(&g->ev_done[b], hipEventDisableTiming));
        }
        // highest priority: when a finished frame's slices arrive, device 0 is already rendering the next frame on its
        // (lowest-priority) compute stream, and the row-order kernel takes the first wave slots that come free
        int prio_least = 0, prio_greatest = 0;
        FSG_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        FSG_TRY(hipStreamCreateWithPriority(&g->post, hipStreamNonBlocking, prio_greatest));
    }
    g->cur = 0;
    // every member renders into a slice buffer of the common (padded) size; member 0 straight into its gather slot
    for (int b = 0; b < 2; b++) {
        g->slices[b].assign(world, nullptr);
        g->copied_recorded[b].assign(world, 0);
    }
    for (uint32_t r = 0; r < world; r++) {
        void *buf = nullptr;
        if (r == 0) {
            buf = g->gathered[0];
        } else {
            FSG_TRY(hipSetDevice(g->devices[r]));
            for (int b = 1; b >= 0; b--) {
                FSG_TRY(hipMalloc(&buf, sb));
                FSG_TRY(hipMemset(buf, 0, sb));
                g->slices[b][r] = buf;
            }
        }
        FSG_TRY(fs_set_external_iter_buffer(g->members[r], buf, sb));
    }
    // The clears above run on the devices' default streams, which the members' (non-blocking) streams do not wait for: a clear
    // that is still pending could land on top of what the first frame's kernels write.  One wait per geometry.
    for (uint32_t r = 0; r < world; r++) {
        FSG_TRY(hipSetDevice(g->devices[r]));
        FSG_TRY(hipDeviceSynchronize());
    }
    return 0;
}

// ---- replicated uploads (inputs are small next to the frame: 257 KB at C3; the host pointer is read once per member)
uint32_t fs_group_upload_orbit(fs_group *g, uint64_t generation, int type_tag, uint32_t iter_bytes, const void *entries,
                               uint64_t orbit_size, uint64_t uncompressed_size, uint64_t period_maybe_zero)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_upload_orbit(r, generation, type_tag, iter_bytes, entries, orbit_size, uncompressed_size, period_maybe_zero));
    return 0;
}
uint32_t fs_group_upload_orbit_compressed(fs_group *g, uint64_t generation, int type_tag, uint32_t iter_bytes,
                                          const void *entries, uint64_t compressed_size, uint64_t uncompressed_size,
                                          uint64_t period_maybe_zero, const void *orbit_x_low, const void *orbit_y_low)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_upload_orbit_compressed(r, generation, type_tag, iter_bytes, entries, compressed_size, uncompressed_size,
                                           period_maybe_zero, orbit_x_low, orbit_y_low));
    return 0;
}
uint32_t fs_group_upload_la(fs_group *g, uint64_t generation, int type_tag, uint32_t iter_bytes, const void *las,
                            uint32_t n_las, const void *stages, uint32_t n_stages, int is_valid, int use_at,
                            const void *at_info)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_upload_la(r, generation, type_tag, iter_bytes, las, n_las, stages, n_stages, is_valid, use_at, at_info));
    return 0;
}
uint32_t fs_group_upload_bla(fs_group *g, int type_tag, const void *const *levels, const uint64_t *level_sizes,
                             int32_t n_levels, int32_t lm2)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_upload_bla(r, type_tag, levels, level_sizes, n_levels, lm2));
    return 0;
}
uint32_t fs_group_upload_orbit_scaled(fs_group *g, int type_tag, uint32_t iter_bytes, const void *entries_t,
                                      const void *entries_f32, uint64_t orbit_size, uint64_t period_maybe_zero)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_upload_orbit_scaled(r, type_tag, iter_bytes, entries_t, entries_f32, orbit_size, period_maybe_zero));
    return 0;
}

// ---- renders: one asynchronous launch per member, each on its own device and compute stream
uint32_t fs_group_render_lav2(fs_group *g, int type_tag, int mode, int parity, const void *coords, uint64_t n_iterations)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_render_lav2(r, type_tag, mode, parity, coords, n_iterations));
    return 0;
}
uint32_t fs_group_render_bla(fs_group *g, int type_tag, const void *coords, uint64_t n_iterations)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_render_bla(r, type_tag, coords, n_iterations));
    return 0;
}
uint32_t fs_group_render_scaled(fs_group *g, int type_tag, const void *coords, uint64_t n_iterations)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_render_scaled(r, type_tag, coords, n_iterations));
    return 0;
}
uint32_t fs_group_render_direct(fs_group *g, int type_tag, const void *coords, uint64_t n_iterations)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_render_direct(r, type_tag, coords, n_iterations));
    return 0;
}
uint32_t fs_group_clear(fs_group *g)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_clear(r));
    return 0;
}

uint32_t fs_group_sync(fs_group *g)
{
    for (fs_renderer *r : g->members)
        FSG_TRY(fs_sync_compute(r));
    if (g->post) {
        FSG_TRY(hipSetDevice(g->devices[0]));
        FSG_TRY(hipStreamSynchronize(g->post));
    }
    return 0;
}

// Host waits until the fs_group_render_current issued `frames_back` calls ago (0 = the latest, 1 = the one before) has
// delivered its frame (and reduction) to the caller's buffers.  A pipelined host loop: render k, render_current k,
// wait_current(1) -- frame k-1 is complete while frame k is still being rendered.
uint32_t fs_group_wait_current(fs_group *g, uint32_t frames_back)
{
    if (frames_back >= (uint32_t)fs_group::kSets || frames_back >= g->frames_posted)
        return frames_back >= g->frames_posted ? 0u : (uint32_t)hipErrorInvalidValue;
    // the set the frame posted `frames_back` calls ago went through, counted back from the set the NEXT frame lands in (`cur`
    // is what fs_group_render_current itself rotates, so the two can never disagree -- also after a second
    // fs_group_init_memory, which starts again at set 0)
    const int b = (g->cur + 2 * fs_group::kSets - 1 - (int)frames_back) % fs_group::kSets;
    FSG_TRY(hipSetDevice(g->devices[0]));
    return (uint32_t)hipEventSynchronize(g->ev_done[b]);
}

// Gather + RenderCurrent: slices -> device 0 over xGMI, row order restored, min / max / sum, D2H of the padded frame.
// Asynchronous: on device 0's POST stream, behind the members' kernels on the device (events, no host round trip), so the
// members' NEXT frame can be launched right away and runs while this one is delivered (fs_group_wait_current /
// fs_group_sync wait); iter_buffer / reduction may be NULL.
// The progressive form: a snapshot.  Nothing here waits for a kernel: every member's slice is copied AS IT IS NOW on that
// member's display stream (high priority, GPU_Render.cu:247-267) into a buffer set of its own on device 0, where device 0's
// display stream puts the rows in order, colours them (member 0's palette), reduces and copies to the host.  Peer copies
// for every transport (an RCCL exchange on a second stream of the same communicator would queue behind the frame gather).
static uint32_t group_current_progressive(fs_group *g, uint64_t n_iterations, void *iter_buffer, fs_color16 *color_buffer,
                                          fs_reduction *reduction)
{
    const uint32_t world = (uint32_t)g->members.size();
    const size_t sb = g->slice_bytes();
    const size_t frame_rows = ((size_t)g->height + 7u) / 8u * 8u;
    const size_t frame_bytes = frame_rows * g->rounded_width * g->iter_bytes;
    hipStream_t d0 = (hipStream_t)fs_display_stream(g->members[0]);
    FSG_TRY(hipSetDevice(g->devices[0]));
    if (!g->prog_gathered) {
        FSG_TRY(hipMalloc(&g->prog_gathered, sb * world));
        FSG_TRY(hipMemset(g->prog_gathered, 0, sb * world));
        FSG_TRY(hipMalloc(&g->prog_frame, frame_bytes));
        FSG_TRY(hipMemset(g->prog_frame, 0, frame_bytes));
        FSG_TRY(hipMalloc((void **)&g->prog_colors, fs_color_buffer_elements(g->members[0]) * sizeof(fs_color16)));
        FSG_TRY(hipMalloc((void **)&g->prog_reduction, sizeof(fs_reduction)));
        // (the two clears run on the default stream, which the display streams do not wait for: without this wait they could
        // land on top of the snapshot the copies below assemble -- a first snapshot that came back partly zero, once in a few
        // runs of the whole test suite)
        FSG_TRY(hipStreamSynchronize(nullptr));
    }
    // member 0 renders straight into slot 0 of the set in rotation
    FSG_TRY(hipMemcpyAsync(g->prog_gathered, g->gathered[g->cur], sb, hipMemcpyDeviceToDevice, d0));
    for (uint32_t r = 1; r < world; r++) {
        hipStream_t dr = (hipStream_t)fs_display_stream(g->members[r]);
        FSG_TRY(hipSetDevice(g->devices[r]));
        FSG_TRY(hipMemcpyPeerAsync((char *)g->prog_gathered + sb * r, g->devices[0], g->slices[g->cur][r], g->devices[r], sb, dr));
        hipEvent_t done;
        FSG_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        FSG_TRY(hipEventRecord(done, dr));
        FSG_TRY(hipSetDevice(g->devices[0]));
        FSG_TRY(hipStreamWaitEvent(d0, done, 0));
        FSG_TRY(hipEventDestroy(done));
    }
    FSG_TRY(hipSetDevice(g->devices[0]));
    fsk_gather_rows(g->prog_gathered, g->prog_frame, g->index, g->rounded_width * g->iter_bytes, g->height, d0);
    FSG_TRY(hipGetLastError());
    if (color_buffer)
        FSG_TRY(fs_colorize_frame(g->members[0], g->prog_frame, n_iterations, g->prog_colors, color_buffer, d0));
    if (reduction) {
        g->prog_seed = fs_reduction{g->iter_bytes == 8 ? ~0ull : 0xFFFFFFFFull, 0, 0};
        FSG_TRY(hipMemcpyAsync(g->prog_reduction, &g->prog_seed, sizeof(fs_reduction), hipMemcpyHostToDevice, d0));
        fsk_reduce(g->prog_frame, g->iter_bytes == 8, g->rounded_width, g->width, g->height, g->prog_reduction, d0);
        FSG_TRY(hipGetLastError());
        FSG_TRY(hipMemcpyAsync(reduction, g->prog_reduction, sizeof(fs_reduction), hipMemcpyDefault, d0));
    }
    if (iter_buffer)
        FSG_TRY(hipMemcpyAsync(iter_buffer, g->prog_frame, frame_bytes, hipMemcpyDefault, d0));
    return 0;
}

uint32_t fs_group_sync_display(fs_group *g)
{
    // the members' display streams first (their copies feed device 0's), then device 0's
    for (size_t r = g->members.size(); r-- > 0;)
        FSG_TRY(fs_sync_display(g->members[r]));
    return 0;
}

uint32_t fs_group_render_current(fs_group *g, uint64_t n_iterations, void *iter_buffer, fs_reduction *reduction)
{
    return fs_group_render_current_colors(g, n_iterations, iter_buffer, nullptr, reduction, 0);
}

// The direct host path (round 6).  The gather funnels the whole frame through device 0 and its ONE PCIe link -- 531 MB = 9.5 ms per
// frame at BASELINE's 8-GPU configuration against a 5.9-ms kernel per member.  Here every member copies its own bands straight to
// their rows of the caller's frame over its OWN link (fs_copy_bands_to_host: one 2-D copy, destination pitch = band stride), on a copy
// stream of its own behind its kernel, while it already renders the next frame into its other slice.  What still needs the frame
// on one device -- colours, min / max / sum -- keeps the gather (66 MB of Color16 instead of 531 MB of counts at that configuration).
static uint32_t group_direct_copies(fs_group *g, int b, void *iter_buffer, hipStream_t p0)
{
    const uint32_t world = (uint32_t)g->members.size();
    if (g->copy_streams.size() != world) { // first use: one copy stream and three events per member, on its device
        g->copy_streams.assign(world, nullptr);
        g->ev_kernel.assign(world, nullptr);
        for (int k = 0; k < 2; k++)
            g->ev_copied[k].assign(world, nullptr);
        for (uint32_t r = 0; r < world; r++) {
            FSG_TRY(hipSetDevice(g->devices[r]));
            FSG_TRY(hipStreamCreateWithFlags(&g->copy_streams[r], hipStreamNonBlocking));
            FSG_TRY(hipEventCreateWithFlags(&g->ev_kernel[r], hipEventDisableTiming));
            for (int k = 0; k < 2; k++)
                FSG_TRY(hipEventCreateWithFlags(&g->ev_copied[k][r], hipEventDisableTiming));
        }
    }
    for (uint32_t r = 0; r < world; r++) {
        const void *src = r == 0 ? g->gathered[b] : g->slices[b][r];
        FSG_TRY(hipSetDevice(g->devices[r]));
        FSG_TRY(hipEventRecord(g->ev_kernel[r], (hipStream_t)fs_compute_stream(g->members[r])));
        FSG_TRY(hipStreamWaitEvent(g->copy_streams[r], g->ev_kernel[r], 0));
        FSG_TRY(fs_copy_bands_to_host(g->members[r], src, iter_buffer, g->copy_streams[r]));
        FSG_TRY(hipEventRecord(g->ev_copied[b][r], g->copy_streams[r]));
        g->copied_recorded[b][r] = 1;
    }
    // "the frame is in the caller's buffer" (ev_done) = every member's copy has landed: the post stream collects them
    FSG_TRY(hipSetDevice(g->devices[0]));
    for (uint32_t r = 0; r < world; r++)
        FSG_TRY(hipStreamWaitEvent(p0, g->ev_copied[b][r], 0));
    return 0;
}

uint32_t fs_group_set_host_path(fs_group *g, int host_path)
{
    if (host_path != 0 && host_path != 1)
        return (uint32_t)hipErrorInvalidValue;
    g->host_path = host_path;
    return 0;
}
int fs_group_host_path(const fs_group *g) { return g->host_path; }

uint32_t fs_group_render_current_colors(fs_group *g, uint64_t n_iterations, void *iter_buffer, fs_color16 *color_buffer,
                                        fs_reduction *reduction, int progressive)
{
    const uint32_t world = (uint32_t)g->members.size();
    if (!g->gathered[0])
        return 0; // memory not initialised: silent, like GPURenderer::RenderCurrent
    if (progressive)
        return group_current_progressive(g, n_iterations, iter_buffer, color_buffer, reduction);
    const size_t sb = g->slice_bytes();
    const int b = g->cur;
    void *const gathered = g->gathered[b];
    void *const frame = g->frame[b];
    hipStream_t s0 = (hipStream_t)fs_compute_stream(g->members[0]);
    hipStream_t p0 = g->post;
    const bool direct = g->host_path == 1 && iter_buffer != nullptr;
    // under the direct path the frame is only brought together on device 0 when something needs it there
    const bool need_gather = !direct || color_buffer != nullptr || reduction != nullptr;
    FSG_TRY(hipSetDevice(g->devices[0]));
    // member 0's slice is in slot 0 once its kernel has finished
    FSG_TRY(hipEventRecord(g->ev_rendered[b], s0));
    FSG_TRY(hipStreamWaitEvent(p0, g->ev_rendered[b], 0));
    FSG_TRY(hipEventRecord(g->ev_a, p0));
    if (direct)
        FSG_TRY(group_direct_copies(g, b, iter_buffer, p0));
    if (need_gather && world > 1 && g->transport == 0) {
        Rccl &q = rccl();
        if (q.GroupStart() != 0)
            return FS_ERR_7;
        // (every exit below closes the group: a thread left inside an open ncclGroup corrupts its later RCCL calls)
        bool ok = true;
        for (uint32_t r = 1; r < world && ok; r++) {
            // rank 0 receives slice r on the post stream -- behind the k_gather_rows that last read this set's slots (same
            // stream), so a slot is never overwritten while it is still being read; rank r sends behind ITS kernel
            ok = q.Recv((char *)gathered + sb * r, sb, kNcclUint8, (int)r, g->comms[0], p0) == 0 &&
                 q.Send(g->slices[b][r], sb, kNcclUint8, 0, g->comms[r], (hipStream_t)fs_compute_stream(g->members[r])) == 0;
        }
        if (q.GroupEnd() != 0 || !ok)
            return FS_ERR_7;
    } else if (need_gather && world > 1) {
        for (uint32_t r = 1; r < world; r++) {
            // the copy runs on the SENDER's stream (ordered behind its kernel); the post stream then waits for it on the
            // device.  The copy writes gather slot r of this set, which the k_gather_rows of the frame that used the set
            // last may still be reading (this call is asynchronous; a fast member can be frames ahead of a slow one): the
            // sender first waits for that set's ev_consumed (a never-recorded event does not wait).
            hipStream_t sr = (hipStream_t)fs_compute_stream(g->members[r]);
            FSG_TRY(hipSetDevice(g->devices[r]));
            FSG_TRY(hipStreamWaitEvent(sr, g->ev_consumed[b], 0));
            FSG_TRY(hipMemcpyPeerAsync((char *)gathered + sb * r, g->devices[0], g->slices[b][r], g->devices[r], sb, sr));
            hipEvent_t done;
            FSG_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
            FSG_TRY(hipEventRecord(done, sr));
            FSG_TRY(hipSetDevice(g->devices[0]));
            FSG_TRY(hipStreamWaitEvent(p0, done, 0));
            FSG_TRY(hipEventDestroy(done)); // released once the recorded work has completed
        }
    }
    FSG_TRY(hipSetDevice(g->devices[0]));
    if (need_gather) {
        fsk_gather_rows(gathered, frame, g->index, g->rounded_width * g->iter_bytes, g->height, p0);
        FSG_TRY(hipGetLastError());
    }
    FSG_TRY(hipEventRecord(g->ev_consumed[b], p0)); // this set's gather slots may be written again
    FSG_TRY(hipEventRecord(g->ev_b, p0));
    // colours of the whole frame (antialias + palette) on device 0, behind the row order: member 0 holds the palette and the
    // whole-frame geometry; frames in flight share its colour buffer one after the other (one post stream)
    if (color_buffer)
        FSG_TRY(fs_colorize_frame(g->members[0], frame, n_iterations, nullptr, color_buffer, p0));
    if (reduction) {
        g->reduce_seed = fs_reduction{g->iter_bytes == 8 ? ~0ull : 0xFFFFFFFFull, 0, 0};
        FSG_TRY(hipMemcpyAsync(g->reduction, &g->reduce_seed, sizeof(fs_reduction), hipMemcpyHostToDevice, p0));
        fsk_reduce(frame, g->iter_bytes == 8, g->rounded_width, g->width, g->height, g->reduction, p0);
        FSG_TRY(hipGetLastError());
        FSG_TRY(hipMemcpyAsync(reduction, g->reduction, sizeof(fs_reduction), hipMemcpyDefault, p0));
    }
    if (iter_buffer && !direct) {
        const size_t frame_rows = ((size_t)g->height + 7u) / 8u * 8u;
        FSG_TRY(hipMemcpyAsync(iter_buffer, frame, frame_rows * g->rounded_width * g->iter_bytes, hipMemcpyDefault, p0));
    }
    FSG_TRY(hipEventRecord(g->ev_done[b], p0));
    g->frames_posted++;
    // the next frame lands in the other set: member 0 renders into ITS slot 0 -- once the frame that used that set last has
    // been put in row order (its gather slots are free then) -- and every other member into its other slice -- once the direct
    // copy of the frame that used it last has left it (a gather reads a slice on the member's own compute stream: in order)
    g->cur = (b + 1) % fs_group::kSets;
    FSG_TRY(hipStreamWaitEvent(s0, g->ev_consumed[g->cur], 0));
    if (!g->copied_recorded[g->cur].empty() && g->copied_recorded[g->cur][0])
        FSG_TRY(hipStreamWaitEvent(s0, g->ev_copied[g->cur][0], 0));
    FSG_TRY(fs_set_external_iter_buffer(g->members[0], g->gathered[g->cur], sb));
    for (uint32_t r = 1; r < world; r++) {
        FSG_TRY(hipSetDevice(g->devices[r]));
        if (g->copied_recorded[g->cur][r])
            FSG_TRY(hipStreamWaitEvent((hipStream_t)fs_compute_stream(g->members[r]), g->ev_copied[g->cur][r], 0));
        FSG_TRY(fs_set_external_iter_buffer(g->members[r], g->slices[g->cur][r], sb));
    }
    return 0;
}

float fs_group_gather_ms(fs_group *g)
{
    float ms = -1.0f;
    if (g->ev_a && hipSetDevice(g->devices[0]) == hipSuccess && hipEventSynchronize(g->ev_b) == hipSuccess &&
        hipEventElapsedTime(&ms, g->ev_a, g->ev_b) == hipSuccess)
        g->last_gather_ms = ms;
    return g->last_gather_ms;
}

} // extern "C"
