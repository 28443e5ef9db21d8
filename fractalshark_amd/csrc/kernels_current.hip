// kernels_current.hip -- the device half of GPURenderer::RenderCurrent (FractalSharkGpuLib/GPU_Render.cu:556-581):
// antialias + palette (antialiasing_kernel, AntialiasingKernel.cuh:3-71) and min / max / sum of the iteration buffer
// (max_kernel, ReductionKernels.cuh:73-142).  Both are one pass over the iteration buffer and HBM-bound: 4*AA^2 B read +
// 8 B written per colour pixel, 4 B read per iteration-buffer element (8 B for IterType = uint64_t).  The shape of both
// kernels is therefore "every lane issues the widest aligned load its row segment allows and nothing else gets in the
// way": one 16-byte load per AA row for AA = 4 (rows are multiples of 16 elements, so a 4-element group never
// straddles an alignment boundary), no per-element 64-bit division anywhere.
#include "kernel_common.hpp"

namespace {

// n % d for a 32-bit n and 2^8 <= d < 2^24, with the quotient estimated in binary32 and finished with integer
// compare-free corrections (all full-rate instructions; the compiler's generic 32-bit remainder costs about twice as
// much and the 64-bit one an order of magnitude more).
//   q_est = trunc(float(n) * float(1/d)): relative error <= 3 * 2^-24, q <= 2^32 / 2^8 = 2^24, so q_est is within
//   {q-2 .. q+1}.  With q' = max(q_est, 2) - 2 the remainder candidate n - q'*d lies in [0, 5d) subset of [0, 8d), and
//   three "r = min(r, r - k*d)" steps (unsigned: r - k*d wraps above r exactly when r < k*d) bring it into [0, d).
//   q' < 2^24 and d < 2^24, so the low 32 bits of q'*d come from the 24-bit multiplier.
struct FastMod {
    uint32_t d;
    float inv;
    uint32_t usable; // 1: the bounds above hold
};

__device__ __forceinline__ uint32_t fast_mod(uint32_t n, const FastMod &m)
{
    uint32_t q = (uint32_t)((float)n * m.inv);
    q = (q > 2u ? q : 2u) - 2u;
    uint32_t r = n - __umul24(q, m.d);
    uint32_t t = r - 4u * m.d;
    r = t < r ? t : r;
    t = r - 2u * m.d;
    r = t < r ? t : r;
    t = r - m.d;
    r = t < r ? t : r;
    return r;
}

template <class IterT> struct Vec;
template <> struct Vec<uint32_t> {
    using V4 = uint4;
    using V2 = uint2;
};
template <> struct Vec<uint64_t> {
    using V4 = ulonglong4;
    using V2 = ulonglong2;
};

// One colour pixel per lane, 64 consecutive pixels of a colour row per wave: with AA = 4 a wave reads 1 KiB of
// consecutive bytes per AA row (64 x uint4), fully coalesced.  Integer sums, so the order of accumulation is free.
template <class IterT, uint32_t AA, bool kFast>
__global__ void __launch_bounds__(256) k_antialias(const IterT *__restrict__ iters, uint32_t rounded_width,
                                                   fs_color16 *__restrict__ colors, const fs_color16 *__restrict__ pal,
                                                   uint32_t pal_iters, FastMod fm, uint32_t aux_depth, uint32_t color_w,
                                                   uint32_t color_h, uint64_t n_iterations)
{
    const uint32_t ox = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t oy = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (ox >= color_w || oy >= color_h)
        return;
    uint32_t acc_r = 0, acc_g = 0, acc_b = 0; // <= 16 x 65535 fits 32 bits
    const IterT *row = iters + (size_t)(oy * AA) * rounded_width + (size_t)ox * AA;
#pragma unroll
    for (uint32_t iy = 0; iy < AA; iy++, row += rounded_width) {
        IterT v[AA];
        if constexpr (AA == 4) {
            const auto q = *reinterpret_cast<const typename Vec<IterT>::V4 *>(row);
            v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
        } else if constexpr (AA == 2) {
            const auto q = *reinterpret_cast<const typename Vec<IterT>::V2 *>(row);
            v[0] = q.x, v[1] = q.y;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < AA; k++)
                v[k] = row[k];
        }
#pragma unroll
        for (uint32_t k = 0; k < AA; k++) {
            const IterT n = v[k];
            if (n < n_iterations) {
                uint32_t p;
                if constexpr (kFast)
                    p = fast_mod((uint32_t)(n >> aux_depth), fm);
                else
                    p = (uint32_t)((n >> aux_depth) % pal_iters);
                const uint2 c = *reinterpret_cast<const uint2 *>(pal + p); // one 8-byte gather {r|g<<16, b|a<<16}
                acc_r += c.x & 0xFFFFu;
                acc_g += c.x >> 16;
                acc_b += c.y & 0xFFFFu;
            }
        }
    }
    constexpr uint32_t total = AA * AA;
    uint2 o;
    o.x = (acc_r / total) | ((acc_g / total) << 16);
    o.y = (acc_b / total) | (65535u << 16);
    *reinterpret_cast<uint2 *>(colors + (size_t)oy * color_w + ox) = o;
}

// Min / max / sum of the valid (unpadded) part of the iteration buffer: max_kernel (ReductionKernels.cuh:73-142)
// without its unsynchronised output reset -- the host seeds {Min = numeric_limits<IterType>::max(), Max = 0, Sum = 0} on
// the stream before the launch.  2-D: x walks 4-element groups of a row (rows are multiples of 16 elements), y strides
// rows -- no division; wave shuffles, then one LDS hop so that a workgroup issues a single atomic triple.
template <class IterT>
__global__ void __launch_bounds__(256) k_reduce(const IterT *__restrict__ iters, uint32_t rounded_width, uint32_t width,
                                                uint32_t rows, fs_reduction *out)
{
    uint64_t mn = (uint64_t)(IterT)~(IterT)0, mx = 0, sum = 0;
    const uint32_t x0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (x0 < width) {
        const uint32_t nvalid = width - x0 < 4u ? width - x0 : 4u;
        const IterT *p = iters + (size_t)blockIdx.y * rounded_width + x0;
        const size_t stride = (size_t)gridDim.y * rounded_width;
        auto fold = [&](const typename Vec<IterT>::V4 &q) {
            const uint64_t v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (uint32_t k = 0; k < 4; k++)
                if (k < nvalid) {
                    mn = v[k] < mn ? v[k] : mn;
                    mx = v[k] > mx ? v[k] : mx;
                    sum += v[k];
                }
        };
        using V4 = typename Vec<IterT>::V4;
        uint32_t y = blockIdx.y;
        // four rows per trip: four independent 16-byte loads in flight per lane
        for (; y + 3u * gridDim.y < rows; y += 4u * gridDim.y, p += 4u * stride) {
            const V4 q0 = *reinterpret_cast<const V4 *>(p);
            const V4 q1 = *reinterpret_cast<const V4 *>(p + stride);
            const V4 q2 = *reinterpret_cast<const V4 *>(p + 2u * stride);
            const V4 q3 = *reinterpret_cast<const V4 *>(p + 3u * stride);
            fold(q0);
            fold(q1);
            fold(q2);
            fold(q3);
        }
        for (; y < rows; y += gridDim.y, p += stride)
            fold(*reinterpret_cast<const V4 *>(p));
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t omn = __shfl_down(mn, off), omx = __shfl_down(mx, off);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
        sum += __shfl_down(sum, off);
    }
    __shared__ uint64_t part[3][4];
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) {
        part[0][wave] = mn;
        part[1][wave] = mx;
        part[2][wave] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < 4; w++) {
            mn = part[0][w] < mn ? part[0][w] : mn;
            mx = part[1][w] > mx ? part[1][w] : mx;
            sum += part[2][w];
        }
        atomicMin((unsigned long long *)&out->Min, (unsigned long long)mn);
        atomicMax((unsigned long long *)&out->Max, (unsigned long long)mx);
        atomicAdd((unsigned long long *)&out->Sum, (unsigned long long)sum);
    }
}

template <class IterT, bool kFast>
void launch_antialias(const void *iters, uint32_t rounded_width, fs_color16 *colors, const fs_color16 *pal,
                      uint32_t pal_iters, FastMod fm, uint32_t aux_depth, uint32_t aa, uint32_t color_w, uint32_t color_h,
                      uint64_t n_iterations, hipStream_t s)
{
    const dim3 g((color_w + 63) / 64, (color_h + 3) / 4), b(256);
    const IterT *it = (const IterT *)iters;
    switch (aa) {
    case 1:
        hipLaunchKernelGGL((k_antialias<IterT, 1, kFast>), g, b, 0, s, it, rounded_width, colors, pal, pal_iters, fm,
                           aux_depth, color_w, color_h, n_iterations);
        break;
    case 2:
        hipLaunchKernelGGL((k_antialias<IterT, 2, kFast>), g, b, 0, s, it, rounded_width, colors, pal, pal_iters, fm,
                           aux_depth, color_w, color_h, n_iterations);
        break;
    case 3:
        hipLaunchKernelGGL((k_antialias<IterT, 3, kFast>), g, b, 0, s, it, rounded_width, colors, pal, pal_iters, fm,
                           aux_depth, color_w, color_h, n_iterations);
        break;
    default:
        hipLaunchKernelGGL((k_antialias<IterT, 4, kFast>), g, b, 0, s, it, rounded_width, colors, pal, pal_iters, fm,
                           aux_depth, color_w, color_h, n_iterations);
        break;
    }
}

} // namespace

void fsk_antialias(const void *iters, int iter_u64, uint32_t rounded_width, fs_color16 *colors, const fs_color16 *pal,
                   uint32_t pal_iters, uint32_t aux_depth, uint32_t aa, uint32_t color_w, uint32_t color_h,
                   uint64_t n_iterations, hipStream_t s)
{
    FastMod fm;
    fm.d = pal_iters;
    fm.inv = pal_iters ? (float)(1.0 / (double)pal_iters) : 0.0f;
    fm.usable = pal_iters >= 256u && pal_iters < (1u << 24);
    if (iter_u64) // 64-bit counts: the shifted value may exceed 32 bits, generic remainder
        launch_antialias<uint64_t, false>(iters, rounded_width, colors, pal, pal_iters, fm, aux_depth, aa, color_w, color_h,
                                          n_iterations, s);
    else if (fm.usable)
        launch_antialias<uint32_t, true>(iters, rounded_width, colors, pal, pal_iters, fm, aux_depth, aa, color_w, color_h,
                                         n_iterations, s);
    else
        launch_antialias<uint32_t, false>(iters, rounded_width, colors, pal, pal_iters, fm, aux_depth, aa, color_w, color_h,
                                          n_iterations, s);
}

void fsk_reduce(const void *iters, int iter_u64, uint32_t rounded_width, uint32_t width, uint32_t rows,
                fs_reduction *out, hipStream_t s)
{
    if (!rows || !width)
        return;
    const uint32_t gx = (width + 1023u) / 1024u;
    // enough workgroups to fill 256 CUs several times over, few enough that the atomics (one triple per workgroup) stay
    // in the low thousands
    uint32_t gy = (2048u + gx - 1u) / gx;
    gy = gy < rows ? gy : rows;
    const dim3 g(gx, gy), b(256);
    if (iter_u64)
        hipLaunchKernelGGL(k_reduce<uint64_t>, g, b, 0, s, (const uint64_t *)iters, rounded_width, width, rows, out);
    else
        hipLaunchKernelGGL(k_reduce<uint32_t>, g, b, 0, s, (const uint32_t *)iters, rounded_width, width, rows, out);
}

// ------------------------------------------------------------------------------------------------
// Row reassembly of the multi-GPU tiler (csrc/group.cpp): out row y = gathered row index[y].  One 16-byte vector per
// lane, rows are multiples of 16 elements: a pure HBM copy (read + write of the frame once).
namespace {
__global__ void __launch_bounds__(256) k_gather_rows(const uint4 *__restrict__ in, uint4 *__restrict__ out,
                                                     const uint32_t *__restrict__ index, uint32_t row_vec4, uint32_t rows)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= row_vec4)
        return;
    for (uint32_t y = blockIdx.y; y < rows; y += gridDim.y)
        out[(size_t)y * row_vec4 + x] = in[(size_t)index[y] * row_vec4 + x];
}
} // namespace

void fsk_gather_rows(const void *in, void *out, const uint32_t *index, uint32_t row_bytes, uint32_t rows, hipStream_t s)
{
    const uint32_t v4 = row_bytes / 16u;
    if (!v4 || !rows)
        return;
    const uint32_t gx = (v4 + 255u) / 256u;
    uint32_t gy = (4096u + gx - 1u) / gx;
    gy = gy < rows ? gy : rows;
    hipLaunchKernelGGL(k_gather_rows, dim3(gx, gy), dim3(256), 0, s, (const uint4 *)in, (uint4 *)out, index, v4, rows);
}

// ------------------------------------------------------------------------------------------------
// "Long tiles first": the launch order of a frame's 8 x 8 tiles from a probe of their centre pixels (fs_render_bla).
// One workgroup: every thread counts the long tiles of its contiguous chunk, an LDS scan ranks the chunks, then the long
// tiles are written in tile order, followed by the others in tile order (a stable two-way partition), followed by the
// "no tile" filler for the launch's surplus waves.
namespace {
__global__ void __launch_bounds__(1024) k_tile_order(const uint32_t *__restrict__ probe, uint32_t pitch, uint32_t tiles_x,
                                                     uint32_t tiles_y, uint32_t threshold, uint32_t *__restrict__ order,
                                                     uint32_t n_slots)
{
    __shared__ uint32_t s_cnt[1024];
    const uint32_t n = tiles_x * tiles_y, chunk = (n + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * chunk < n ? threadIdx.x * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    // a tile counts as long when its own centre or a neighbour's is still running: the set's boundary is ragged, and a
    // tile next to an interior one often holds interior pixels away from its centre
    auto is_long = [&](uint32_t i) {
        const uint32_t tx = i % tiles_x, ty = i / tiles_x;
        const uint32_t x0 = tx ? tx - 1 : 0, x1 = tx + 1 < tiles_x ? tx + 1 : tx, y0 = ty ? ty - 1 : 0,
                       y1 = ty + 1 < tiles_y ? ty + 1 : ty;
        for (uint32_t y = y0; y <= y1; y++)
            for (uint32_t x = x0; x <= x1; x++)
                if (probe[(size_t)y * pitch + x] >= threshold)
                    return true;
        return false;
    };
    uint32_t mine = 0;
    for (uint32_t i = lo; i < hi; i++)
        mine += is_long(i) ? 1u : 0u;
    s_cnt[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) { // inclusive Hillis-Steele scan
        const uint32_t v = threadIdx.x >= d ? s_cnt[threadIdx.x - d] : 0u;
        __syncthreads();
        s_cnt[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t total_long = s_cnt[1023];
    uint32_t at_long = s_cnt[threadIdx.x] - mine; // long tiles before this chunk
    uint32_t at_short = lo - at_long;            // short tiles before this chunk
    // One long tile per WORKGROUP while there are three short ones to go with each (round 5): the four waves of a workgroup
    // share a CU, and never-escaping waves that share one were measured to slow each other down (tools/c2_cu_pace.py: the
    // long waves of one workgroup speed up and slow down together from launch to launch).  Long tile k goes to slot 4 k,
    // the first 3 k short tiles fill the slots next to them, the rest follow.  Otherwise: the long tiles first, as before.
    const bool spread = 3u * total_long <= n - total_long;
    for (uint32_t i = lo; i < hi; i++) {
        if (is_long(i)) {
            order[spread ? 4u * at_long : at_long] = i;
            at_long++;
        } else {
            const uint32_t sidx = at_short++;
            order[!spread ? total_long + sidx : (sidx < 3u * total_long ? 4u * (sidx / 3u) + 1u + sidx % 3u : total_long + sidx)] = i;
        }
    }
    for (uint32_t i = n + threadIdx.x; i < n_slots; i += 1024u)
        order[i] = 0xFFFFFFFFu;
    if (threadIdx.x == 0) // the number of long tiles (their waves ask for issue priority); the top bit: one per workgroup
        order[n_slots] = total_long | (spread ? 0x80000000u : 0u);
}
} // namespace

void fsk_tile_order(const uint32_t *probe, uint32_t probe_pitch, uint32_t tiles_x, uint32_t tiles_y, uint32_t threshold,
                    uint32_t *order, uint32_t n_slots, hipStream_t s)
{
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, probe, probe_pitch, tiles_x, tiles_y, threshold, order, n_slots);
}

// ------------------------------------------------------------------------------------------------
// "Longest tiles first" from recorded costs (k_lav2_hdr32_fast writes one word per 8 x 8 tile: its longest lane's step
// count; the next frame of the same geometry is launched in this order, fs_render_lav2).  A frame ends one long wave after
// its LAST wave was dispatched, and the waves of one frame differ 2.5x in length: started first,
// the long ones finish while the chip is still full.
// A stable counting sort into 256 cost classes, highest first: class = 255 - (exponent and three mantissa bits of the cost
// as a float) -- 8 classes per octave over the whole 32-bit range, so no pass over the costs is needed to find their
// range.  Tiles of one class keep their raster order: neighbours (whose costs are alike) still start together and walk the
// same stretch of the orbit.  Three small launches (the first version was ONE workgroup and took 0.65 ms at 129 600 tiles
// -- a serial chain of dependent loads per thread -- in front of every frame):
//   k_cost_hist     kSortWaves waves, each over a contiguous chunk of tiles: tiles per class, per wave
//   k_cost_scan     one workgroup: exclusive scan of the (class, wave) counts in class-major order
//   k_cost_scatter  the same waves: every tile to  offset[class][wave] + its rank among the wave's earlier tiles of the class
// Inside a wave a group of 64 tiles is ranked with ballots: eight votes give each lane the mask of the lanes that share
// its class (its rank = the set bits below it), and an LDS word per class hands that mask to the lane that keeps the
// class's running offset.
namespace {
constexpr uint32_t kSortWaves = 128;  // waves over the tile array (4 per workgroup)
constexpr uint32_t kCostClasses = 256;

__device__ __forceinline__ uint32_t cost_class(uint32_t c)
{
    // monotone in c: 0 -> 255 (last), 1 -> 255, 2 -> 247, ..., 2^32 - 1 -> 0 (float conversion rounds to nearest: monotone)
    const uint32_t q = c ? (__float_as_uint((float)c) >> 20) - (127u << 3) : 0u;
    return kCostClasses - 1u - (q < kCostClasses ? q : kCostClasses - 1u);
}

__device__ __forceinline__ uint32_t sort_chunk(uint32_t n) // tiles per wave: a whole number of 64-tile groups
{
    return ((n + kSortWaves - 1u) / kSortWaves + 63u) / 64u * 64u;
}

// mask of the lanes (of the 64 active ones) whose 8-bit class equals this lane's
__device__ __forceinline__ uint64_t class_peers(uint32_t cls)
{
    uint64_t peers = ~0ull;
#pragma unroll
    for (uint32_t bit = 0; bit < 8; bit++) {
        const uint64_t b = __builtin_amdgcn_ballot_w64(((cls >> bit) & 1u) != 0u);
        peers &= ((cls >> bit) & 1u) ? b : ~b;
    }
    return peers;
}

// One pass of a wave over its chunk.  kScatter == false: counts[class * kSortWaves + wave] = tiles of the class in the
// chunk; true: order[offsets[class * kSortWaves + wave] + rank] = tile.
template <bool kScatter>
__global__ void __launch_bounds__(256) k_cost_pass(const uint32_t *__restrict__ cost, uint32_t n, uint32_t *__restrict__ table,
                                                   uint32_t *__restrict__ order, uint32_t n_slots)
{
    __shared__ uint64_t s_mask[4][kCostClasses];
    __shared__ uint32_t s_base[4][kCostClasses];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, w = blockIdx.x * 4u + wv;
    // (volatile: the lanes of a wave talk to each other through these words; a wave's LDS operations execute in order)
    volatile uint64_t *mask = s_mask[wv];
    volatile uint32_t *base = s_base[wv];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++)
        base[lane + 64u * j] = kScatter ? table[(lane + 64u * j) * kSortWaves + w] : 0u;
    const uint32_t chunk = sort_chunk(n);
    const uint32_t lo = w * chunk < n ? w * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    for (uint32_t g = lo; g < hi; g += 64u) {
        const uint32_t tile = g + lane;
        const bool have = tile < hi;
        // (a lane beyond the end votes in a class of its own kind: class 0 with the `have` bit cleared never matches a tile)
        const uint32_t cls = have ? cost_class(cost[tile]) : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
            mask[lane + 64u * j] = 0ull;
        const uint64_t live = __builtin_amdgcn_ballot_w64(have);
        const uint64_t peers = class_peers(cls) & live;
        if (have)
            mask[cls] = peers; // (every lane of a class writes the same word)
        if (kScatter && have) {
            const uint64_t below = peers & ((1ull << lane) - 1ull);
            order[base[cls] + (uint32_t)__popcll(below)] = tile;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
            base[lane + 64u * j] = base[lane + 64u * j] + (uint32_t)__popcll(mask[lane + 64u * j]);
    }
    if (!kScatter) {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
            table[(lane + 64u * j) * kSortWaves + w] = base[lane + 64u * j];
    } else {
        // the launch's surplus waves render nothing
        for (uint32_t i = n + blockIdx.x * 256u + threadIdx.x; i < n_slots; i += gridDim.x * 256u)
            order[i] = 0xFFFFFFFFu;
        if (blockIdx.x == 0 && threadIdx.x == 0)
            order[n_slots] = 0u;
    }
}

// exclusive scan of the kCostClasses x kSortWaves counts, in place (class-major: all of class 0, then class 1, ...)
__global__ void __launch_bounds__(1024) k_cost_scan(uint32_t *__restrict__ table)
{
    constexpr uint32_t kPer = kCostClasses * kSortWaves / 1024u;
    __shared__ uint32_t s_scan[1024];
    const uint32_t t = threadIdx.x;
    uint32_t v[kPer], mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        v[k] = table[t * kPer + k];
        mine += v[k];
    }
    s_scan[t] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t a = t >= d ? s_scan[t - d] : 0u;
        __syncthreads();
        s_scan[t] += a;
        __syncthreads();
    }
    uint32_t run = s_scan[t] - mine;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        table[t * kPer + k] = run;
        run += v[k];
    }
}
static_assert(kCostClasses * kSortWaves % 1024u == 0, "the scan gives every thread the same number of counters");
} // namespace

uint32_t fsk_tile_order_work_words(uint32_t) { return kCostClasses * kSortWaves; }

void fsk_tile_order_by_cost(const uint32_t *cost, uint32_t n_tiles, uint32_t *tmp, uint32_t *order, uint32_t n_slots,
                            hipStream_t s)
{
    hipLaunchKernelGGL(k_cost_pass<false>, dim3(kSortWaves / 4), dim3(256), 0, s, cost, n_tiles, tmp, order, n_slots);
    hipLaunchKernelGGL(k_cost_scan, dim3(1), dim3(1024), 0, s, tmp);
    hipLaunchKernelGGL(k_cost_pass<true>, dim3(kSortWaves / 4), dim3(256), 0, s, cost, n_tiles, tmp, order, n_slots);
}
