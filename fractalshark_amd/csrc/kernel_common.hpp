// kernel_common.hpp -- small device helpers shared by the kernel translation units (kernels*.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace {

// Global row of local row L under the band layout (fs_set_row_bands).
__device__ __forceinline__ uint32_t global_row(const FsFrame &f, uint32_t L)
{
    const uint32_t k = L / f.band_rows;
    const uint32_t rr = L - k * f.band_rows;
    return f.band_first + k * f.band_stride + rr;
}

// Pixel of this lane under the square-tile mapping: a wave covers an 8 x 8 pixel tile (4 tiles side by side per 256-thread
// block) instead of 64 pixels of one row.  Iteration counts are correlated in two dimensions, so a compact tile keeps the
// lanes of a wave closer together in how long they run and in which orbit position they read.
__device__ __forceinline__ void tile_pixel(uint32_t &X, uint32_t &L)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#ifdef FS_TILE_STRIDE
    // (A/B build, round 6: workgroups take the tile blocks in a strided order -- block v of the launch renders block v * FS_TILE_STRIDE mod N --
    // so that a region of long pixels is spread over the launch instead of ending it)
    const uint32_t nb = gridDim.x * gridDim.y;
    const uint32_t v = (uint32_t)(((uint64_t)(blockIdx.y * gridDim.x + blockIdx.x) * (uint64_t)(FS_TILE_STRIDE)) % nb);
    const uint32_t bx = v % gridDim.x, by = v / gridDim.x;
#else
    const uint32_t bx = blockIdx.x, by = blockIdx.y;
#endif
    X = (bx * (blockDim.x >> 6) + wave) * 8u + (lane & 7u);
    L = by * 8u + (lane >> 3);
}

// ... or the tile a tile order names for this wave (FsLav2ArgsT::tile_order: row-major tile number of the local buffer, all ones = none)
__device__ __forceinline__ void ordered_tile_pixel(const uint32_t *__restrict__ order, uint32_t tiles_x, uint32_t &X, uint32_t &L)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)order[w]);
    if (tile != 0xFFFFFFFFu) {
        const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        X = tx * 8u + (lane & 7u);
        L = ty * 8u + (lane >> 3);
    } else {
        X = 0xFFFFFFFFu, L = 0xFFFFFFFFu;
    }
}

// ... or, with a recorded pixel order (kernels_order.hip): lane s of the launch takes element order[s] of the local iteration
// buffer.  Elements in the padding (column >= width, row >= local_rows) are not pixels: the caller's bounds test drops them.
__device__ __forceinline__ void ordered_pixel(const FsFrame &f, const uint32_t *__restrict__ order, uint32_t &X, uint32_t &L)
{
    const uint32_t slot = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t n = f.rounded_width * ((f.local_rows + 7u) & ~7u);
    if (slot < n) {
        const uint32_t id = order[slot];
        L = id / f.rounded_width;
        X = id - L * f.rounded_width;
    } else {
        X = 0xFFFFFFFFu, L = 0xFFFFFFFFu;
    }
}

// One result into the iteration buffer: OutputIterMatrix[ConvertLocToIndex(X, Y, width)] (GPU_Render.cu:73-79) for
// IterType = uint32_t or uint64_t.  Counts are computed in 32 bits (the ABI refuses n_iterations >= 2^32).
__device__ __forceinline__ void store_iter(uint32_t *out, const FsFrame &f, uint32_t L, uint32_t X, uint32_t v)
{
    const size_t idx = (size_t)L * f.rounded_width + X;
    if (f.iter_u64)
        reinterpret_cast<uint64_t *>(out)[idx] = v;
    else
        out[idx] = v;
}

// 64-bit counting: the full value into a uint64_t buffer (IterType = uint64_t); a 4-byte buffer can only be paired with
// these kernels through the FS_VARIANT_FLAG_WIDE test switch at caps below 2^32, where the low word is the whole count
__device__ __forceinline__ void store_iter(uint32_t *out, const FsFrame &f, uint32_t L, uint32_t X, uint64_t v)
{
    const size_t idx = (size_t)L * f.rounded_width + X;
    if (f.iter_u64)
        reinterpret_cast<uint64_t *>(out)[idx] = v;
    else
        out[idx] = (uint32_t)v;
}

// The iteration cap as the kernel's counter type: IterT = uint32_t takes the low word (the host only launches such a kernel
// for caps below 2^32), IterT = uint64_t the full value (IterType = uint64_t of the reference's templates with a cap the
// 32-bit counters cannot hold; the iteration buffer then holds uint64_t elements).
template <class IterT> __device__ __forceinline__ IterT iter_cap(uint32_t lo, uint32_t hi)
{
    return sizeof(IterT) == 8 ? (IterT)(((uint64_t)hi << 32) | lo) : (IterT)lo;
}

__device__ __forceinline__ void add_stats(uint64_t *stats, uint64_t at, uint64_t la, uint64_t pt, uint64_t px)
{
    // stats[4]: lane slots the wave occupied in the perturbation loop = 64 x (longest lane); with [2] it gives
    // the SIMD lane utilisation of the loop.
    uint64_t mx = pt;
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    // one atomic per wave per counter
    for (int off = 32; off > 0; off >>= 1) {
        at += __shfl_down(at, off);
        la += __shfl_down(la, off);
        pt += __shfl_down(pt, off);
        px += __shfl_down(px, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long *)&stats[0], (unsigned long long)at);
        atomicAdd((unsigned long long *)&stats[1], (unsigned long long)la);
        atomicAdd((unsigned long long *)&stats[2], (unsigned long long)pt);
        atomicAdd((unsigned long long *)&stats[3], (unsigned long long)px);
        atomicAdd((unsigned long long *)&stats[4], (unsigned long long)(mx * 64u));
    }
}

} // namespace
