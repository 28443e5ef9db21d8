// kernels_order.hip -- "pixels in the order of the previous frame's counts" for the LAv2 kernels whose waves otherwise wait for
// their slowest lane (k_lav2_lit<double>, k_lav2_2x32: View-14-class frames are 1700 AT iterations per pixel with a spread inside
// an 8 x 8 tile that leaves a quarter of the lane slots idle -- lane occupancy 0.74, profiles/r05_*).
//
// The same idea as the tuned kernel's "tiles launched longest first from the costs the previous frame recorded" (appendix 5.4), one
// level down: after a frame, the iteration buffer itself is the record.  Its elements are sorted by count, descending (device
// radix sort of (count, buffer position) pairs -- hipCUB, a setup step that runs once per view, not the hot path); the next frame
// of the same view hands lane s of the launch the pixel at order[s], so the 64 lanes of a wave hold pixels that ran equally long
// last time, and the longest ones are dispatched first.  Which lane renders which pixel changes no pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <hipcub/hipcub.hpp>

#include "kernels.h"

namespace {
__global__ void k_iota(uint32_t *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = i;
}
} // namespace

size_t fsk_pixel_order_temp_bytes(uint32_t n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                                       (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, 0, 32, (hipStream_t) nullptr);
    return bytes;
}

// counts: the iteration buffer (uint32, n elements incl. padding); work: 2 n words (sorted keys, identity); order: n words out.
// key_bits: the keys are known to be below 2^key_bits (32 = anything): the sort only passes over those bits.
hipError_t fsk_pixel_order_build(const uint32_t *counts, uint32_t n, uint32_t *work, uint32_t *order, void *temp, size_t temp_bytes,
                                 hipStream_t s, int key_bits)
{
    uint32_t *keys_out = work, *iota = work + n;
    hipLaunchKernelGGL(k_iota, dim3((n + 255u) / 256u), dim3(256), 0, s, iota, n);
    return hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, counts, keys_out, (const uint32_t *)iota, order, (int)n, 0,
                                                       key_bits > 0 && key_bits < 32 ? key_bits : 32, s);
}
