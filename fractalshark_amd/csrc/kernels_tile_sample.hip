// kernels_tile_sample.hip -- an order for the FIRST frame of a view (round 6).
//
// A frame in the tile mapping ends on its longest waves: the pixels that cost most -- the ones PerformAT iterates longest -- come in
// regions, and when the launch reaches such a region last the chip idles behind it (C4 as HDRFloat<CudaDblflt>: 264 ms for the same
// 1.23e11 vector instructions the ordered frame issues in 193).  The recorded orders need a previous frame of the same view.  This one
// needs nothing: ONE pixel per 8 x 8 tile runs PerformAT's loop in binary64 (1/64 of the AT work; for the 2x32 frame on the record's
// values converted exactly, head + tail -- an estimate of the double-float loop's count, which is all an ORDER needs), the tiles are
// sorted by that count, longest first, and wave w of the frame's launch renders tile order[w] (FsLav2ArgsT::tile_order).  Tiles stay
// tiles -- neighbours in a wave, the same records and orbit entries -- only their launch order changes; no pixel changes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"
#include "at_math.hpp"
#include "kernels.h"
#include "kernel_common.hpp"
#include "lav2_common.hpp"

namespace {

#ifndef FS_TILE_SAMPLE_CAP
#define FS_TILE_SAMPLE_CAP 512u /* measured on C4 (one box, first frame, HDRFloat<double> / <CudaDblflt>): uncut 45.2 / 208.3 ms, 8192: 44.0 / 207.5, 2048: 43.3 / 206.4, 1024: 43.1 / 208.1, 512: 43.0 / 205.2 */
#endif
constexpr uint32_t kSampleCap = FS_TILE_SAMPLE_CAP;

__global__ void __launch_bounds__(256) k_at_tile_sample64(FsTileSampleArgs A)
{
    using F = double;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= A.n_slots)
        return;
    const uint32_t n_tiles = A.tiles_x * A.tiles_y;
    if (t >= n_tiles) {
        A.cost[t] = 0u; // (slots of the launch beyond the last tile: they sort last and name no tile, see k_tile_order_finish)
        return;
    }
    const uint32_t ty = t / A.tiles_x, tx = t - ty * A.tiles_x;
    uint32_t X = tx * 8u + 3u, L = ty * 8u + 3u;
    X = X < A.frame.width ? X : A.frame.width - 1u;
    L = L < A.frame.local_rows ? L : A.frame.local_rows - 1u;
    const uint32_t Y = global_row(A.frame, L);
    uint32_t own = 0;
    if (Y < A.frame.height) {
        fs::hreal<F> deltaReal, deltaImaginary;
        pixel_delta<F>(A.coords, X, Y, deltaReal, deltaImaginary);
        const fs::hcplx<F> dc = fs::hc_from_hr(deltaReal, deltaImaginary);
        if (fs::hr_cmp_pos(fs::hc_cheb(dc), A.ThresholdC) <= 0) {
            // (an ORDER needs to know which tiles are long, not how long the longest are: the loop is cut at kSampleCap iterations -- the
            // pass is as long as its longest lane's chain, 2.6 ms uncut on C4's view against 45 for the frame)
            const uint32_t full = A.n_iterations / A.StepLength;
            const uint32_t ATMaxIt = full < kSampleCap ? full : kSampleCap;
            fs::hcplx<F> c = fs::hc_add(fs::hc_mul(dc, A.CCoeff), A.RefC);
            fs::hc_reduce(c);
            fs::hcplx<F> z;
            uint32_t i, i_exec = 0, i_own = 0;
            at_perform<F, uint32_t>(c, A.SqrEscapeRadius, ATMaxIt, z, i, &i_exec, &i_own);
            own = i_own + 1u; // (+ 1: a tile whose pixel takes the AT step at all sorts before the ones that do not)
        }
    }
    A.cost[t] = own;
}

// order[] as the radix sort left it names SLOTS (0 .. n_slots); slots beyond the last tile become "no tile"
__global__ void k_tile_order_finish(uint32_t *order, uint32_t n_slots, uint32_t n_tiles)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots && order[i] >= n_tiles)
        order[i] = 0xFFFFFFFFu;
}

} // namespace

void fsk_at_tile_sample64(const FsTileSampleArgs &A, hipStream_t s)
{
    hipLaunchKernelGGL(k_at_tile_sample64, dim3((A.n_slots + 255u) / 256u), dim3(256), 0, s, A);
}

void fsk_tile_order_finish(uint32_t *order, uint32_t n_slots, uint32_t n_tiles, hipStream_t s)
{
    hipLaunchKernelGGL(k_tile_order_finish, dim3((n_slots + 255u) / 256u), dim3(256), 0, s, order, n_slots, n_tiles);
}
