"""Per-rank kernel time at N GPUs, measured on one GPU: renders each rank's interleaved row bands (fs_set_row_bands)
of the C3 frame in turn.  max over ranks = the kernel part of the N-GPU frame time.
Usage: python tools/emulate_ranks.py [--world 8] [--parity cpu|cpu_gpustage]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fractalshark_amd import (GPURenderer, LAV2_FULL, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR32, inputs, tiling)  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--parity", default="cpu")
    ap.add_argument("--band", type=int, default=0)
    ap.add_argument("--tile-order", choices=["natural", "warm", "both"], default="both",
                    help="natural: FS_VARIANT_NATURAL_TILE_ORDER (every frame cold); warm: a rank's launches after its first "
                         "run longest tiles first from the costs the previous one recorded")
    ap.add_argument("--repeats", type=int, default=4,
                    help="launches per rank; the minimum is reported (kernel time on an otherwise idle GPU: run-to-run "
                         "differences of +-3 %% between identical launches are clock / placement noise, and on a real "
                         "8-GPU node every rank runs alone on its own GPU)")
    a = ap.parse_args()
    v = inputs.View.builtin(5, 3840, 2160, antialiasing=1)
    o = inputs.Orbit(v)
    la = inputs.LATable(o, host_threads=16)
    co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(o)]
    parity = PARITY_CPU if a.parity == "cpu" else PARITY_CPU_GPUSTAGE
    r = GPURenderer(0)
    assert r.InitializeMemory(3840, 2160, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, o, 0, None, la) == 0
    band = a.band or tiling.band_height(1)
    for order in (["natural", "warm"] if a.tile_order == "both" else [a.tile_order]):
        assert r.set_kernel_variant(0, natural_tile_order=(order == "natural")) == 0
        for world in sorted({1, a.world}):
            times = []
            spreads = []
            for rank in range(world):
                assert r.SetRowBands(rank * band, band, world * band) == 0
                best = 1e9
                samples = []
                for i in range(max(1, a.repeats) + (1 if order == "warm" else 0)):
                    assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL,
                                               parity=parity) == 0
                    assert r.SyncComputeStream() == 0
                    if order == "warm" and i == 0:
                        continue  # the frame that records the costs
                    assert r.last_frame_tile_ordered() == (order == "warm")
                    samples.append(r.last_kernel_ms())
                    best = min(best, samples[-1])
                times.append(round(best, 3))
                spreads.append(round(max(samples) - min(samples), 3))
            print(json.dumps({"world": world, "tile_order": order, "band_rows": band, "parity": a.parity,
                              "kernel_ms_per_rank": times, "max_ms": max(times), "mean_ms": round(sum(times) / len(times), 3),
                              "sum_ms": round(sum(times), 3), "repeats": a.repeats,
                              "max_spread_between_repeats_ms": max(spreads)}), flush=True)


if __name__ == "__main__":
    main()
