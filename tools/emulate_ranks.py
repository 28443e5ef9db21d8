"""Per-rank kernel time at N GPUs, measured on ONE GPU: renders each rank's interleaved row bands (fs_set_row_bands) of a bench
workload's frame in turn, every rank alone on the device as it would be on its own GPU.  max over ranks = the kernel part of
the N-GPU frame time; next to it the bytes the gather moves to rank 0 and what they cost at a STATED xGMI rate (one link per
peer into GPU 0: the slices arrive in parallel, so the gather time is one slice / one link).

Round 6: the read-back is priced too, MEASURED on this box's PCIe link -- the whole frame through one device (`--host-path gather`:
what rank 0 copies out per frame) and one rank's bands through fs_copy_bands_to_host (`--host-path direct`: every rank over its own
link, in parallel) -- and the line prints the PIPELINE's steady state per frame, not the kernel: a frame leaves every
max(kernel of the slowest rank, what the delivery path does per frame) once frames overlap two deep.
  gather:  max(kernel, gather at the stated link rate + row-order kernel at a stated HBM rate + measured whole-frame D2H)
  direct:  max(kernel, measured D2H of one rank's bands)    [assumes the host's memory system takes N links at once]

  python tools/emulate_ranks.py [--workload c3_lav2|c2_po|c5_bla|c4_hdr64|c4_2x32|c4_scaled] [--worlds 2,4,8] [--repeats 4]

One JSON line per (tile order, world).  Per rank: median and max of the repeats (the minimum is kept for continuity with round
4's lines, which reported it)."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from fractalshark_amd import (GPURenderer, LAV2_FULL, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR2X32, T_HDR32, T_HDR64, tiling)  # noqa: E402

ROW_ORDER_GBS = 3300.0  # k_gather_rows at C4's 531-MB buffer (3.3 TB/s, read + write counted)
XGMI_LINK_GBS = 64.0  # ASSUMED, not measured (no 2-GPU box this round): one xGMI link, one direction, sustained; the link peak is
                      # ~153 GB/s for both directions together, i.e. ~76 GB/s one way


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3_lav2", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--worlds", default="8")
    ap.add_argument("--world", type=int, default=0, help="(round-4 spelling of --worlds N)")
    ap.add_argument("--parity", default=None)
    ap.add_argument("--band", type=int, default=0)
    ap.add_argument("--tile-order", choices=["natural", "cold", "warm", "both", "all"], default="both",
                    help="natural: FS_VARIANT_NATURAL_TILE_ORDER (the A/B switch: nothing is ever recorded or ordered); cold: every "
                         "launch is the FIRST frame of its view (fs_forget_tile_costs before it -- a zooming viewer); warm: a rank's "
                         "launches after its first run in the order the previous ones recorded; both = natural + warm, all = the three")
    ap.add_argument("--repeats", type=int, default=4)
    a = ap.parse_args()
    worlds = [a.world] if a.world else [int(x) for x in a.worlds.split(",") if x]
    inp = bench.make_inputs(a.workload, parity=a.parity)
    W, H, AA, n_iter = inp["W"], inp["H"], inp["AA"], inp["n_iter"]
    parity = PARITY_CPU if inp["parity"] == "cpu" else PARITY_CPU_GPUSTAGE
    T = T_HDR2X32 if inp["is2x32"] else (T_HDR64 if inp["is64"] else T_HDR32)
    orbit, la = inp["orbit"], inp["la"]
    r = GPURenderer(0)
    lib = r._lib
    assert r.InitializeMemory(W, H, AA, None, 0, 0, 0, False) == 0
    if inp["is2x32"]:
        assert r.InitializePerturb(1, inp["orbit2"], 0, None, inp["la2"]) == 0
    elif inp["is_lav2"]:
        assert r.InitializePerturb(1, orbit, 0, None, la) == 0
    elif inp["is_scaled"]:
        assert lib.fs_upload_orbit_scaled(r._h, T_HDR32, 4, orbit.bad_data_ptr, orbit.bad_f32_data_ptr, orbit.count, orbit.period) == 0
    else:
        assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, orbit.data_ptr, orbit.count, orbit.count, orbit.period) == 0
        bla = inp["bla"]
        if bla is not None:
            assert lib.fs_upload_bla(r._h, T_HDR32, bla.level_ptrs, bla.level_sizes, bla.num_levels, bla.lm2) == 0
        else:
            assert lib.fs_upload_bla(r._h, T_HDR32, None, None, 0, 0) == 0
    coords, coords_arr = inp["coords"], inp["coords_arr"]

    def render():
        if inp["is_lav2"]:
            e = r.RenderPerturbLAv2(None, None, None, *coords, n_iter, T=T, Mode=LAV2_FULL, parity=parity)
        elif inp["is_scaled"]:
            e = lib.fs_render_scaled(r._h, T_HDR32, coords_arr.ctypes.data, n_iter)
        else:
            e = lib.fs_render_bla(r._h, T_HDR32, coords_arr.ctypes.data, n_iter)
        assert e == 0, GPURenderer.ConvertErrorToString(e)
        assert r.SyncComputeStream() == 0
        return r.last_kernel_ms()

    band = a.band or tiling.band_height(AA)
    rw = r.rounded_width
    # page-locked host frame for the read-back measurements
    import time

    import numpy as np
    rows_padded = (H + 7) // 8 * 8
    host = np.zeros((rows_padded, rw), np.uint32)
    assert lib.fs_host_register(host.ctypes.data, host.nbytes) == 0

    def d2h_ms(repeats=5):
        """fs_copy_bands_to_host of the CURRENT banding (no bands = the whole buffer), median wall ms incl. the stream sync."""
        assert r.SyncComputeStream() == 0
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            assert r.CopyBandsToHost(host.ctypes.data) == 0
            assert r.SyncComputeStream() == 0
            ts.append((time.perf_counter() - t0) * 1e3)
        return round(statistics.median(ts), 3)
    for order in (["natural", "warm"] if a.tile_order == "both" else ["natural", "cold", "warm"] if a.tile_order == "all"
                  else [a.tile_order]):
        assert r.set_kernel_variant(0, natural_tile_order=(order == "natural")) == 0
        for world in sorted({1, *worlds}):
            med, mx, mn, band_d2h = [], [], [], []
            ordered = None
            for rank in range(world):
                assert r.SetRowBands(rank * band, band, world * band) == 0
                band_d2h.append(d2h_ms())
                samples = []
                if order == "warm":
                    # up to the steady state: the frame that records the costs (or runs the probe) -- and, for the pixel order of
                    # the HDRFloat<double> / <CudaDblflt> frames, the second frame of the view, which sorts
                    for _ in range(3):
                        render()
                        if r.last_frame_tile_ordered():
                            break
                for i in range(max(1, a.repeats)):
                    if order == "cold":
                        r.forget_tile_costs()
                    samples.append(render())
                ordered = bool(r.last_frame_tile_ordered())
                med.append(round(statistics.median(samples), 3))
                mx.append(round(max(samples), 3))
                mn.append(round(min(samples), 3))
            slice_bytes = tiling.max_local_rows(H, world, band) * rw * 4
            gather_bytes = slice_bytes * (world - 1)
            assert r.SetRowBands(0, 0, 0) == 0
            whole_d2h = d2h_ms()
            frame_bytes = rows_padded * rw * 4
            gather_ms = slice_bytes / (XGMI_LINK_GBS * 1e9) * 1e3 if world > 1 else 0.0
            row_order_ms = 2 * frame_bytes / (ROW_ORDER_GBS * 1e9) * 1e3 if world > 1 else 0.0
            kernel = max(med)
            steady_gather = max(kernel, gather_ms + row_order_ms + whole_d2h)
            steady_direct = max(kernel, max(band_d2h))
            print(json.dumps({
                "workload": inp["key"], "parity": inp["parity"], "world": world, "tile_order": order,
                "tile_order_in_effect": ordered, "band_rows": band, "repeats": a.repeats,
                "kernel_ms_per_rank_median": med, "kernel_ms_per_rank_max": mx, "kernel_ms_per_rank_min": mn,
                "slowest_rank_median_ms": max(med), "slowest_rank_max_ms": max(mx), "mean_of_medians_ms": round(sum(med) / len(med), 3),
                "sum_of_medians_ms": round(sum(med), 3),
                "gather_bytes_to_rank0": gather_bytes, "slice_bytes": slice_bytes,
                "gather_ms_at_stated_link_rate": round(slice_bytes / (XGMI_LINK_GBS * 1e9) * 1e3, 3) if world > 1 else 0.0,
                "stated_xgmi_link_gbs": XGMI_LINK_GBS,
                "frame_bytes": frame_bytes, "d2h_whole_frame_ms_measured": whole_d2h,
                "d2h_whole_frame_gbs": round(frame_bytes / whole_d2h / 1e6, 1),
                "d2h_one_ranks_bands_ms_measured": band_d2h, "d2h_one_ranks_bands_gbs": round(slice_bytes / max(band_d2h) / 1e6, 1),
                "row_order_ms_at_stated_hbm_rate": round(row_order_ms, 3),
                "pipeline_steady_state_ms": {"gather": round(steady_gather, 3), "direct": round(steady_direct, 3)},
                "what": "steady state of the two-deep frame pipeline = max(slowest rank's kernel, per-frame work of the delivery path): "
                        "gather = slice over one xGMI link (stated rate) + row-order kernel (stated rate) + the WHOLE frame over rank "
                        "0's PCIe link (measured here); direct = one rank's bands over its own PCIe link (measured here; assumes the "
                        "host takes N links at once)"}), flush=True)
    lib.fs_host_unregister(host.ctypes.data)
    r.close()


if __name__ == "__main__":
    main()
