#!/usr/bin/env python3
"""How the lanes of k_lav2_hdr64's waves agree, counted by its counting instantiation (statistics words 8..15): wave steps of the
perturbation and LA loops, and how many of them have lanes on DIFFERENT arms of an HDRFloatComplex add or a lane that rebases --
what decides whether a per-arm fast path or a branch-free form is the right shape for the loop.  C4's frame (View 14, 15360x8640)
in the tile mapping and in the count order.  Tallies are per wave, taken from each wave's first lane (a sample of the wave's steps).
  FSMI355_STATS_KEEP_ORDER=1 python tools/c4_arm_probe.py [--width W --height H]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fractalshark_amd import GPURenderer, LAV2_FULL, PARITY_CPU_GPUSTAGE, T_HDR64  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
a = ap.parse_args()
inp = bench.make_inputs("c4_hdr64", width=a.width, height=a.height)
W, H, AA = inp["W"], inp["H"], inp["AA"]
r = GPURenderer(0)
assert r.InitializeMemory(W, H, AA, None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, inp["orbit"], 0, None, inp["la"]) == 0


def frame(stats):
    r.enable_step_count(stats)
    assert r.RenderPerturbLAv2(None, None, None, *inp["coords"], inp["n_iter"], T=T_HDR64, Mode=LAV2_FULL, parity=PARITY_CPU_GPUSTAGE) == 0
    assert r.SyncComputeStream() == 0
    raw = (C.c_uint64 * 40)()
    if stats:
        assert r._lib.fs_read_stats_raw(r._h, raw, 40) == 0
    return list(raw), bool(r.last_frame_tile_ordered()), r.last_kernel_ms()


names = ["pt_wave_steps", "pt_mixed_2Z+dz", "pt_mixed_dz*t+dc", "pt_mixed_Z+dz", "pt_steps_with_a_rebasing_lane",
         "la_wave_steps", "la_steps_with_a_mixed_add", "la_steps_with_a_rebasing_lane",
         "la_steps_all_lanes_one_record", "la_distinct_records_summed", "pt_steps_all_lanes_one_entry", "pt_distinct_entries_summed"]
for label, warm in (("tile mapping (first frame)", 0), ("count order (third frame on)", 3)):
    r.forget_tile_costs()
    for _ in range(warm):
        frame(False)
    raw, ordered, ms = frame(True)
    d = {"frame": label, "ordered": ordered, "W": W, "H": H, "perturb_lane_steps": raw[2], "la_lane_steps": raw[1],
         "lane_slots_pt": raw[4]}
    d.update({n: raw[8 + k] for k, n in enumerate(names)})
    if any(raw[20:24]):
        d["la_statement_exits_by_status"] = raw[20:24]
    if any(raw[24:28]):
        d["pt_statement_exits_by_status"] = raw[24:28]
    if any(raw[20:28]):  # the probe build (FS_H64_LA_ASM_DEBUG=1)
        d["la_statement_half_steps_and_general_sums_A_S_C"] = raw[28:32]
        d["pt_statement_half_steps_generalA_rebasing_generalC"] = raw[32:36]
    else:
        d["all_lanes_on_arm"] = {"la 2Ref+dz [a alone, a on top]": raw[28:30], "la newDz ZCoeff + dc CCoeff [a alone, a on top, b on top, b alone]": raw[30:34],
                                 "la nextRef+dz' [a alone, a on top]": raw[34:36], "pt 2Z+dz [a alone, a on top]": raw[36:38],
                                 "pt Z'+dz' [a alone, a on top]": raw[38:40]}
    if raw[13]:
        d["la_distinct_records_per_wave_step"] = round(raw[17] / raw[13], 2)
    if raw[8]:
        d["pt_distinct_entries_per_wave_step"] = round(raw[19] / raw[8], 2)
    print(json.dumps(d), flush=True)
r.close()
