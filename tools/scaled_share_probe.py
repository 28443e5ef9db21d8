"""Share of the perturbation steps of the tuned LAv2 kernel that run inside scaled runs, by frame size (View 5).
Usage: python tools/scaled_share_probe.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, PARITY_CPU, T_HDR32, inputs  # noqa: E402

r = GPURenderer(0)
for w, h in ((480, 270), (960, 540), (1920, 1080), (3840, 2160)):
    v = inputs.View.builtin(5, w, h, antialiasing=1)
    o = inputs.Orbit(v)
    la = inputs.LATable(o)
    co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(o)]
    assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, o, 0, None, la) == 0
    for lds in (False, True):
        r.set_kernel_variant(0, lds_orbit=lds)
        r.enable_step_count(True)
        assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
        assert r.SyncComputeStream() == 0
        st = r.read_step_count()
        raw = (C.c_uint64 * 32)()
        assert r._lib.fs_read_stats_raw(r._h, raw, 32) == 0
        blk_free, blk_tested = raw[8], raw[9]
        ms_stats = r.last_kernel_ms()
        r.enable_step_count(False)
        assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
        assert r.SyncComputeStream() == 0
        print(json.dumps({"size": "%dx%d" % (w, h), "lds": lds, "kernel_ms": round(r.last_kernel_ms(), 3), "kernel_ms_counting_build": round(ms_stats, 3),
                          "steps_per_pixel": round(st["perturb_steps"] / (w * h), 1),
                          "scaled_share": round(st["scaled_steps"] / max(1, st["perturb_steps"]), 4),
                          "careful_share": round(st["careful_steps"] / max(1, st["perturb_steps"]), 4),
                          "steps_per_run": round(st["scaled_steps"] / max(1, st["scaled_runs"]), 1),
                          "blocks_without_bound_tests": round(blk_free / max(1, blk_free + blk_tested), 4),
                          "wave_blocks": blk_free + blk_tested,
                          "careful_passes_per_wave": round(raw[10] / (w * h / 64), 1),
                          "run_entries_tried_per_wave": round(raw[12] / (w * h / 64), 1),
                          "runs_started_per_wave": round(raw[13] / (w * h / 64), 1),
                          "runs_shorter_than_8_per_wave": round(raw[14] / (w * h / 64), 1),
                          "entry_failures_per_wave": round(raw[16] / (w * h / 64), 1),
                          "careful_passes_per_wave_with": {"a_rebase_in_some_lane": round(raw[17] / (w * h / 64), 1),
                                                           "a_rebase_in_every_running_lane": round(raw[18] / (w * h / 64), 1),
                                                           "an_escape": round(raw[19] / (w * h / 64), 1)},
                          "careful_passes_per_wave_arriving_at": {"an_entry_no_scaled_step_may_arrive_at": round(raw[20] / (w * h / 64), 1),
                                                                   "such_an_entry_and_nothing_happens": round(raw[21] / (w * h / 64), 1),
                                                                   "another_entry_and_nothing_happens": round(raw[22] / (w * h / 64), 1),
                                                                   "as_a_back_off_wait": round(raw[23] / (w * h / 64), 1)},
                          "per_lane_entry_path": {"wave_steps_per_wave": round(raw[28] / (w * h / 64), 1), "runs_per_wave": round(raw[29] / (w * h / 64), 1)},
                          "generic_step_share_of_passes": round(raw[11] / max(1, raw[10]), 4),
                          "lane_utilisation": round(st["perturb_steps"] / max(1, st["lane_slots"]), 4)}), flush=True)
r.set_kernel_variant(0)
