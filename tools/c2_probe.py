"""C2 (View 5, perturbation only, 1920x1080) is bounded by the waves that hold interior pixels: each runs its 4.7 M
steps alone on its SIMD.  This probe renders the frame, picks the 8-row band with the most pixels at the cap, renders
that band alone (every wave of it is alone on its SIMD) and reports what such a wave's steps are made of:
ns per step of the longest chain, steps per scaled run, single (careful) steps and literal wave-trips per run."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, T_HDR32, inputs  # noqa: E402

W, H = 1920, 1080
v = inputs.View.builtin(5, W, H, antialiasing=1)
o = inputs.Orbit(v)
co = v.coords_perturb(o)
r = GPURenderer(0)
assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
lib = r._lib
assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, o.data_ptr, o.count, o.count, o.period) == 0
assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
r.SyncComputeStream()
frame_ms = r.last_kernel_ms()
out = r.new_iter_buffer()
assert r.RenderCurrent(v.num_iterations, out) == 0
r.SyncComputeStream()
cap = out[:H, :W] >= v.num_iterations
per_band = cap.reshape(H // 8, 8, W).sum(axis=(1, 2))
band = int(os.environ.get("C2_BAND", per_band.argmax()))
print(json.dumps({"frame_ms": round(frame_ms, 2), "pixels_at_cap": int(cap.sum()), "band": band,
                  "band_pixels_at_cap": int(per_band[band]), "bands_with_cap_pixels": int((per_band > 0).sum())}))
assert r.SetRowBands(band * 8, 8, H) == 0
for stats in (False, True):
    r.enable_step_count(stats)
    for _ in range(2):
        assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
        r.SyncComputeStream()
    ms = r.last_kernel_ms()
    d = {"band_kernel_ms": round(ms, 2), "instrumented": stats, "ns_per_step_of_longest_chain": round(ms * 1e6 / v.num_iterations, 2)}
    if stats:
        st = r.read_step_count()
        raw = (C.c_uint64 * 32)()
        assert lib.fs_read_stats_raw(r._h, raw, 32) == 0
        d.update({"lane_steps_in_the_untested_loop": raw[8], "lane_steps_in_tested_blocks": 4 * raw[9],
                  "runs_ended_by (per wave, as lane 0 saw them)": {"their_length": raw[11], "H": raw[12], "a_tested_block_failing": raw[13],
                                                                   "floor_status_3": raw[14]}, "block_test_roll_backs_status_4": raw[15]})
        runs = max(1, st["scaled_runs"])
        d.update({"steps": st["perturb_steps"], "scaled_steps": st["la_steps"], "scaled_runs": st["scaled_runs"],
                  "single_steps": st["at_iterations"], "literal_wave_trips": st["careful_steps"],
                  "steps_per_scaled_run": round(st["la_steps"] / runs, 1),
                  "single_steps_per_run": round(st["at_iterations"] / runs, 3),
                  "other_steps_per_run": round((st["perturb_steps"] - st["la_steps"] - st["at_iterations"]) / runs, 3)})
    print(json.dumps(d))
