"""Where a wave's time goes in k_lav2_hdr32_fast.  Needs a build with -DFS_PROFILE_CYCLES
(`FS_PROFILE_CYCLES=1 python -c "from fractalshark_amd import _build; _build.build_render(force=True)"`, and a normal
forced rebuild afterwards): the instrumented kernel then
reports shader-clock cycles per wave in the stats slots: [0] perturbation loop, [1] scaled-run block (entry + bodies +
exit), [3] scaled bodies only).  Full frame (8 waves per SIMD) and one 8-row band (every wave alone on its SIMD)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, PARITY_CPU, T_HDR32, inputs  # noqa: E402

v = inputs.View.builtin(5, 3840, 2160, antialiasing=1)
o = inputs.Orbit(v)
la = inputs.LATable(o, host_threads=16)
co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(o)]
r = GPURenderer(0)
assert r.InitializeMemory(3840, 2160, 1, None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, o, 0, None, la) == 0
r.enable_step_count(True)
for name, bands, waves in (("full", None, 3840 * 2160 // 64), ("band135", (135 * 8, 8, 270 * 8), 480),
                           ("band7", (7 * 8, 8, 270 * 8), 480)):
    if bands:
        assert r.SetRowBands(*bands) == 0
    for _ in range(2):
        assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
        assert r.SyncComputeStream() == 0
    s = r.read_step_count()
    import ctypes as C
    raw = (C.c_uint64 * 32)()
    assert r._lib.fs_read_stats_raw(r._h, raw, 32) == 0
    loop, run, body = s["at_iterations"], s["la_steps"], s["pixels"]
    print(json.dumps({"case": name, "kernel_ms": round(r.last_kernel_ms(), 3), "waves": waves,
                      "cyc_loop_per_wave": loop // waves, "cyc_run_block_per_wave": run // waves,
                      "cyc_bodies_per_wave": body // waves, "cyc_outside_run_block": (loop - run) // waves,
                      "cyc_entry_exit": (run - body) // waves,
                      "cyc_hand_written_statement": raw[24] // waves, "cyc_tested_blocks": raw[25] // waves, "cyc_hot_runs": raw[26] // waves,
                      "tested_blocks_per_wave": raw[9] / waves, "untested_blocks_per_wave": raw[8] / waves,
                      "run_entries_per_wave": raw[12] / waves, "runs_started_per_wave": raw[13] / waves,
                      "careful_passes_per_wave": raw[10] / waves,
                      "shader_clock_ghz_in_the_loop": round(loop / max(1, raw[27]) * 0.1, 3),
                      "steps_per_wave_lane": s["perturb_steps"] // (waves * 64), "scaled_lane_steps": s["scaled_steps"],
                      "runs_per_wave": s["scaled_runs"] / waves, "careful_per_wave_lane": s["careful_steps"] / waves / 64}), flush=True)
