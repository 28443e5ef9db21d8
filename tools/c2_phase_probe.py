#!/usr/bin/env python3
"""C2: what the time of a band's waves is made of (each wave alone on its SIMD): shader-clock cycles in the scaled-run block,
in the hand-scheduled statement inside it, in the exponent-tracking block, and the rest (single steps, bookkeeping).
Needs FS_PROFILE_CYCLES=1 python -c "from fractalshark_amd import _build; _build.build_render(force=True)"."""
import ctypes as C
import json
import os
import sys

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
r.enable_step_count(True)
for band in [int(b) for b in (sys.argv[1:] or ["72"])]:
    assert r.SetRowBands(band * 8, 8, H) == 0
    for _ in range(2):
        assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
        r.SyncComputeStream()
    raw = (C.c_uint64 * 32)()
    assert lib.fs_read_stats_raw(r._h, raw, 32) == 0
    st = r.read_step_count()
    total, run, asm, quiet, n_run, n_asm = [int(raw[16 + i]) for i in range(6)]
    rest = total - run - quiet
    steps = st["perturb_steps"]
    print(json.dumps({"band": band, "kernel_ms": round(r.last_kernel_ms(), 2),
                      "share_run_block": round(run / total, 3), "share_statement": round(asm / total, 3),
                      "share_run_entry_exit_and_tested_blocks": round((run - asm) / total, 3),
                      "share_exponent_tracking_block": round(quiet / total, 3), "share_single_steps_and_rest": round(rest / total, 3),
                      "run_blocks_entered": n_run, "statement_invocations": n_asm,
                      "cycles_per_run_block_outside_statement": round((run - asm) / max(1, n_run)),
                      "cycles_per_statement_invocation": round(asm / max(1, n_asm)),
                      "cycles_per_pass_outside_run_block": round((total - run) / max(1, n_run)),
                      "scaled_runs": st["scaled_runs"], "single_steps": st["at_iterations"], "lane_steps": steps}), flush=True)
