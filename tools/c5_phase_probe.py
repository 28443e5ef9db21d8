"""Where a wave's time goes in the BLA kernel (k_perturb_scalar<float, kBla>) on C5 (View 19, 7680x4320 by default).

Needs a library built with -DFS_PROFILE_CYCLES:
  FS_PROFILE_CYCLES=1 python -c "from fractalshark_amd import _build; _build.build_render(force=True)"
(and a normal forced rebuild afterwards).  The instrumented (step-counting) kernel then adds, per wave, the shader-clock
cycles it spent in each phase of its loop and how often it passed through it (the clock is read on the scalar unit: one
reading per pass of the wave, whatever its lane mask is) to words 16..27 of the statistics buffer:
  lookup   BLAS::LookupBackwards probes (one pass = one round of the per-lane `while (a table entry applies)` loop)
  jump     BLA::getValue + z = Z + dz, the norms, escape / rebase tests (straight-line form)
  step     the single perturbation step (straight-line form, quiet or with z)
  literal  fall-backs to the reference-order code (jump or step)
Cycles are wall cycles of the wave with 8 waves sharing a SIMD, so they are shares, not issue counts; the vector
instruction counts per pass are read off the ISA (tools/c5_isa_blocks.py).
Usage: python tools/c5_phase_probe.py [width height]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, T_HDR32, inputs  # noqa: E402
import ctypes as C  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
v = inputs.View.builtin(19, W, H, antialiasing=1)
o = inputs.Orbit(v)
bla = inputs.BLATable(o)
co = v.coords_perturb(o)
r = GPURenderer(0)
assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
lib = r._lib
assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, o.data_ptr, o.count, o.count, o.period) == 0
assert lib.fs_upload_bla(r._h, T_HDR32, bla.level_ptrs, bla.level_sizes, bla.num_levels, bla.lm2) == 0
r.enable_step_count(True)
for _ in range(2):
    assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
    assert r.SyncComputeStream() == 0
raw = (C.c_uint64 * 32)()
assert lib.fs_read_stats_raw(r._h, raw, 32) == 0
st = r.read_step_count()
ph = list(raw)[16:29]
names = ["lookup", "jump", "step", "literal"]
cyc = dict(zip(names, ph[0:4]))
n = dict(zip(names, ph[4:8]))
outer, lanes_jump, lanes_step, waves = ph[8], ph[9], ph[10], ph[11]
tot = sum(cyc.values())
out = {"frame": "%dx%d" % (W, H), "kernel_ms_instrumented": round(r.last_kernel_ms(), 3), "waves": waves,
       "perturb_steps": st["perturb_steps"], "bla_jumps": st["la_steps"], "lane_slots": st["lane_slots"],
       "outer_trips_per_wave": round(outer / max(1, waves), 1),
       "phases": {k: {"cycle_share": round(cyc[k] / max(1, tot), 4), "passes_per_wave": round(n[k] / max(1, waves), 1),
                      "cycles_per_pass": round(cyc[k] / max(1, n[k]), 1)} for k in names},
       "active_lanes_per_jump_pass": round(lanes_jump / max(1, n["jump"]), 2),
       "active_lanes_per_step_pass": round(lanes_step / max(1, n["step"]), 2),
       "lookup_passes_per_outer_trip": round(n["lookup"] / max(1, outer), 3),
       "jump_passes_per_outer_trip": round(n["jump"] / max(1, outer), 3),
       "step_passes_taken_as_one_scaled_step": round(ph[12] / max(1, n["step"]), 4)}
print(json.dumps(out))
