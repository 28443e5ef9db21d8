"""Which forms the actions of the BLA kernel take on C5's view (step-counting build of k_perturb_scalar<float, kBla, kNat>):
lane-passes through the quiet step / the step that forms z / the literal step, the quiet jump / the jump that forms z.
Usage: python tools/c5_action_probe.py [width height]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, T_HDR32, inputs  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
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
assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
assert r.SyncComputeStream() == 0
raw = (C.c_uint64 * 32)()
assert lib.fs_read_stats_raw(r._h, raw, 32) == 0
st = r.read_step_count()
names = ["quiet_step", "step_with_z", "literal_step", "quiet_jump", "jump_with_z"]
d = dict(zip(names, list(raw)[8:13]))
d["literal_jump"] = st["la_steps"] - d["quiet_jump"] - d["jump_with_z"]
print(json.dumps({"frame": "%dx%d" % (W, H), "perturb_steps": st["perturb_steps"], "bla_jumps": st["la_steps"], "lane_passes": d,
                  "share_of_steps": {k: round(d[k] / max(1, st["perturb_steps"]), 5) for k in names[:3]},
                  "share_of_jumps": {k: round(d[k] / max(1, st["la_steps"]), 5) for k in ("quiet_jump", "jump_with_z", "literal_jump")}}))
