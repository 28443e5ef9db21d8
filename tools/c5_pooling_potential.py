"""How many wave passes the BLA kernel would save if the running pixels of a workgroup's four waves (4 adjacent 8 x 8 tiles) were
re-packed into as few waves as possible: per-pixel STEP counts from the probe build of the hand-written kernel
(FS_BLA_FAST_PROBE=1 build, FSMI355_BLA_STEPS_OUT=1), then passes now = sum over waves of the longest lane, passes pooled = sum over
trips of ceil(running pixels of the workgroup / 64).  Usage: python tools/c5_pooling_potential.py [width height]"""
import json
import os
import sys

import numpy as np

os.environ["FSMI355_BLA_STEPS_OUT"] = "1"
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
assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
buf = r.new_iter_buffer()
assert r.RenderCurrent(v.num_iterations, buf) == 0
assert r.SyncComputeStream() == 0
st = buf[:H // 8 * 8, :W // 32 * 32].astype(np.int64)
# waves: 8 x 8 tiles; workgroups: 4 tiles side by side (32 x 8 pixels)
t = st.reshape(H // 8, 8, W // 32, 4, 8).transpose(0, 2, 3, 1, 4).reshape(H // 8, W // 32, 4, 64)
wave_max = t.max(axis=3)
now = int(wave_max.sum())
mean_steps = float(st.mean())
# pooled: for every workgroup, running(t) = number of pixels with steps > t; passes = sum_t ceil(running / 64)
wg = np.sort(t.reshape(-1, 256), axis=1)[:, ::-1]  # descending
# pixels sorted descending: the k-th longest pixel (0-based) is running during steps_k trips; ceil(running/64) increments at k = 0, 64, 128, 192
pooled = int(wg[:, 0].sum() + wg[:, 64].sum() + wg[:, 128].sum() + wg[:, 192].sum())
ideal = int(np.ceil(st.sum() / 64.0))
print(json.dumps({"frame": "%dx%d" % (W, H), "mean_steps_per_pixel": round(mean_steps, 1),
                  "wave_passes_now": now, "wave_passes_pooled_per_workgroup": pooled, "wave_passes_at_full_occupancy": ideal,
                  "pooled_over_now": round(pooled / now, 4), "full_over_now": round(ideal / now, 4),
                  "longest_lane_per_wave_mean": round(float(wave_max.mean()), 1)}))
