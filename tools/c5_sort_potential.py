"""If the pixels of a BLA frame were handed to the lanes SORTED by the iteration count the previous frame of the view gave them
(instead of 8 x 8 tiles), how many wave passes would the step loop need?  Per-pixel step counts from the probe build of the
hand-written kernel (FS_BLA_FAST_PROBE=1 build; FSMI355_BLA_STEPS_OUT=1), per-pixel counts from the same kernel.
Usage: python tools/c5_sort_potential.py [width height]"""
import json
import os
import sys

import numpy as np

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


def frame():
    assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
    buf = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, buf) == 0
    assert r.SyncComputeStream() == 0
    return buf[:H // 8 * 8, :W // 32 * 32].astype(np.int64)


counts = frame()
os.environ["FSMI355_BLA_STEPS_OUT"] = "1"
steps = frame()
t = steps.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
now = int(t.max(axis=1).sum())
order = np.argsort(counts.ravel(), kind="stable")
s_sorted = steps.ravel()[order]
n = s_sorted.size // 64 * 64
by_count = int(s_sorted[:n].reshape(-1, 64).max(axis=1).sum())
# the same with the log-scale bucket key a device-side grouping would use (float bits >> 12)
key = (counts.ravel().astype(np.float32).view(np.uint32) >> 12)
order2 = np.argsort(key, kind="stable")
s2 = steps.ravel()[order2]
by_bucket = int(s2[:n].reshape(-1, 64).max(axis=1).sum())
ideal = int(np.ceil(steps.sum() / 64.0))
print(json.dumps({"frame": "%dx%d" % (W, H), "wave_passes_tiles_8x8": now, "sorted_by_count": by_count, "sorted_by_bucketed_count": by_bucket,
                  "full_occupancy": ideal, "sorted_over_now": round(by_count / now, 4), "bucketed_over_now": round(by_bucket / now, 4),
                  "full_over_now": round(ideal / now, 4)}))
