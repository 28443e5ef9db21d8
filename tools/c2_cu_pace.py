#!/usr/bin/env python3
"""C2: pace (ns per step) of every long wave of some 8-row bands rendered alone, over several repeats, by placement
(XCC / SE / CU): is a lone wave's pace a property of where it runs?  Measurement build as for tools/c2_wave_trace.py."""
import collections
import json
import os
import sys

import numpy as np

os.environ.setdefault("FSMI355_TRACE_WAVES", "40000")
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
by_place = collections.defaultdict(list)
by_wave = collections.defaultdict(list)
by_mhz = []
for band in [int(b) for b in (sys.argv[1:] or ["72", "74", "76"])]:
    assert r.SetRowBands(band * 8, 8, H) == 0
    waves = W // 8
    for rep in range(6):
        assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
        r.SyncComputeStream()
        n = 16 + 4 * waves
        buf = np.zeros(n, np.uint64)
        assert lib.fs_read_stats_raw(r._h, buf.ctypes.data, n) == 0
        t = buf[16:].reshape(-1, 4)
        steps_all = t[:, 3] & np.uint64(0xFFFFFFFF)
        for w in np.nonzero(steps_all >= v.num_iterations - 1)[0]:
            ns = (int(t[w, 1]) - int(t[w, 0])) * 10.0 / int(steps_all[w])
            mhz = (int(t[w, 3]) >> 32) * 1024.0 / ((int(t[w, 1]) - int(t[w, 0])) / 100.0)
            by_mhz.append((round(ns, 2), round(mhz)))
            hw = int(t[w, 2]) & 0xFFFFFFFF
            xcc = (int(t[w, 2]) >> 32) & 0xF
            by_place[(xcc, (hw >> 13) & 7, (hw >> 8) & 0xF)].append(ns)
            by_wave[(band, int(w))].append(round(ns, 1))
rows = sorted(((round(float(np.mean(x)), 2), k, len(x), round(float(np.min(x)), 1), round(float(np.max(x)), 1)) for k, x in by_place.items()))
allns = [x for v_ in by_place.values() for x in v_]
print(json.dumps({"lib": os.path.basename(os.environ.get("FSMI355_LIB", "libfsmi355.so")), "all_long_waves": len(allns),
                  "mean_ns_per_step": round(float(np.mean(allns)), 2), "median": round(float(np.median(allns)), 2),
                  "p90": round(float(np.percentile(allns, 90)), 2), "max": round(float(np.max(allns)), 2), "min": round(float(np.min(allns)), 2)}))
if os.environ.get("C2_PACE_SUMMARY") == "1":
    sys.exit(0)
print(json.dumps({"by_xcc_se_cu_mean_ns_n_min_max": [[list(k), m, n, lo, hi] for (m, k, n, lo, hi) in rows]}))
xs = collections.defaultdict(list)
for (xcc, se, cu), x in by_place.items():
    xs[xcc] += x
print(json.dumps({"by_xcc_mean_ns": {str(k): [round(float(np.mean(x)), 2), len(x)] for k, x in sorted(xs.items())}}))
print(json.dumps({"ns_per_step_and_shader_mhz_sorted": sorted(by_mhz)}))
print(json.dumps({"by_wave": {"%d/%d" % k: x for k, x in sorted(by_wave.items())}}))
