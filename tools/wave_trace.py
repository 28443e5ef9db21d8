#!/usr/bin/env python3
"""When and where every wave of the tuned LAv2 kernel ran: per-SIMD occupancy over time for one rank of an N-GPU split
(measured on one GPU).  Needs a measurement build:
    FS_TRACE_WAVES=1 python -c "from fractalshark_amd import _build; _build.build_render(force=True)"
(and a normal forced rebuild afterwards).  Usage: FSMI355_TRACE_WAVES=140000 python tools/wave_trace.py [--world 8]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, PARITY_CPU, T_HDR32, inputs, tiling  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=0)
a = ap.parse_args()
os.environ.setdefault("FSMI355_TRACE_WAVES", "140000")
v = inputs.View.builtin(5, 3840, 2160, antialiasing=1)
o = inputs.Orbit(v)
la = inputs.LATable(o, host_threads=16)
co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(o)]
r = GPURenderer(0)
assert r.InitializeMemory(3840, 2160, 1, None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, o, 0, None, la) == 0
band = tiling.band_height(1)
if a.world > 1:
    assert r.SetRowBands(a.rank * band, band, a.world * band) == 0
rows = r.local_rows
waves = (3840 // 8) * (rows // 8)
r.enable_step_count(True)
for _ in range(2):
    assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
    assert r.SyncComputeStream() == 0
ms = r.last_kernel_ms()
n = 16 + 4 * waves
buf = np.zeros(n, np.uint64)
assert r._lib.fs_read_stats_raw(r._h, buf.ctypes.data, n) == 0
t = buf[16:].reshape(-1, 4)
# waves are numbered (blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave: 480 tiles per 8-row band
band_of = (np.arange(len(t)) // 480)[t[:, 1] > 0]
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
start = (t[:, 0] - t0).astype(np.float64) / 100.0  # us (100 MHz)
end = (t[:, 1] - t0).astype(np.float64) / 100.0
hw = (t[:, 2] & 0xFFFFFFFF).astype(np.uint32)
xcc = (t[:, 2] >> 32).astype(np.uint32) & 0xF
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
dur = end - start
steps = t[:, 3].astype(np.float64)
total = end.max()
# occupancy over time: number of resident waves per 100 us bucket, averaged over the SIMDs seen
nb = int(total // 100) + 1
occ = np.zeros(nb)
for s_, e_ in zip(start, end):
    b0, b1 = int(s_ // 100), int(e_ // 100)
    occ[b0:b1 + 1] += 1
nsimd = len(np.unique(key))
per = np.bincount(np.unique(key, return_inverse=True)[1])
band_cost = np.bincount(band_of, weights=dur)
band_steps = np.bincount(band_of, weights=steps)
print(json.dumps({"band_wave_ms_sum": [round(float(x) / 1e3, 1) for x in band_cost],
                  "band_wave_steps_sum_k": [int(x / 1e3) for x in band_steps]}))
print(json.dumps({"world": a.world, "rank": a.rank, "kernel_ms_with_trace_build": round(ms, 3), "waves": int(len(t)),
                  "simds_seen": int(nsimd), "waves_per_simd_min_mean_max": [int(per.min()), round(float(per.mean()), 2), int(per.max())],
                  "span_us": round(float(total), 1),
                  "wave_duration_us_pct_5_50_95_max": [round(float(x), 1) for x in np.percentile(dur, [5, 50, 95, 100])],
                  "wave_steps_pct_5_50_95_max": [int(x) for x in np.percentile(steps, [5, 50, 95, 100])],
                  "ns_per_step_pct_5_50_95": [round(float(x), 1) for x in np.percentile(dur * 1e3 / np.maximum(steps, 1), [5, 50, 95])],
                  "last_start_us": round(float(start.max()), 1),
                  "resident_waves_per_simd_by_100us": [round(float(x) / nsimd, 2) for x in occ]}))
