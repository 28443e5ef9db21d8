#!/usr/bin/env python3
"""C2: when and where the waves of one 8-row band (rendered alone) ran -- per repeat: kernel ms, the slowest waves with their
duration, steps, ns per step and placement (XCC / SE / CU / SIMD), and how many of the long waves shared a SIMD or a CU.
Needs the measurement build:  FS_TRACE_WAVES=1 python -c "from fractalshark_amd import _build; _build.build_render(force=True)"
Usage: python tools/c2_wave_trace.py [band ...]"""
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
for band in [int(b) for b in (sys.argv[1:] or ["69", "73"])]:
    assert r.SetRowBands(band * 8, 8, H) == 0
    waves = W // 8
    for rep in range(4):
        assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
        r.SyncComputeStream()
        ms = r.last_kernel_ms()
        n = 16 + 4 * waves
        buf = np.zeros(n, np.uint64)
        assert lib.fs_read_stats_raw(r._h, buf.ctypes.data, n) == 0
        t = buf[16:].reshape(-1, 4)
        dur = (t[:, 1] - t[:, 0]).astype(np.float64) / 100.0  # us (100 MHz)
        hw = (t[:, 2] & 0xFFFFFFFF).astype(np.uint32)
        xcc = ((t[:, 2] >> 32) & 0xF).astype(np.uint32)
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
        cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        simd_key = cu_key * 4 + simd
        long_w = np.nonzero(t[:, 3] > v.num_iterations // 2)[0]
        order = long_w[np.argsort(-dur[long_w])]
        per_simd = {int(k): int(c) for k, c in zip(*np.unique(simd_key[long_w], return_counts=True)) if c > 1}
        per_cu = np.unique(cu_key[long_w], return_counts=True)[1]
        rows = [{"wave": int(w), "ms": round(dur[w] / 1e3, 2), "steps": int(t[w, 3]), "ns_per_step": round(dur[w] * 1e3 / max(1, int(t[w, 3])), 2),
                 "xcc": int(xcc[w]), "se": int(se[w]), "cu": int(cu[w]), "simd": int(simd[w]),
                 "long_waves_on_its_cu": int((cu_key[long_w] == cu_key[w]).sum())} for w in list(order[:4]) + list(order[-2:])]
        print(json.dumps({"band": band, "rep": rep, "kernel_ms": round(ms, 2), "long_waves": int(len(long_w)),
                          "simds_with_two_or_more_long_waves": len(per_simd), "long_waves_per_cu_histogram": np.bincount(per_cu).tolist(),
                          "slowest_and_fastest": rows}), flush=True)
