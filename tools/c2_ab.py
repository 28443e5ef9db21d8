#!/usr/bin/env python3
"""C2 (View 5, perturbation only, 1920x1080), frames of one view back to back: kernel ms of every frame (HIP events on the
compute stream), for the library named by FSMI355_LIB (default: the in-tree build).
Usage: [FSMI355_LIB=path] python tools/c2_ab.py [--frames 8]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, T_HDR32, inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
a = ap.parse_args()
W, H = 1920, 1080
v = inputs.View.builtin(5, W, H, antialiasing=1)
o = inputs.Orbit(v)
co = v.coords_perturb(o)
r = GPURenderer(0)
assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
lib = r._lib
assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, o.data_ptr, o.count, o.count, o.period) == 0
ms = []
for _ in range(a.frames + 1):
    assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
    r.SyncComputeStream()
    ms.append(round(r.last_kernel_ms(), 2))
out = r.new_iter_buffer()
assert r.RenderCurrent(v.num_iterations, out) == 0
r.SyncComputeStream()
import zlib
crc = zlib.crc32(np.ascontiguousarray(out[:H, :W]).astype("<u4").tobytes()) & 0xFFFFFFFF
print(json.dumps({"lib": os.path.basename(os.environ.get("FSMI355_LIB", "libfsmi355.so")),                   "cold_ms": ms[0], "warm_ms": ms[1:], "warm_median": float(np.median(ms[1:])), "warm_min": min(ms[1:]),
                  "frame_crc32": "%08x" % crc}), flush=True)
