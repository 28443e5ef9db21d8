#!/usr/bin/env python3
"""C3 (View 5, 3840x2160, HDRFloat<float> LAv2, CPU parity), frames of one view back to back: kernel ms of every frame (HIP events
on the compute stream) and the frame's CRC-32, for the library named by FSMI355_LIB (default: the in-tree build).
Usage: [FSMI355_LIB=path] python tools/c3_ab.py [--frames 8]"""
import argparse
import json
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, PARITY_CPU, T_HDR32, inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
a = ap.parse_args()
W, H = 3840, 2160
v = inputs.View.builtin(5, W, H, antialiasing=1)
o = inputs.Orbit(v)
la = inputs.LATable(o, host_threads=16)
co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(o)]
r = GPURenderer(0)
assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, o, 0, None, la) == 0
ms = []
for _ in range(a.frames + 1):
    assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
    assert r.SyncComputeStream() == 0
    ms.append(round(r.last_kernel_ms(), 2))
out = r.new_iter_buffer()
assert r.RenderCurrent(v.num_iterations, out) == 0
r.SyncComputeStream()
crc = zlib.crc32(np.ascontiguousarray(out[:H, :W]).astype("<u4").tobytes()) & 0xFFFFFFFF
print(json.dumps({"lib": os.path.basename(os.environ.get("FSMI355_LIB", "libfsmi355.so")), "cold_ms": ms[0], "warm_ms": ms[1:],
                  "warm_median": float(np.median(ms[1:])), "warm_min": min(ms[1:]), "frame_crc32": "%08x" % crc}), flush=True)
