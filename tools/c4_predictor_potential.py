#!/usr/bin/env python3
"""C4 (View 14, 15360 x 8640, HDRFloat<double> LAv2): how much of what the recorded pixel order gains (lanes of a wave run equally
long) could a PREDICTED order gain on the first frame of a view?  The frame's own counts stand for the cost; waves are 64
consecutive pixels of: 8 x 8 tiles (today's cold frame), the order of the actual counts (today's warm frame), the order of counts
predicted from one sample per 8 x 8 / 16 x 16 / 32 x 32 tile (nearest and bilinear).  Reported: wave-passes = sum over waves of the
longest lane, relative to the tile mapping.   python tools/c4_predictor_potential.py [scale-down factor]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fractalshark_amd import GPURenderer, LAV2_FULL, PARITY_CPU_GPUSTAGE, T_HDR64  # noqa: E402

div = int(sys.argv[1]) if len(sys.argv) > 1 else 1
inp = bench.make_inputs("c4_hdr64", width=3840 // div, height=2160 // div)
W, H, AA, n_iter = inp["W"], inp["H"], inp["AA"], inp["n_iter"]
r = GPURenderer(0)
assert r.InitializeMemory(W, H, AA, None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, inp["orbit"], 0, None, inp["la"]) == 0
assert r.RenderPerturbLAv2(None, None, None, *inp["coords"], n_iter, T=T_HDR64, Mode=LAV2_FULL, parity=PARITY_CPU_GPUSTAGE) == 0
buf = r.new_iter_buffer()
assert r.RenderCurrent(n_iter, buf) == 0
assert r.SyncComputeStream() == 0
ms = r.last_kernel_ms()
H8, W8 = H // 32 * 32, W // 32 * 32
c = buf[:H8, :W8].astype(np.float32)
del buf


def passes(order):
    s = c.ravel()[order]
    n = s.size // 64 * 64
    return float(s[:n].reshape(-1, 64).max(axis=1).sum())


tiles = c.reshape(H8 // 8, 8, W8 // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
now = float(tiles.max(axis=1).sum())
ideal = float(c.sum() / 64.0)
res = {"frame": "%dx%d" % (W8, H8), "kernel_ms": round(ms, 2), "tiles_8x8": 1.0, "full_occupancy": round(ideal / now, 4),
       "sorted_by_actual_count": round(passes(np.argsort(-c.ravel(), kind="stable")) / now, 4)}
yy, xx = np.mgrid[0:H8, 0:W8].astype(np.float32)
for t in (8, 16, 32):
    coarse = c[t // 2::t, t // 2::t]  # one sample per t x t tile (its centre)
    near = np.repeat(np.repeat(coarse, t, axis=0), t, axis=1)
    res["nearest_%d" % t] = round(passes(np.argsort(-near.ravel(), kind="stable")) / now, 4)
    # bilinear between the samples (clamped at the frame's edge)
    gy = np.clip((yy - t / 2) / t, 0, coarse.shape[0] - 1)
    gx = np.clip((xx - t / 2) / t, 0, coarse.shape[1] - 1)
    y0 = np.floor(gy).astype(np.int32)
    x0 = np.floor(gx).astype(np.int32)
    y1 = np.minimum(y0 + 1, coarse.shape[0] - 1)
    x1 = np.minimum(x0 + 1, coarse.shape[1] - 1)
    fy, fx = gy - y0, gx - x0
    pred = (coarse[y0, x0] * (1 - fy) * (1 - fx) + coarse[y0, x1] * (1 - fy) * fx + coarse[y1, x0] * fy * (1 - fx) +
            coarse[y1, x1] * fy * fx)
    res["bilinear_%d" % t] = round(passes(np.argsort(-pred.ravel(), kind="stable")) / now, 4)
    del pred, near, gy, gx, y0, x0, y1, x1, fy, fx
print(json.dumps(res), flush=True)
