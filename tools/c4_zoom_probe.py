#!/usr/bin/env python3
"""A zooming viewer on C4's frame (View 14, 15360x8640, HDRFloat<double> or <CudaDblflt>): every frame has new coordinates (the pixel
spacing shrinks by --zoom per frame around the same centre, same reference orbit), so no frame ever meets its own recorded order.
Kernel ms per frame (AT pass + frame kernel), and, for reference, the last view rendered cold and warm.
Round 6 used it on a PROTOTYPE that let a frame run in the order of the previous view of the same shape (FSMI355_STALE_ORDER, with a
re-sort every n-th frame): profiles/r06p_c4_zoom_stale_order_probe.jsonl -- at 2 % zoom per frame the previous view's order makes the
frame 83-104 ms against 55-59 ms with no order at all and 37.5 with its own: the count order is worth nothing one view later, a zoom's
frames ARE first frames.  The prototype is not in the library; with the library as it is this tool measures the first-frame cost of a
zoom sequence.   python tools/c4_zoom_probe.py [--frames 16] [--zoom 0.98] [--workload c4_hdr64]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fractalshark_amd import GPURenderer, LAV2_FULL, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR2X32, T_HDR64  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--zoom", type=float, default=0.98)
ap.add_argument("--workload", default="c4_hdr64", choices=["c4_hdr64", "c4_2x32"])
a = ap.parse_args()
inp = bench.make_inputs(a.workload)
W, H, AA = inp["W"], inp["H"], inp["AA"]
is2 = inp["is2x32"]
T = T_HDR2X32 if is2 else T_HDR64
par = PARITY_CPU if inp["parity"] == "cpu" else PARITY_CPU_GPUSTAGE
r = GPURenderer(0)
assert r.InitializeMemory(W, H, AA, None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, inp["orbit2"] if is2 else inp["orbit"], 0, None, inp["la2"] if is2 else inp["la"]) == 0
base = inp["coords"]


def coords(scale):
    if is2:  # (head, tail, e): scale head and tail alike
        return [(base[0][0] * scale, base[0][1] * scale, base[0][2]), (base[1][0] * scale, base[1][1] * scale, base[1][2]), base[2], base[3]]
    return [(base[0][0] * scale, base[0][1]), (base[1][0] * scale, base[1][1]), base[2], base[3]]


def frame(co):
    t0 = time.perf_counter()
    assert r.RenderPerturbLAv2(None, None, None, *co, inp["n_iter"], T=T, Mode=LAV2_FULL, parity=par) == 0
    assert r.SyncComputeStream() == 0
    wall = (time.perf_counter() - t0) * 1e3
    a_, b_ = r.kernel_ms_split_history(1)
    return {"kernel_ms": round(a_[0] + b_[0], 3), "at_pass_ms": round(a_[0], 3), "wall_ms_incl_sorts": round(wall, 3),
            "ordered": bool(r.last_frame_tile_ordered())}


out = {"workload": inp["key"], "zoom_per_frame": a.zoom, "env": {k: os.environ.get(k) for k in ("FSMI355_STALE_ORDER", "FSMI355_STALE_REFRESH")}}
for _ in range(3):
    f0 = frame(coords(1.0))
out["warm_at_zoom_1"] = f0
seq = []
for k in range(1, a.frames + 1):
    seq.append(frame(coords(a.zoom ** k)))
out["zoom_frames"] = seq
out["zoom_mean_kernel_ms"] = round(sum(f["kernel_ms"] for f in seq) / len(seq), 3)
out["zoom_mean_wall_ms_incl_sorts"] = round(sum(f["wall_ms_incl_sorts"] for f in seq) / len(seq), 3)
# the last view cold and warm, for reference
r.forget_tile_costs()
out["last_view_cold"] = frame(coords(a.zoom ** a.frames))
for _ in range(3):
    lw = frame(coords(a.zoom ** a.frames))
out["last_view_warm"] = lw
print(json.dumps(out), flush=True)
r.close()
