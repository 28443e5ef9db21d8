#!/usr/bin/env python3
"""RenderCurrent's two kernels (antialias + palette, min / max / sum) at BASELINE config C4's geometry
(3840x2160 x AA4 = 15360x8640 iteration buffer = 531 MB): achieved HBM GB/s against the 8 TB/s roof.
Algorithmic bytes: antialias = 4*AA^2 B read + 8 B written per colour pixel; reduce = 4 B per element.
The iteration buffer is filled by the direct double kernel on View 0 (any content does).

  python tools/bench_render_current.py [--width 3840 --height 2160 --aa 4 --repeats 20]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from fractalshark_amd import GPURenderer, T_F64, _capi, inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--aa", type=int, default=4)
ap.add_argument("--repeats", type=int, default=20)
ap.add_argument("--iter-bytes", type=int, default=4)
args = ap.parse_args()

PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md
v = inputs.View.builtin(0, args.width, args.height, antialiasing=args.aa)
W, H = args.width * args.aa, args.height * args.aa
# Default palette, depth 8 (7 << 8 entries), restated here so the tool does not need the test oracle
pal = []
cur = (0, 0, 0)
mv = 65535
for tgt in ((mv, 0, 0), (mv, mv, 0), (0, mv, 0), (0, mv, mv), (0, 0, mv), (mv, 0, mv), (0, 0, 0)):
    d = [(tgt[k] - cur[k]) / 256 for k in range(3)]
    for i in range(256):
        pal.append(tuple(int(cur[k] + d[k] * (i + 1)) & 0xFFFF for k in range(3)) + (0,))
    cur = pal[-1][:3]
pal = np.array(pal, np.uint16)
r = GPURenderer(0)
assert r.InitializeMemory(W, H, args.aa, pal, len(pal), 0, 1, False, iter_bytes=args.iter_bytes) == 0
dx, dy, minx, maxy = v.coords_direct_f64(args.aa)
assert r.Render(None, minx, maxy, dx, dy, v.num_iterations, T=T_F64) == 0
assert r.SyncComputeStream() == 0
ms = (C.c_float * 2)()
assert r._lib.fs_time_render_current(r._h, v.num_iterations, 3, ms) == 0  # warm-up
assert r._lib.fs_time_render_current(r._h, v.num_iterations, args.repeats, ms) == 0
red = _capi.Reduction()
assert r.RenderCurrent(v.num_iterations, None, None, red) == 0
assert r.SyncComputeStream() == 0
rw = r.rounded_width
ib = args.iter_bytes
aa_bytes = W * H * ib + (W // args.aa) * (H // args.aa) * 8
red_bytes = rw * H * ib
out = {"geometry": "%dx%d x AA%d = %dx%d, IterType %d B" % (args.width, args.height, args.aa, W, H, ib),
       "k_antialias": {"ms": round(ms[0], 4), "algorithmic_bytes": aa_bytes, "GB_s": round(aa_bytes / ms[0] / 1e6, 1),
                       "frac_of_hbm_peak": round(aa_bytes / ms[0] / 1e6 / PEAK, 4)},
       "k_reduce": {"ms": round(ms[1], 4), "algorithmic_bytes": red_bytes, "GB_s": round(red_bytes / ms[1] / 1e6, 1),
                    "frac_of_hbm_peak": round(red_bytes / ms[1] / 1e6 / PEAK, 4)},
       "reduction": {"min": red.Min, "max": red.Max, "sum": red.Sum}, "peak_GB_s": PEAK}
print(json.dumps(out))
r.close()
