"""C4 (View 14, zoom 2^-21645): per-pixel difference between the frame of GpuHDRx2x32PerturbedLAv2 (HDRFloat<CudaDblflt>, the
BASELINE config as specified) and the frame of GpuHDRx64PerturbedLAv2 (HDRFloat<double>) from the SAME orbit and LA table
(UseSmallExponents, the table the 2x32 inputs are converted from), LA stage test in the GPU direction for both.
On MI355X FP64 runs at the non-packed FP32 rate, so the 48-bit double-float type -- ~200 binary32 operations per AT
iteration -- loses to plain double by 5x (679 vs 135 ms); this histogram is what a caller gives up by taking the HDR64
kernel instead.  Usage: python tools/c4_2x32_vs_hdr64.py [width height]   (default 3840 2160, antialiasing 4)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import (GPURenderer, LAV2_FULL, PARITY_CPU_GPUSTAGE, T_HDR2X32, T_HDR64, inputs)  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
v = inputs.View.builtin(14, w, h, antialiasing=None)
AA = v.antialiasing
W, H = w * AA, h * AA
o = inputs.Orbit(v, is64=True)
la = inputs.LATable(o, use_small_exponents=True)
o2 = inputs.Orbit2x32(o)
la2 = inputs.LATable2x32(la)
r = GPURenderer(0)
assert r.InitializeMemory(W, H, AA, None, 0, 0, 0, False) == 0
n = v.num_iterations

assert r.InitializePerturb(1, o, 0, None, la) == 0
co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(o)]
assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR64, Mode=LAV2_FULL, parity=PARITY_CPU_GPUSTAGE) == 0
f64 = r.new_iter_buffer()
assert r.RenderCurrent(n, f64) == 0 and r.SyncComputeStream() == 0
ms64 = r.last_kernel_ms()

assert r.InitializePerturb(2, o2, 0, None, la2) == 0
co2 = [(float(c["head"]), float(c["tail"]), int(c["e"])) for c in v.coords_perturb_2x32(o2)]
assert r.RenderPerturbLAv2(None, None, None, *co2, n, T=T_HDR2X32, Mode=LAV2_FULL) == 0
f2 = r.new_iter_buffer()
assert r.RenderCurrent(n, f2) == 0 and r.SyncComputeStream() == 0
ms2 = r.last_kernel_ms()

d = f64[:H, :W].astype(np.int64) - f2[:H, :W].astype(np.int64)  # HDR64 minus 2x32
vals, counts = np.unique(d, return_counts=True)
order = np.argsort(-counts)
hist = {str(int(vals[k])): int(counts[k]) for k in order[:24]}
print(json.dumps({"frame": "view14 %dx%d (aa %d)" % (W, H, AA), "pixels": int(W * H), "kernel_ms_hdr64": round(ms64, 2),
                  "kernel_ms_2x32": round(ms2, 2), "identical_pixels": int((d == 0).sum()),
                  "identical_fraction": round(float((d == 0).mean()), 6),
                  "within_3": round(float((np.abs(d) <= 3).mean()), 6),
                  "max_abs_difference": int(np.abs(d).max()),
                  "mean_abs_difference": round(float(np.abs(d).mean()), 4),
                  "histogram_hdr64_minus_2x32_top24": hist}))
