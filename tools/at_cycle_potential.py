#!/usr/bin/env python3
"""How soon does the AT iteration z -> z*z + c of a pixel that never escapes repeat a state bit for bit?  (CPU only.)
View 14 (C4): c of every pixel of a coarse grid from the view's own ATInfo (c = dc * CCoeff + RefC, ATInfo.h:155-188), the
iteration in binary64 as the kernel's steady-state loop runs it, Brent's cycle search sampled every `chunk` iterations.
Reported: pixels that run the whole ATMaxIt, how many of them lock into an exact cycle, and after how many iterations."""
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import inputs  # noqa: E402

W, H = 160, 90
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 64
v = inputs.View.builtin(14, W, H, antialiasing=1)
ob = inputs.Orbit(v, is64=True)
la = inputs.LATable(ob)
at = la.at
co = v.coords_perturb(ob)  # dx, dy, centerX, centerY as {m, e}
dx, dy, cx, cy = [(float(c["m"]), int(c["e"])) for c in co]
step = int(at.StepLength)
at_max = v.num_iterations // step
esc = math.ldexp(at.SqrEscapeRadius.m, at.SqrEscapeRadius.e)
thr_c = (at.ThresholdC.m, at.ThresholdC.e)
ys, xs = np.mgrid[0:H, 0:W]
# pixel delta (HDR): dre = dx * x - centerX, dim = -dy * y - centerY  -> as mantissa * 2^e with a common exponent e0
e0 = max(dx[1], cx[1], dy[1], cy[1]) + 12
dre = dx[0] * xs * math.ldexp(1.0, dx[1] - e0) - cx[0] * math.ldexp(1.0, cx[1] - e0)
dim = -dy[0] * ys * math.ldexp(1.0, dy[1] - e0) - cy[0] * math.ldexp(1.0, cy[1] - e0)
cc = complex(at.CCoeff.re, at.CCoeff.im)
sc = math.ldexp(1.0, at.CCoeff.e + e0)
refc = complex(math.ldexp(at.RefC.re, at.RefC.e), math.ldexp(at.RefC.im, at.RefC.e))
c = (dre + 1j * dim) * cc * sc + refc
c = c.ravel()
z = np.zeros_like(c)
alive = np.ones(c.size, bool)         # not escaped
found = np.zeros(c.size, np.int64)    # iteration at which an exact repeat was seen (0 = none)
saved = z.copy()
saved_at = np.zeros(c.size, np.int64)
next_save = np.full(c.size, chunk, np.int64)
it = 0
limit = min(at_max, 20000)
while it < limit and (alive & (found == 0)).any():
    with np.errstate(all="ignore"):
        for _ in range(chunk):
            z = np.where(alive & (found == 0), z * z + c, z)
    it += chunk
    with np.errstate(all="ignore"):
        n2 = z.real * z.real + z.imag * z.imag
        alive &= np.isfinite(n2) & ~(n2 > esc)
        z = np.where(alive, z, 0)  # (an escaped pixel is done: keep its slot quiet)
    hit = alive & (found == 0) & (z.real.view(np.int64) == saved.real.view(np.int64)) & (z.imag.view(np.int64) == saved.imag.view(np.int64))
    found[hit] = it
    sv = alive & (found == 0) & (it >= next_save)
    saved = np.where(sv, z, saved)
    saved_at[sv] = it
    next_save[sv] = it * 2
never = alive
f = found[never & (found > 0)]
print(json.dumps({"view": 14, "grid": "%dx%d" % (W, H), "at_step_length": step, "at_max_iterations": int(at_max), "simulated_to": int(limit),
                  "chunk": chunk, "pixels": int(c.size), "pixels_not_escaped_by_then": int(never.sum()),
                  "of_them_in_an_exact_cycle": int((never & (found > 0)).sum()),
                  "detected_at_iteration_percentiles_50_90_99_max": [int(np.percentile(f, p)) for p in (50, 90, 99, 100)] if f.size else None,
                  "mean_iterations_until_detected": float(f.mean()) if f.size else None,
                  "at_iterations_with_the_search_over_without": float((np.where(never & (found > 0), found, np.where(never, limit, 0))).sum() / max(1, never.sum() * limit))}))
