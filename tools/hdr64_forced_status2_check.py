#!/usr/bin/env python3
"""The exits of the hand-written HDRFloat<double> statements that no view reaches by itself (status 2: a norm too small for the value
compare, an abnormal complex0 at a rebase) -- exercised by a build whose threshold makes EVERY step take them
(tools/build_variant.py h64st2 kernels_hdr64.hip -DFS_H64_ASM_TINY=1e300; FSMI355_LIB=build/ab/libfsmi355_h64st2.so): the
production kernel of that build against the literal kernel (FS_VARIANT_LITERAL, which has no statements) over a few parity
cases.  Exit code 0 = every frame identical."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_hdr64_fast as T  # noqa: E402
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_LAO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR64, inputs  # noqa: E402

assert "h64st2" in os.environ.get("FSMI355_LIB", ""), "run with FSMI355_LIB pointing at the status-2 build"
r = GPURenderer(0)
bad = 0
for n_view in (3, 5, 7, 14, 17, 19):
    v = inputs.View.builtin(n_view, T.W, T.H, antialiasing=1)
    ob = inputs.Orbit(v, is64=True)
    la = inputs.LATable(ob)
    n = min(v.num_iterations, T.CAP)
    co = T._pairs(v.coords_perturb(ob))
    assert r.InitializeMemory(T.W, T.H, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(0, ob, 0, None, la) == 0
    for mode, parity in ((LAV2_FULL, PARITY_CPU_GPUSTAGE), (LAV2_FULL, PARITY_CPU), (LAV2_LAO, PARITY_CPU_GPUSTAGE)):
        assert r.set_kernel_variant(0) == 0
        a = T._render(r, co, n, mode, parity)
        assert r.set_kernel_variant(1) == 0
        b = T._render(r, co, n, mode, parity)
        same = bool(np.array_equal(a, b))
        bad += not same
        print("view %d mode %d parity %d: %s" % (n_view, mode, parity, "identical" if same else "DIFFERENT (%d pixels)" % int((a != b).sum())), flush=True)
r.set_kernel_variant(0)
r.close()
sys.exit(1 if bad else 0)
