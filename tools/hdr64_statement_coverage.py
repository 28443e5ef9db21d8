#!/usr/bin/env python3
"""Which exits of the hand-written HDRFloat<double> statements (la_step_asm.hpp, pt_step_asm.hpp) the parity cases of
tests/test_gpu_hdr64_fast.py actually take: the same views and modes through the probe build of the library, exits by status summed
per view.  A status no case reaches is a path parity has not seen.
  python tools/build_variant.py h64dbg kernels_hdr64.hip -DFS_H64_LA_ASM_DEBUG=1
  FSMI355_LIB=$PWD/build/ab/libfsmi355_h64dbg.so python tools/hdr64_statement_coverage.py"""
import ctypes as C
import json
import os
import sys
from decimal import Decimal, getcontext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_hdr64_fast as T  # noqa: E402
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_LAO, LAV2_PO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR64, inputs  # noqa: E402

r = GPURenderer(0)
total = [0] * 8
for name, builtin, gen in T._views():
    if builtin is not None:
        v = inputs.View.builtin(builtin, T.W, T.H, antialiasing=1)
    else:
        getcontext().prec = 80
        (cx, cy), wd = gen
        cxd, cyd, w = Decimal(cx), Decimal(cy), Decimal(wd)
        h = w * T.H / T.W
        v = inputs.View(str(cxd - w / 2), str(cyd - h / 2), str(cxd + w / 2), str(cyd + h / 2), T.W, T.H, num_iterations=50000)
    ob = inputs.Orbit(v, is64=True)
    if ob.count > 2_000_000:
        continue
    la = inputs.LATable(ob)
    n = min(v.num_iterations, T.CAP)
    co = T._pairs(v.coords_perturb(ob))
    assert r.InitializeMemory(T.W, T.H, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(0, ob, 0, None, la) == 0
    acc = [0] * 8
    for mode in (LAV2_FULL, LAV2_PO, LAV2_LAO):
        for parity in (PARITY_CPU_GPUSTAGE, PARITY_CPU):
            if mode == LAV2_PO and parity == PARITY_CPU:
                continue
            r.enable_step_count(True)
            assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR64, Mode=mode, parity=parity) == 0
            assert r.SyncComputeStream() == 0
            raw = (C.c_uint64 * 40)()
            assert r._lib.fs_read_stats_raw(r._h, raw, 40) == 0
            r.enable_step_count(False)
            for k in range(8):
                acc[k] += raw[20 + k]
    for k in range(8):
        total[k] += acc[k]
    print(json.dumps({"view": name, "la_exits_status_0_1_2_3": acc[:4], "pt_exits_status_0_1_2_3": acc[4:]}), flush=True)
print(json.dumps({"view": "ALL", "la_exits_status_0_1_2_3": total[:4], "pt_exits_status_0_1_2_3": total[4:]}))
r.close()
