"""HDRFloat<CudaDblflt> LAv2 kernel (k_lav2_2x32): how many lane-steps of the perturbation loop take the literal step instead of
the packed straight-line one (pt_step_pk, kernels_2x32.hip), and kernel times on a deep and three shallow views.

The count needs a library built with -DFS_2X32_PROBE:
  FS_2X32_PROBE=1 python -c "from fractalshark_amd import _build; _build.build_render()"
(and a normal rebuild afterwards); without it the count reads 0 and only the times mean anything.
Usage: python tools/x2_step_probe.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_PO, T_HDR2X32, inputs  # noqa: E402

out = {}
for view, W, H, mode, name in ((14, 960, 540, LAV2_FULL, "view14_full_aa4"), (5, 1920, 1080, LAV2_FULL, "view5_full"),
                               (5, 480, 270, LAV2_PO, "view5_po_cap20000"), (19, 1920, 1080, LAV2_FULL, "view19_full"),
                               (3, 1920, 1080, LAV2_FULL, "view3_full")):
    r = GPURenderer(0)
    v = inputs.View.builtin(view, W, H, antialiasing=None if view == 14 else 1)
    ob = inputs.Orbit(v, is64=True)
    la = inputs.LATable(ob, use_small_exponents=True)
    AA = v.antialiasing
    assert r.InitializeMemory(W * AA, H * AA, AA, None, 0, 0, 0, False) == 0
    o2, la2 = inputs.Orbit2x32(ob), inputs.LATable2x32(la)
    co = [(float(c["head"]), float(c["tail"]), int(c["e"])) for c in v.coords_perturb_2x32(o2)]
    assert r.InitializePerturb(1, o2, 0, None, la2) == 0
    n = v.num_iterations if mode != LAV2_PO else min(v.num_iterations, 20000)
    ms = []
    for _ in range(3):
        assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR2X32, Mode=mode) == 0
        assert r.SyncComputeStream() == 0
        ms.append(r.last_kernel_ms())
    r.enable_step_count(True)
    assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR2X32, Mode=mode) == 0
    assert r.SyncComputeStream() == 0
    raw = (C.c_uint64 * 32)()
    assert r._lib.fs_read_stats_raw(r._h, raw, 32) == 0
    st = r.read_step_count()
    out[name] = {"kernel_ms": round(min(ms), 3), "perturb_steps": int(st["perturb_steps"]), "literal_lane_steps": int(raw[12]),
                 "at_iterations": int(st.get("at_iterations", 0)), "at_lane_slots_of_waves_that_enter_at": int(raw[13])}
    del r
print(json.dumps(out))
