"""Empirical side of the block bound of k_make_quiet_orbit: "max(max|w|, max|dc|) at a block's first entry within
the block bound  =>  each of its four arrivals passes its own bound test".  Needs the verification build
(FS_VERIFY_BLOCK_BOUND=1 in the environment for the build AND for this run: the hand-scheduled untested loop is compiled out, every
block runs the tested form, and a block that passes the block test while one of its arrivals fails its bound test is counted).
Frames: C3's view at 1920x1080 in both parity modes and perturbation only at 480x270 (k_perturb_scalar's float path, C2's
kernel), every built-in view below two million orbit entries (LAv2 Full and perturbation only) plus generated ones at 96x54.  The count must be zero.  Usage: FS_VERIFY_BLOCK_BOUND=1 python tools/block_bound_check.py"""
import ctypes as C
import json
import os
import sys
from decimal import Decimal, getcontext

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_PO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR32, _build, inputs  # noqa: E402

assert os.environ.get("FS_VERIFY_BLOCK_BOUND") == "1", "build and run with FS_VERIFY_BLOCK_BOUND=1"
_build.build_render()
r = GPURenderer(0)
total = {"blocks": 0, "blocks_passing_the_block_test": 0, "violations": 0}


def check(name, v, mode, parity, cap=None):
    ob = inputs.Orbit(v)
    if ob.count > 2_000_000:
        return
    la = inputs.LATable(ob)
    co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(ob)]
    n = v.num_iterations if cap is None else min(v.num_iterations, cap)
    assert r.InitializeMemory(v.width, v.height, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(0, ob, 0, None, la) == 0
    r.enable_step_count(True)
    assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR32, Mode=mode, parity=parity) == 0
    assert r.SyncComputeStream() == 0
    raw = (C.c_uint64 * 32)()
    assert r._lib.fs_read_stats_raw(r._h, raw, 32) == 0
    r.enable_step_count(False)
    # LAv2 Full runs k_lav2_hdr32_fast (per-wave counts in [9], [8], [15]); perturbation only runs k_perturb_scalar's float
    # path (per-lane counts: [9] blocks, [8] lane-steps of the blocks that pass, [10] violations)
    po = mode == LAV2_PO
    blocks, passing, viol = (raw[9], raw[8] // 4, raw[10]) if po else (raw[9], raw[8], raw[15])
    total["blocks"] += blocks
    total["blocks_passing_the_block_test"] += passing
    total["violations"] += viol
    print(json.dumps({"frame": name, "counted_per": "lane" if po else "wave", "blocks": blocks, "passing_the_block_test": passing,
                      "violations": viol}), flush=True)


v5 = inputs.View.builtin(5, 1920, 1080, antialiasing=1)
check("view5 1920x1080 full cpu", v5, LAV2_FULL, PARITY_CPU)
check("view5 1920x1080 full gpustage", v5, LAV2_FULL, PARITY_CPU_GPUSTAGE)
check("view5 480x270 po cap 300000", inputs.View.builtin(5, 480, 270, antialiasing=1), LAV2_PO, PARITY_CPU, cap=300000)
for nview in sorted(inputs.builtin_views()):
    if nview in (10, 15, 22):
        continue
    v = inputs.View.builtin(nview, 96, 54, antialiasing=1)
    try:
        check("view%d full" % nview, v, LAV2_FULL, PARITY_CPU)
        check("view%d po" % nview, v, LAV2_PO, PARITY_CPU, cap=60000)
    except Exception as e:  # a view the float-exponent inputs cannot express
        print(json.dumps({"frame": "view%d" % nview, "skipped": str(e)}))
getcontext().prec = 80
for ci, (cx, cy) in enumerate([("-0.5482057480704757084582125675467330293766992786373239", "-0.5775708389036038428051089822018505586755517268027721"),
                               ("-0.1528465308235274786391493323577", "1.0397032701234428320367513768879")]):
    for wd in ("1e-8", "1e-14", "1e-22", "1e-31", "1e-40"):
        cxd, cyd, w = Decimal(cx), Decimal(cy), Decimal(wd)
        h = w * 54 / 96
        v = inputs.View(str(cxd - w / 2), str(cyd - h / 2), str(cxd + w / 2), str(cyd + h / 2), 96, 54, num_iterations=50000)
        check("gen%d_%s full" % (ci, wd), v, LAV2_FULL, PARITY_CPU)
print(json.dumps(total))
sys.exit(1 if total["violations"] else 0)
