"""Empirical side of the every-SECOND-state floor test of the scaled runs (k_lav2_hdr32_fast, DESIGN.md 4.2, FS_FL_EVERY=0).

The every-state form (FS_FL_EVERY=1) certifies each state of a run; the shipped form tests a trip's second state only, against a
higher floor, and its argument has one gap: a trip whose untested FIRST state has a part below 2^-56 (the every-state floor) while
its second state passes.  Only on such "exposed" trips can the two forms differ at all.  The verification build
(FS_VERIFY_FLOOR=1 in the environment for the build AND for this run) runs the shipped form and records, per invocation of the
hand-scheduled loop (and per tested block), whether any lane's first state fell below 2^-56.  The count should be zero; where
it is, the frame is covered by the every-state argument.  Usage: FS_VERIFY_FLOOR=1 python tools/floor_check.py"""
import ctypes as C
import json
import os
import sys
from decimal import Decimal, getcontext

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_PO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR32, _build, inputs  # noqa: E402

assert os.environ.get("FS_VERIFY_FLOOR") == "1", "build and run with FS_VERIFY_FLOOR=1"
_build.build_render()
r = GPURenderer(0)
total = {"wave_trips": 0, "exposed_loop_invocations_or_blocks": 0}


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
    trips, exposed = 2 * (raw[8] + raw[9]), raw[15]  # [8] / [9]: 4-step blocks without / with bound tests (per wave)
    total["wave_trips"] += trips
    total["exposed_loop_invocations_or_blocks"] += exposed
    print(json.dumps({"frame": name, "wave_trips": trips, "exposed": exposed}), flush=True)


v5 = inputs.View.builtin(5, 1920, 1080, antialiasing=1)
check("view5 1920x1080 full cpu", v5, LAV2_FULL, PARITY_CPU)
check("view5 1920x1080 full gpustage", v5, LAV2_FULL, PARITY_CPU_GPUSTAGE)
for nview in sorted(inputs.builtin_views()):
    if nview in (10, 15, 22):
        continue
    v = inputs.View.builtin(nview, 96, 54, antialiasing=1)
    try:
        check("view%d full" % nview, v, LAV2_FULL, PARITY_CPU)
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
sys.exit(0)
