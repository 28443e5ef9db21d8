"""C3 (View 5, 3840x2160, HDRFloat<float> LAv2, CPU-direction stage test): kernel time of the Full frame against the frame with the
AT shortcut and the LA stages only (LAV2_LAO), with the step counters -- what the prologue costs next to the perturbation steps.
Usage: python tools/c3_phase_split.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_LAO, PARITY_CPU, T_HDR32, inputs  # noqa: E402

r = GPURenderer(0)
v = inputs.View.builtin(5, 3840, 2160, antialiasing=1)
ob = inputs.Orbit(v)
la = inputs.LATable(ob)
co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(ob)]
assert r.InitializeMemory(v.width, v.height, 1, None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, ob, 0, None, la) == 0
out = {"frame": "%dx%d" % (v.width, v.height)}
for name, mode in (("full", LAV2_FULL), ("la_only", LAV2_LAO)):
    ms = []
    for _ in range(5):
        assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=mode, parity=PARITY_CPU) == 0
        assert r.SyncComputeStream() == 0
        ms.append(r.last_kernel_ms())
    out["kernel_ms_" + name] = round(min(ms), 3)
r.enable_step_count(True)
for name, mode in (("full", LAV2_FULL), ("la_only", LAV2_LAO)):
    assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=mode, parity=PARITY_CPU) == 0
    assert r.SyncComputeStream() == 0
    st = r.read_step_count()
    out["counters_" + name] = {k: int(st[k]) for k in ("perturb_steps", "at_iterations", "la_steps", "lane_slots") if k in st}
print(json.dumps(out))
