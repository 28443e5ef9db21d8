"""What the tuned scaled kernel's steps are made of on the c4_scaled workload (View 14, 3840x2160, cap 65 536):
binary32 steps inside wave-voted runs, binary32 steps through the literal code, full-precision steps, rescales."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, inputs  # noqa: E402

W, H, CAP = int(os.environ.get("W", 3840)), int(os.environ.get("H", 2160)), 65536
v = inputs.View.builtin(14, W, H, antialiasing=1)
ob = inputs.Orbit(v)
co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(ob)]
r = GPURenderer(0)
assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
for variant in (1, 0):
    r.set_kernel_variant(variant)
    for stats in (False, True):
        r.enable_step_count(stats)
        assert r.RenderPerturbBLAScaled(None, ob, ob, None, None, *co, CAP) == 0
        r.SyncComputeStream()
        d = {"variant": "literal" if variant else "tuned", "instrumented": stats, "kernel_ms": round(r.last_kernel_ms(), 2)}
        if stats:
            st = r.read_step_count()
            d.update({"rescales": st["at_iterations"], "full_steps": st["la_steps"], "float_steps": st["perturb_steps"],
                      "float_steps_in_runs": st["scaled_steps"], "runs": st["scaled_runs"],
                      "lane_slots": st["lane_slots"]})
            if st["scaled_runs"]:
                d["steps_per_run"] = round(st["scaled_steps"] / st["scaled_runs"], 1)
                d["fraction_in_runs"] = round(st["scaled_steps"] / max(1, st["perturb_steps"]), 4)
        print(json.dumps(d))
