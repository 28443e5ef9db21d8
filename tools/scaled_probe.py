"""Where the C3 perturbation steps go: scaled runs / exponent-tracking runs / careful steps."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fractalshark_amd import GPURenderer, LAV2_FULL, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR32, inputs
v = inputs.View.builtin(5, 3840, 2160, antialiasing=1)
o = inputs.Orbit(v); la = inputs.LATable(o, host_threads=16)
co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(o)]
r = GPURenderer(0)
assert r.InitializeMemory(3840, 2160, 1, None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, o, 0, None, la) == 0
for parity in (PARITY_CPU, PARITY_CPU_GPUSTAGE):
    for variant in (0, 2):
        r.set_kernel_variant(variant)
        r.enable_step_count(True)
        assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=parity) == 0
        r.SyncComputeStream(); st = r.read_step_count(); r.enable_step_count(False)
        ms = []
        for _ in range(3):
            r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=parity); r.SyncComputeStream()
            ms.append(r.last_kernel_ms())
        st["ms"] = round(min(ms), 3); st["variant"] = variant; st["parity"] = parity
        st["steps_per_run"] = round(st["scaled_steps"] / max(1, st["scaled_runs"]), 1)
        print(json.dumps(st), flush=True)
