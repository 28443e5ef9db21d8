"""C4 through HDRFloat<double> (default) or HDRFloat<CudaDblflt> (argument "2x32"): where the frame's time goes, by LAv2 mode
(View 14, 15360x8640 = 3840x2160 x AA 4, GPU-direction stage test): Full (AT + LA stages + perturbation steps) against LA only
(AT + LA stages), with the step counters of the Full frame.  Usage: python tools/c4_phase_split.py [2x32]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_LAO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR2X32, T_HDR64, inputs  # noqa: E402

X2 = len(sys.argv) > 1 and sys.argv[1] == "2x32"

r = GPURenderer(0)
v = inputs.View.builtin(14, 3840, 2160, antialiasing=None)
ob = inputs.Orbit(v, is64=True)
la = inputs.LATable(ob, use_small_exponents=X2)
AA = v.antialiasing
W, H = v.width * AA, v.height * AA
assert r.InitializeMemory(W, H, AA, None, 0, 0, 0, False) == 0
if X2:
    o2, la2 = inputs.Orbit2x32(ob), inputs.LATable2x32(la)
    co = [(float(c["head"]), float(c["tail"]), int(c["e"])) for c in v.coords_perturb_2x32(o2)]
    assert r.InitializePerturb(1, o2, 0, None, la2) == 0
    T_HDR64, PARITY_CPU_GPUSTAGE = T_HDR2X32, PARITY_CPU  # noqa: F811 (the 2x32 kernel has the GPU-direction stage test only)
else:
    co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(ob)]
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
out = {"frame": "%dx%d" % (W, H), "type": "HDRFloat<CudaDblflt>" if X2 else "HDRFloat<double>"}
for name, mode in (("full", LAV2_FULL), ("la_only", LAV2_LAO)):
    ms = []
    for _ in range(3):
        assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR64, Mode=mode, parity=PARITY_CPU_GPUSTAGE) == 0
        assert r.SyncComputeStream() == 0
        ms.append(r.last_kernel_ms())
    out["kernel_ms_" + name] = round(min(ms), 3)
r.enable_step_count(True)
assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR64, Mode=LAV2_FULL, parity=PARITY_CPU_GPUSTAGE) == 0
assert r.SyncComputeStream() == 0
st = r.read_step_count()
out.update({k: int(st[k]) for k in ("perturb_steps", "at_iterations", "la_steps", "lane_slots") if k in st})
print(json.dumps(out))
