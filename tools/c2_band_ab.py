"""C2: chosen 8-row bands alone, the plain and the step-counting instantiation of k_perturb_scalar alternately (kernel ms)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, T_HDR32, inputs  # noqa: E402

W, H = 1920, 1080
v = inputs.View.builtin(5, W, H, antialiasing=1)
o = inputs.Orbit(v)
co = v.coords_perturb(o)
r = GPURenderer(0)
assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
lib = r._lib
assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, o.data_ptr, o.count, o.count, o.period) == 0
for band in [int(b) for b in (sys.argv[1:] or ["69", "72", "73"])]:
    assert r.SetRowBands(band * 8, 8, H) == 0
    res = {"band": band, "plain_ms": [], "counting_ms": []}
    for rep in range(3):
        for stats in (False, True):
            r.enable_step_count(stats)
            assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
            r.SyncComputeStream()
            res["counting_ms" if stats else "plain_ms"].append(round(r.last_kernel_ms(), 2))
    print(json.dumps(res), flush=True)
