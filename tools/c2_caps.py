import json, os, sys
sys.path.insert(0, '/root/repo')
from fractalshark_amd import GPURenderer, T_HDR32, inputs
W, H = 1920, 1080
v = inputs.View.builtin(5, W, H, antialiasing=1)
o = inputs.Orbit(v)
co = v.coords_perturb(o)
r = GPURenderer(0)
assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
lib = r._lib
assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, o.data_ptr, o.count, o.count, o.period) == 0
for cap in (16384, 65536, 262144, 1048576, 2097152, v.num_iterations):
    ms = []
    for _ in range(3):
        assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, cap) == 0
        r.SyncComputeStream()
        ms.append(round(r.last_kernel_ms(), 2))
    print(json.dumps({"cap": cap, "kernel_ms": ms, "ns_per_cap_step": round(min(ms) * 1e6 / cap, 1)}))
