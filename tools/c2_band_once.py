"""C2: one 8-row band alone, N frames of the plain kernel (for counter runs under rocprofv3).  python tools/c2_band_once.py BAND [N]"""
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
band = int(sys.argv[1])
assert r.SetRowBands(band * 8, 8, H) == 0
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
    r.SyncComputeStream()
    print(round(r.last_kernel_ms(), 2), flush=True)
