"""C2 (View 5, perturbation only, 1920x1080): every 8-row band that holds pixels at the iteration cap, rendered alone (each
of its waves then has a SIMD to itself): kernel ms per band = the pace of that band's slowest chain.  The frame cannot be
faster than the slowest of them."""
import json
import os
import sys

import numpy as np

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
for _ in range(2):
    assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
    r.SyncComputeStream()
frame_ms = r.last_kernel_ms()
out = r.new_iter_buffer()
assert r.RenderCurrent(v.num_iterations, out) == 0
r.SyncComputeStream()
cap = out[:H, :W] >= v.num_iterations
per_band = cap.reshape(H // 8, 8, W).sum(axis=(1, 2))
print(json.dumps({"frame_ms": round(frame_ms, 2), "pixels_at_cap": int(cap.sum())}), flush=True)
for band in np.nonzero(per_band)[0]:
    assert r.SetRowBands(int(band) * 8, 8, H) == 0
    for _ in range(2):
        assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
        r.SyncComputeStream()
    ms = r.last_kernel_ms()
    tiles = cap[band * 8:band * 8 + 8].reshape(8, W // 8, 8).any(axis=(0, 2))
    print(json.dumps({"band": int(band), "pixels_at_cap": int(per_band[band]), "tiles_with_cap_pixels": int(tiles.sum()),
                      "band_alone_kernel_ms": round(ms, 2), "ns_per_step_of_longest_chain": round(ms * 1e6 / v.num_iterations, 2)}),
          flush=True)
