"""HBM occupied by a long reference orbit in the two SimpleCompression modes (fs_set_compressed_orbit_mode), and a frame
rendered from each.

Synthetic input: the orbit of c = -0.75, which never escapes, taken to N entries (default 1e8)
without periodicity detection, compressed by the host's RefOrbitCompressor restatement with error exponent 20.
  mode 0  the waypoints are expanded on upload: (N + 2) x 16 B prepared entries + 2 x (N + 2) x 16 B companion arrays;
  mode 1  only the waypoints stay resident (24 B each) and the LAv2 kernel decompresses as it walks the orbit.
The same 64 x 36 perturbation-only frame is rendered from both (iteration cap 20 000) and must be identical.
Usage: python tools/orbit_memory_modes.py [N]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_PO, PARITY_CPU_GPUSTAGE, T_HDR32, inputs  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
W, H, CAP = 64, 36, 20000
from decimal import Decimal, getcontext  # noqa: E402
getcontext().prec = 80
cx, w = Decimal("-0.75"), Decimal("1e-30")
hh = w * H / W
v = inputs.View(str(cx - w / 2), str(-hh / 2), str(cx + w / 2), str(hh / 2), W, H, num_iterations=CAP)
t0 = time.time()
ob = inputs.Orbit(v, max_iter=N, periodicity=False, compression_exp=20)
t_orbit = time.time() - t0
assert ob.compressed
co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(ob)]
r = GPURenderer(0)
out = {"orbit_entries": ob.count, "waypoints": ob.compressed_count, "host_orbit_build_s": round(t_orbit, 1)}
frames = {}
for mode, name in ((1, "runtime_decompression"), (0, "expanded_on_upload")):
    assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
    assert r.set_compressed_orbit_mode(mode == 1) == 0
    t0 = time.time()
    assert r.InitializePerturb(0, ob, 0, None, None) == 0
    t_up = time.time() - t0
    assert r.RenderPerturbLAv2(None, None, None, *co, CAP, T=T_HDR32, Mode=LAV2_PO, parity=PARITY_CPU_GPUSTAGE) == 0
    buf = r.new_iter_buffer()
    assert r.RenderCurrent(CAP, buf) == 0 and r.SyncComputeStream() == 0
    frames[name] = buf
    out[name] = {"orbit_hbm_bytes": r.orbit_device_bytes, "upload_s": round(t_up, 3), "kernel_ms": round(r.last_kernel_ms(), 3),
                 "frame_checksum": int(buf[:H, :W].astype(np.uint64).sum())}
r.set_compressed_orbit_mode(False)
out["frames_identical"] = bool(np.array_equal(frames["runtime_decompression"], frames["expanded_on_upload"]))
out["hbm_ratio_expanded_over_compressed"] = round(out["expanded_on_upload"]["orbit_hbm_bytes"] /
                                                  max(1, out["runtime_decompression"]["orbit_hbm_bytes"]), 1)
print(json.dumps(out))
