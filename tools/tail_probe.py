import sys, json
sys.path.insert(0, '/root/repo')
import numpy as np
from fractalshark_amd import GPURenderer, LAV2_FULL, PARITY_CPU, T_HDR32, inputs
v = inputs.View.builtin(5, 3840, 2160, antialiasing=1)
o = inputs.Orbit(v); la = inputs.LATable(o, host_threads=16)
co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(o)]
r = GPURenderer(0)
assert r.InitializeMemory(3840, 2160, 1, None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, o, 0, None, la) == 0
r.enable_step_count(True)
assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
r.SyncComputeStream(); print(r.read_step_count())
r.enable_step_count(False)
out = r.new_iter_buffer(); r.RenderCurrent(v.num_iterations, out); r.SyncComputeStream()
img = out[:2160, :3840].astype(np.int64)
print("iter max", img.max(), "mean", img.mean(), "pct", np.percentile(img, [50, 90, 99, 99.9, 99.99]).tolist())
t = img.reshape(270, 8, 480, 8).max(axis=(1, 3))
print("tile-max pct", np.percentile(t, [50, 90, 99, 99.9, 100]).tolist(), "sum tile-max*64 / sum", t.sum()*64/img.sum())
# where are the heavy tiles: by band row
rows = t.max(axis=1); print("heaviest tile rows (of 270):", np.argsort(rows)[-10:].tolist(), np.sort(rows)[-10:].tolist())
for world in (16, 64, 270):
    times = []
    for rank in range(0, world, max(1, world // 8)):
        r.SetRowBands(rank * 8, 8, world * 8)
        r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU); r.SyncComputeStream()
        times.append(round(r.last_kernel_ms(), 3))
    print(world, times)
