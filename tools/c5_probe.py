import sys, json
sys.path.insert(0, '/root/repo')
from fractalshark_amd import GPURenderer, T_HDR32, inputs
v = inputs.View.builtin(19, 7680, 4320, antialiasing=1)
o = inputs.Orbit(v); bla = inputs.BLATable(o)
co = v.coords_perturb(o)
r = GPURenderer(0)
assert r.InitializeMemory(7680, 4320, 1, None, 0, 0, 0, False) == 0
lib = r._lib
assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, o.data_ptr, o.count, o.count, o.period) == 0
assert lib.fs_upload_bla(r._h, T_HDR32, bla.level_ptrs, bla.level_sizes, bla.num_levels, bla.lm2) == 0
r.enable_step_count(True)
assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
r.SyncComputeStream(); st = r.read_step_count(); print(json.dumps(st), r.last_kernel_ms())
