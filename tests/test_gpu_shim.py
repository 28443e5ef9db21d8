"""GPU: the reference-language side of the boundary EXECUTED -- a C++ program (tests/shim/shim_exec.cpp) drives
View 5 64x36 through the `GPURenderer` members of gpu_render_shim.hpp exactly as Fractal.cpp does (InitializeMemory /
InitializePerturb / RenderPerturbLAv2 Full, LAO, PO / RenderPerturbBLA / Render<double> / RenderCurrent / streams / done
callback / error codes) and every buffer that comes back is compared with the committed fixtures (golden_small.npz,
made by the golden-pinned oracle).  The reference tree does not travel to the GPU box, so the binary is built against
the stand-in types of tests/shim/standin_types.hpp there; the same header is compiled against the real reference
headers on the CPU side (tests/test_shim_real_headers.py)."""
import os
import subprocess

import numpy as np
import pytest

import _oracle
from fractalshark_amd import inputs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))


def build_shim_exec(native_libs, out_dir):
    exe = os.path.join(str(out_dir), "shim_exec")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-private-field", "-o", exe,
           os.path.join(ROOT, "tests", "shim", "shim_exec.cpp"), native_libs.LIB_RENDER,
           "-Wl,-rpath," + os.path.dirname(native_libs.LIB_RENDER), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
           "-lamdhip64", "-pthread"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    return exe


def write_inputs(d):
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    bla = inputs.BLATable(ob)
    v0 = inputs.View.builtin(0, 64, 48)
    import ctypes as C
    ob.entries().tofile(os.path.join(d, "orbit.bin"))
    la.records().tofile(os.path.join(d, "las.bin"))
    la.stages().tofile(os.path.join(d, "stages.bin"))
    open(os.path.join(d, "at.bin"), "wb").write(bytes((C.c_char * C.sizeof(la.at)).from_address(C.addressof(la.at))))
    v.coords_perturb_hdr32(ob).tofile(os.path.join(d, "coords.bin"))
    pal = _oracle.default_palette(8)
    pal.tofile(os.path.join(d, "palette.bin"))
    dx, dy, minx, maxy = [float(x) for x in v0.coords_direct_f64()]
    miny = maxy - dy * 48
    assert miny + dy * 48 == maxy  # the shim rebuilds maxY from the min corner; exact for this view
    np.array([dx, dy, minx, miny], np.float64).tofile(os.path.join(d, "direct.bin"))
    sizes = [int(s) for s in bla.sizes()]
    for l, n in enumerate(sizes):
        if n:
            bla.level(l).tofile(os.path.join(d, "bla_%d.bin" % l))
    with open(os.path.join(d, "meta.txt"), "w") as f:
        f.write("%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d\n" % (
            64, 36, v.num_iterations, ob.count, ob.period, la.count, la.stage_count, int(la.use_at), int(la.is_valid),
            bla.num_levels, bla.lm2, len(pal), 64, 48, v0.num_iterations))
        f.write(" ".join(str(s) for s in sizes) + "\n")
    return v, pal


def test_gpurenderer_members_executed_through_the_shim(native_libs, tmp_path):
    d = str(tmp_path)
    v, pal = write_inputs(d)
    exe = build_shim_exec(native_libs, tmp_path)
    p = subprocess.run([exe, d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout
    res = [l.split(None, 1) for l in open(os.path.join(d, "result.txt")).read().splitlines()]
    codes = {}
    for k, val in res:
        codes.setdefault(k, []).append(val)
    assert int(codes["working"][0]) != 0
    assert codes["uninit_render"] == ["0"]        # memory not initialised: silent (GPU_Render.cu:1007-1009)
    assert codes["bad_aa"] == ["10002"]           # FractalSharkError::Error3
    assert codes["no_orbit"] == ["10005"]         # Error6 (GPU_Render.cu:1015-1022)
    for k in ("init", "init_perturb", "lav2_full", "lav2_lao", "lav2_po", "bla", "init0", "direct", "done_cb", "query"):
        assert codes[k] == ["0"], (k, codes[k])
    assert set(codes["current"]) == {"0"} and set(codes["sync"]) == {"0"}
    assert codes["compute_done"] == ["1"]
    assert "Error3" in codes["errstr"][0]

    def buf(name, shape):
        return np.fromfile(os.path.join(d, name), np.uint32).reshape(shape)

    full = buf("out_lav2_full.bin", (40, 64))
    assert np.array_equal(full, GOLD["view5_lav2_cpu_64x36"])
    assert np.array_equal(buf("out_lav2_lao.bin", (40, 64)), GOLD["view5_lao_cpu_64x36"])
    assert np.array_equal(buf("out_lav2_po.bin", (40, 64)), GOLD["view5_po_64x36"])
    assert np.array_equal(buf("out_bla.bin", (40, 64)), GOLD["view5_bla_64x36"])
    assert np.array_equal(buf("out_direct.bin", (48, 64)), GOLD["view0_direct_f64_64x48"])
    valid = full[:36, :64].astype(np.uint64)
    assert [int(x) for x in codes["reduction"][0].split()] == [int(valid.min()), int(valid.max()), int(valid.sum())]
    # colours: palette lookup of the iteration counts (AA 1), rows not padded (AntialiasingKernel.cuh:21)
    colors = np.fromfile(os.path.join(d, "out_colors.bin"), np.uint16).reshape(-1, 4)[:64 * 36].reshape(36, 64, 4)
    exp = np.zeros((36, 64, 4), np.uint16)
    n = full[:36, :64]
    inside = n >= v.num_iterations
    exp[..., :3] = pal[n % len(pal), :3]
    exp[inside, :3] = 0
    exp[..., 3] = 65535
    assert np.array_equal(colors, exp)
