"""GPU: the HIP path pinned DIRECTLY to the reference's own golden vectors -- the CRC-64s of the twelve 256x256 PNG
renders in FractalSharkTest/TestRenderGoldens.cpp:84-97.  Each case is rendered through the C ABI with the HIP kernel
that twins the CPU RenderAlgorithm of the golden case; the iteration buffer that comes back over RenderCurrent is
encoded by the reference's own PNG writer (oracle/_ref/libpngpin.so, test infrastructure) and its CRC-64 must equal the
literal.  Independently of that library the buffer's CRC-32 must equal tests/golden/golden_crc.json (written by
tests/golden/make_golden_crc.py only after the CPU oracle reproduced the same literal).  For every case the HIP
colour path (antialiasing + palette kernels, RenderCurrent's Color16 output) is pushed through the same encoder and must
give the same CRC-64."""
import json
import os

import numpy as np
import pytest

import _oracle
import golden_cases as gc
from fractalshark_amd import (GPURenderer, LAV2_FULL, PARITY_CPU, T_F64, T_HDR32, T_HDR64, _capi, inputs)

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_crc.json")))


def _pairs(co):
    return [(float(c["m"]), int(c["e"])) for c in co]


@pytest.fixture(scope="module")
def renderer(native_libs):
    assert GPURenderer.TestCudaIsWorking() != 0, "no usable HIP device: the product path has no CPU fallback"
    r = GPURenderer(0)
    yield r
    r.close()


def _hip_render(r, alg, v, ob, table, aa):
    """One golden case through the C ABI.  Returns (iteration buffer, Color16 buffer, reduction)."""
    w, h = gc.W * aa, gc.H * aa
    pal = _oracle.default_palette(8)  # Palette Default, depth 8, aux depth 0 (Fractal.cpp:536-538)
    assert r.InitializeMemory(w, h, aa, pal, len(pal), 0, 1, False) == 0
    assert r.ClearMemory() == 0
    n = v.num_iterations
    lib = r._lib
    if alg == "Cpu64":  # -> Gpu1x64 (mandel_1x_double)
        dx, dy, minx, maxy = v.coords_direct_f64(aa)
        assert r.Render(None, minx, maxy, dx, dy, n, T=T_F64) == 0
    elif alg in ("CpuHDR32", "CpuHDR64"):  # -> GpuHDRx32 / GpuHDRx64 direct
        is64 = alg == "CpuHDR64"
        dx, dy, minx, maxy = _pairs(v.coords_direct_hdr(is64))
        assert r.Render(None, minx, maxy, dx, dy, n, T=T_HDR64 if is64 else T_HDR32) == 0
    elif alg == "Cpu64PerturbedBLA":  # -> Gpu1x64PerturbedBLA
        assert lib.fs_upload_orbit(r._h, 0, T_F64, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
        assert lib.fs_upload_bla(r._h, T_F64, ob.level_ptrs, ob.level_sizes, ob.num_levels, ob.lm2) == 0
        co = ob.coords()
        assert lib.fs_render_bla(r._h, T_F64, co.ctypes.data, n) == 0
    elif "V2" in alg:  # -> GpuHDRx32/x64 Perturbed[RC]LAv2, CPU parity; RC = SimpleCompression upload
        assert ("RC" in alg) == bool(ob.compressed)
        assert r.InitializePerturb(1, ob, 0, None, table) == 0
        dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
        assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, n, T=T_HDR64 if ob.is64 else T_HDR32,
                                   Mode=LAV2_FULL, parity=PARITY_CPU) == 0
    else:  # Cpu32/64PerturbedBLAHDR -> GpuHDRx32/x64PerturbedBLA
        dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
        assert r.RenderPerturbBLA(None, ob, table, None, None, dx, dy, cx, cy, n) == 0
    it = r.new_iter_buffer()
    colors = np.zeros((gc.H, gc.W, 4), np.uint16)  # N_color_cu = 256 x 256 (already multiples of 16 x 8)
    red = _capi.Reduction()
    assert r.RenderCurrent(n, it, colors, red) == 0
    assert r.SyncComputeStream() == 0
    return it, colors, red


@pytest.mark.parametrize("name,view_n,alg,aa,crc64", gc.CASES, ids=[c[0] for c in gc.CASES])
def test_hip_render_reproduces_reference_golden_crc(renderer, native_libs, name, view_n, alg, aa, crc64):
    v, ob, table = gc.build_inputs(inputs, view_n, alg, aa)
    it, colors, red = _hip_render(renderer, alg, v, ob, table, aa)
    assert it.shape == tuple(GOLD[name]["iter_buffer_shape"])
    # (1) committed buffer CRC (made from the golden-pinned oracle) -- needs nothing but the HIP library
    assert gc.buffer_crc32(it) == GOLD[name]["iter_buffer_crc32"], name
    assert red.Sum == GOLD[name]["iter_sum"]
    # (2) the reference's literal, through the reference's own PNG writer
    assert _oracle.pin_lib() is not None, ("oracle/_ref/libpngpin.so did not travel to this box: run "
                                           "__graft_entry__.build() where /root/reference exists")
    assert _oracle.png_crc64(it, gc.W, gc.H, aa, v.num_iterations) == crc64, name
    # (3) the HIP colour path (antialiasing_kernel + palette, AntialiasingKernel.cuh:3-71) through the same writer
    assert _oracle.png_crc64_rgba16(colors, gc.W, gc.H) == crc64, name


# ---- the same RC golden cases with the orbit kept COMPRESSED in HBM and decompressed inside the kernel
RC_CASES = [c for c in gc.CASES if "RC" in c[2]]


@pytest.mark.parametrize("name,view_n,alg,aa,crc64", RC_CASES, ids=[c[0] + "-runtime-decompression" for c in RC_CASES])
def test_rc_goldens_with_in_kernel_decompression(renderer, native_libs, name, view_n, alg, aa, crc64):
    """PerturbExtras::SimpleCompression as the reference's GPU runs it (Perturb.cuh:146-326): only the waypoints are
    uploaded and every pixel walks the orbit with a sequential decompression cursor.  The frames must be the golden ones,
    and the resident orbit must be the waypoints only."""
    assert len(RC_CASES) == 2
    v, ob, table = gc.build_inputs(inputs, view_n, alg, aa)
    assert ob.compressed and ob.compressed_count < ob.count
    r = renderer
    try:
        assert r.set_compressed_orbit_mode(True) == 0
        it, colors, red = _hip_render(r, alg, v, ob, table, aa)
        resident = r.orbit_device_bytes
    finally:
        r.set_compressed_orbit_mode(False)
    assert resident == ob.compressed_count * (40 if ob.is64 else 24)
    assert gc.buffer_crc32(it) == GOLD[name]["iter_buffer_crc32"], name
    assert red.Sum == GOLD[name]["iter_sum"]
    assert _oracle.pin_lib() is not None
    assert _oracle.png_crc64(it, gc.W, gc.H, aa, v.num_iterations) == crc64, name
    assert _oracle.png_crc64_rgba16(colors, gc.W, gc.H) == crc64, name
    # the expanding mode on the same renderer afterwards: same frame, the whole orbit resident
    it2, _, _ = _hip_render(r, alg, v, ob, table, aa)
    assert np.array_equal(it, it2)
    assert r.orbit_device_bytes >= ob.count * (32 if ob.is64 else 16)


@pytest.mark.parametrize("is64", [False, True])
def test_in_kernel_decompression_modes_bands_and_refusals(renderer, native_libs, is64):
    """All LAv2 modes and both parities on a ragged frame, as the middle rank of a three-way row split, with the LA stages
    in use (they hand the perturbation loop an orbit index in the MIDDLE of the orbit: the cursor starts with a binary
    search); entry points that need the expanded orbit say so."""
    from fractalshark_amd import LAV2_LAO, LAV2_PO, PARITY_CPU_GPUSTAGE
    v = inputs.View.builtin(5, 70, 37)
    ob = inputs.Orbit(v, is64=is64, compression_exp=20)
    la = inputs.LATable(ob)
    T = T_HDR64 if is64 else T_HDR32
    co = _pairs(v.coords_perturb(ob))
    r = renderer
    frames = {}
    try:
        for seq in (False, True):
            assert r.set_compressed_orbit_mode(seq) == 0
            assert r.InitializeMemory(70, 37, 1, None, 0, 0, 0, False) == 0
            assert r.SetRowBands(8, 8, 24) == 0
            assert r.InitializePerturb(0, ob, 0, None, la) == 0
            for mode, parity in ((LAV2_FULL, PARITY_CPU), (LAV2_FULL, PARITY_CPU_GPUSTAGE), (LAV2_LAO, PARITY_CPU_GPUSTAGE),
                                 (LAV2_PO, PARITY_CPU_GPUSTAGE)):
                assert r.ClearMemory() == 0
                assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T, Mode=mode, parity=parity) == 0
                out = r.new_iter_buffer()
                assert r.RenderCurrent(v.num_iterations, out) == 0
                assert r.SyncComputeStream() == 0
                frames[(seq, mode, parity)] = out
            if seq:
                assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T, Mode=LAV2_PO, parity=PARITY_CPU) == 10100
                assert r._lib.fs_render_bla(r._h, T, v.coords_perturb(ob).ctypes.data, v.num_iterations) == 10100
                assert r.BuildLAOnDevice(ob, host_fallback=False) == 10100
                assert r.BuildBLAOnDevice(ob) == 10100
    finally:
        r.set_compressed_orbit_mode(False)
    for (seq, mode, parity), out in frames.items():
        if seq:
            assert np.array_equal(out, frames[(False, mode, parity)]), (mode, parity)
    full = _oracle.lav2_hdr32(v, ob, la, stage_test=1)
    got = frames[(True, LAV2_FULL, PARITY_CPU_GPUSTAGE)]
    for k, y in enumerate(range(8, 16)):  # the band this "rank" owns first
        assert np.array_equal(got[k, :70], full[y, :70])
