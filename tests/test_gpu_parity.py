"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs, against the
committed fixtures, and through size-independent properties at larger sizes.  Iteration counts must be
bit-identical (integer outputs: no tolerance)."""
import os

import numpy as np
import pytest

import _oracle
from fractalshark_amd import (GPURenderer, LAV2_FULL, LAV2_LAO, LAV2_PO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_F64,
                              T_HDR32, T_HDR64, _capi, inputs)

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))


def _pairs(co):
    return [(float(c["m"]), int(c["e"])) for c in co]


@pytest.fixture(scope="module")
def renderer(native_libs):
    assert GPURenderer.TestCudaIsWorking() != 0, "no usable HIP device: the product path has no CPU fallback"
    r = GPURenderer(0)
    yield r
    r.close()


@pytest.fixture(scope="module")
def v5_small(native_libs):
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v)
    return v, ob, inputs.LATable(ob), inputs.BLATable(ob)


def _render_lav2(r, v, ob, la, mode, parity, n_iter=None):
    w, h = v.width * v.antialiasing, v.height * v.antialiasing
    assert r.InitializeMemory(w, h, v.antialiasing, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
    assert r.ClearMemory() == 0
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    n = v.num_iterations if n_iter is None else n_iter
    T = T_HDR64 if ob.is64 else T_HDR32
    assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, n, T=T, Mode=mode, parity=parity) == 0
    assert r.SyncComputeStream() == 0
    out = r.new_iter_buffer()
    red = _capi.Reduction()
    assert r.RenderCurrent(n, out, None, red) == 0
    assert r.SyncComputeStream() == 0
    return out, red


def test_lav2_full_cpu_parity_fixture_and_oracle(renderer, v5_small):
    v, ob, la, _ = v5_small
    out, red = _render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU)
    assert np.array_equal(out, GOLD["view5_lav2_cpu_64x36"])
    assert np.array_equal(out, _oracle.lav2_hdr32(v, ob, la, stage_test=0))
    valid = out[:36, :64].astype(np.uint64)
    assert (red.Min, red.Max, red.Sum) == (int(valid.min()), int(valid.max()), int(valid.sum()))


def test_lav2_full_gpustage_parity(renderer, v5_small):
    v, ob, la, _ = v5_small
    out, _ = _render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU_GPUSTAGE)
    assert np.array_equal(out, GOLD["view5_lav2_gpustage_64x36"])


def test_lav2_lao_parity(renderer, v5_small):
    v, ob, la, _ = v5_small
    out, _ = _render_lav2(renderer, v, ob, la, LAV2_LAO, PARITY_CPU)
    assert np.array_equal(out, GOLD["view5_lao_cpu_64x36"])


def test_lav2_po_parity_is_bla_single_step_branch(renderer, v5_small):
    """C2: LAv2Mode::PO <-> single-step branch of CalcCpuPerturbationFractalBLA."""
    v, ob, la, _ = v5_small
    out, _ = _render_lav2(renderer, v, ob, la, LAV2_PO, PARITY_CPU)
    assert np.array_equal(out, GOLD["view5_po_64x36"])


def test_bla_parity(renderer, v5_small):
    v, ob, _, bla = v5_small
    r = renderer
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    assert r.RenderPerturbBLA(None, ob, bla, None, None, dx, dy, cx, cy, v.num_iterations) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    assert np.array_equal(out, GOLD["view5_bla_64x36"])


def test_direct_f64_parity(renderer, native_libs):
    v = inputs.View.builtin(0, 64, 48)
    r = renderer
    assert r.InitializeMemory(64, 48, 1, None, 0, 0, 0, False) == 0
    dx, dy, minx, maxy = v.coords_direct_f64()
    assert r.Render(None, minx, maxy, dx, dy, v.num_iterations, T=T_F64) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    assert np.array_equal(out, GOLD["view0_direct_f64_64x48"])


def test_ragged_size_and_padding(renderer, native_libs):
    """Width/height not multiples of 16/8: padding stays zero, valid region matches the oracle."""
    v = inputs.View.builtin(5, 37, 21)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    out, red = _render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU_GPUSTAGE)
    ref = _oracle.lav2_hdr32(v, ob, la, stage_test=1)
    assert out.shape == ref.shape == (24, 48)
    assert np.array_equal(out, ref)
    assert not out[21:, :].any() and not out[:, 37:].any()
    assert red.Sum == int(ref[:21, :37].astype(np.uint64).sum())


def test_row_bands_reassemble_full_frame(renderer, v5_small):
    """Multi-GPU tiling primitive: interleaved 8-row bands rendered separately equal the whole frame."""
    v, ob, la, _ = v5_small
    full, _ = _render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU_GPUSTAGE)
    r = renderer
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    world = 3
    got = np.zeros_like(full)
    for rank in range(world):
        assert r.SetRowBands(rank * 8, 8, world * 8) == 0
        assert r.ClearMemory() == 0
        assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL,
                                   parity=PARITY_CPU_GPUSTAGE) == 0
        loc = r.new_iter_buffer()
        assert r.RenderCurrent(v.num_iterations, loc) == 0
        assert r.SyncComputeStream() == 0
        k = 0
        for start in range(rank * 8, 36, world * 8):
            n = min(8, 36 - start)
            got[start:start + n] = loc[k:k + n]
            k += n
    assert r.SetRowBands(0, 0, 0) == 0
    assert np.array_equal(got[:36], full[:36])


def test_antialias_colors_match_integer_box_filter(renderer, native_libs):
    v = inputs.View.builtin(0, 32, 24, antialiasing=2)
    pal = _oracle.default_palette(8)
    r = renderer
    assert r.InitializeMemory(64, 48, 2, pal, len(pal), 0, 1, False) == 0
    dx, dy, minx, maxy = v.coords_direct_f64(2)
    assert r.Render(None, minx, maxy, dx, dy, v.num_iterations, T=T_F64) == 0
    it = r.new_iter_buffer()
    colors = np.zeros((32 * 24, 4), np.uint16)  # N_color_cu = 32 x 24 here (multiples of 16 x 8)
    assert r.RenderCurrent(v.num_iterations, it, colors) == 0
    assert r.SyncComputeStream() == 0
    assert np.array_equal(it, _oracle.direct_f64(v, aa=2))
    exp = np.zeros((24, 32, 4), np.uint64)
    for oy in range(24):
        for ox in range(32):
            acc = np.zeros(3, np.uint64)
            for ix in range(2 * ox, 2 * ox + 2):
                for iy in range(2 * oy, 2 * oy + 2):
                    n = int(it[iy, ix])
                    if n < v.num_iterations:
                        acc += pal[n % len(pal), :3].astype(np.uint64)
            exp[oy, ox, :3] = acc // 4
            exp[oy, ox, 3] = 65535
    assert np.array_equal(colors.reshape(24, 32, 4).astype(np.uint64), exp)


def test_errors_and_uninitialised_behaviour(native_libs):
    r = GPURenderer(0)
    # "memory not initialised" returns 0 silently (GPU_Render.cu:564-566,1007-1009)
    assert r.RenderCurrent(10, None) == 0
    assert r.InitializeMemory(64, 36, 5, None, 0, 0, 0, False) == 10002
    assert r.InitializeMemory(63, 36, 2, None, 0, 0, 0, False) == 10003
    assert r.InitializeMemory(64, 35, 2, None, 0, 0, 0, False) == 10004
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    # render without an uploaded orbit -> Error6 (GPU_Render.cu:1015-1022)
    assert r.RenderPerturbLAv2(None, None, None, (1, 0), (1, 0), (1, 0), (1, 0), 10) == 10005
    r.close()


def test_done_callback_fires(renderer, v5_small):
    import threading
    ev = threading.Event()
    assert renderer.EnqueueComputeDoneCallback(ev.set) == 0
    assert renderer.SyncComputeStream() == 0
    assert ev.wait(5.0)


def test_larger_frame_properties(renderer, native_libs):
    """320x180: whole-frame parity vs the oracle on a bounded set of rows + reduction identity + idempotence."""
    v = inputs.View.builtin(5, 320, 180)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    out, red = _render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU_GPUSTAGE)
    rows = (60, 68)
    ref = _oracle.lav2_hdr32(v, ob, la, stage_test=1, rows=rows)
    assert np.array_equal(out[rows[0]:rows[1]], ref[rows[0]:rows[1]])
    assert red.Sum == int(out[:180, :320].astype(np.uint64).sum())
    out2, _ = _render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU_GPUSTAGE)
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("parity", [PARITY_CPU, PARITY_CPU_GPUSTAGE])
def test_tuned_loop_equals_literal_transcription_1080p(renderer, native_libs, parity):
    """The tuned perturbation loop (speculative branch-free step + generic fallback) must give the same iteration
    buffer as the literal operation-by-operation kernel on a full 1920x1080 frame (~5e10 executed pixel-steps),
    and both must match the oracle on a sample of rows."""
    v = inputs.View.builtin(5, 1920, 1080)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    r = renderer
    try:
        assert r.set_kernel_variant(literal=True) == 0
        lit, red_l = _render_lav2(r, v, ob, la, LAV2_FULL, parity)
        assert r.set_kernel_variant(2) == 0  # exponent-tracking quiet runs only
        mid, _ = _render_lav2(r, v, ob, la, LAV2_FULL, parity)
        assert r.set_kernel_variant(literal=False) == 0  # scaled runs first (default)
        tun, red_t = _render_lav2(r, v, ob, la, LAV2_FULL, parity)
    finally:
        r.set_kernel_variant(literal=False)
    assert np.array_equal(lit, tun)
    assert np.array_equal(lit, mid)
    assert (red_l.Min, red_l.Max, red_l.Sum) == (red_t.Min, red_t.Max, red_t.Sum)
    _oracle.set_row_step(135)
    try:
        ref = _oracle.lav2_hdr32(v, ob, la, rows=(67, 1080), threads=16, stage_test=0 if parity == PARITY_CPU else 1)
    finally:
        _oracle.set_row_step(1)
    for y in range(67, 1080, 135):
        assert np.array_equal(tun[y], ref[y]), y


def test_literal_variant_small_fixture(renderer, v5_small):
    v, ob, la, _ = v5_small
    try:
        renderer.set_kernel_variant(literal=True)
        out, _ = _render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU)
    finally:
        renderer.set_kernel_variant(literal=False)
    assert np.array_equal(out, GOLD["view5_lav2_cpu_64x36"])


def test_view19_bla_parity_small(renderer, native_libs):
    """C5 arithmetic: View 19 (zoom 1e158, 412 729-entry orbit, 113 M iterations, 20-level BLA table) at 64x36."""
    v = inputs.View.builtin(19, 64, 36)
    ob = inputs.Orbit(v)
    bla = inputs.BLATable(ob)
    r = renderer
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    assert r.RenderPerturbBLA(None, ob, bla, None, None, dx, dy, cx, cy, v.num_iterations) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    assert np.array_equal(out, _oracle.bla_hdr32(v, ob, bla))


def test_view5_po_parity_rows_1080p(renderer, native_libs):
    """C2 at its BASELINE size (1920x1080): PO entry point vs the BLA function's single-step branch on sample rows."""
    v = inputs.View.builtin(5, 1920, 1080)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    out, red = _render_lav2(renderer, v, ob, la, LAV2_PO, PARITY_CPU)
    _oracle.set_row_step(270)
    try:
        ref = _oracle.bla_hdr32(v, ob, None, rows=(135, 1080), threads=16)
    finally:
        _oracle.set_row_step(1)
    for y in range(135, 1080, 270):
        assert np.array_equal(out[y], ref[y]), y
    assert red.Sum == int(out[:1080, :1920].astype(np.uint64).sum())


# ---- HDRFloat<double> family (GpuHDRx64* <-> Cpu64Perturbed*HDR) and HDR direct kernels (GpuHDRx32 <-> CpuHDR32/64)
@pytest.fixture(scope="module")
def v5_small_64(native_libs):
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v, is64=True)
    return v, ob, inputs.LATable(ob), inputs.BLATable(ob)


@pytest.mark.parametrize("mode,parity,st,omode", [(LAV2_FULL, PARITY_CPU, 0, 0), (LAV2_FULL, PARITY_CPU_GPUSTAGE, 1, 0),
                                                  (LAV2_LAO, PARITY_CPU, 0, 2)])
def test_hdr64_lav2_parity(renderer, v5_small_64, mode, parity, st, omode):
    v, ob, la, _ = v5_small_64
    out, red = _render_lav2(renderer, v, ob, la, mode, parity)
    ref = _oracle.lav2_hdr32(v, ob, la, stage_test=st, mode=omode)
    assert np.array_equal(out, ref)
    assert red.Sum == int(ref[:36, :64].astype(np.uint64).sum())


def test_hdr64_po_and_bla_parity(renderer, v5_small_64):
    v, ob, la, bla = v5_small_64
    out, _ = _render_lav2(renderer, v, ob, la, LAV2_PO, PARITY_CPU)
    assert np.array_equal(out, _oracle.bla_hdr32(v, ob, None))
    r = renderer
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    assert r.RenderPerturbBLA(None, ob, bla, None, None, dx, dy, cx, cy, v.num_iterations) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    assert np.array_equal(out, _oracle.bla_hdr32(v, ob, bla))


@pytest.mark.parametrize("is64", [False, True])
def test_direct_hdr_parity(renderer, native_libs, is64):
    v = inputs.View.builtin(0, 64, 48)
    r = renderer
    assert r.InitializeMemory(64, 48, 1, None, 0, 0, 0, False) == 0
    dx, dy, minx, maxy = _pairs(v.coords_direct_hdr(is64))
    assert r.Render(None, minx, maxy, dx, dy, v.num_iterations, T=T_HDR64 if is64 else T_HDR32) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    assert np.array_equal(out, _oracle.direct_hdr(v, is64))


def test_simple_compression_orbit_parity_hdr64(renderer, native_libs):
    """GpuHDRx64PerturbedRCLAv2 <-> Cpu64PerturbedRCBLAV2HDR (golden CRC 68df9ceecaf1a667 pins the CPU chain)."""
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v, is64=True, compression_exp=20)
    la = inputs.LATable(ob)
    out, _ = _render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU)
    assert np.array_equal(out, _oracle.lav2_hdr32(v, ob, la, stage_test=0))


def test_simple_compression_orbit_parity(renderer, native_libs):
    """GpuHDRx32PerturbedRCLAv2 <-> Cpu32PerturbedRCBLAV2HDR: the waypoints are expanded on the device; the frame must
    equal the CPU function reading the orbit through RuntimeDecompressor (golden CRC b956600cfdfe431a pins that chain)."""
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v, compression_exp=20)
    assert ob.compressed and ob.compressed_count < ob.count
    la = inputs.LATable(ob)
    out, _ = _render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU)
    assert np.array_equal(out, _oracle.lav2_hdr32(v, ob, la, stage_test=0))
    # and it is NOT the uncompressed frame (the decompressed orbit differs in the last bits)
    ob_u = inputs.Orbit(v)
    la_u = inputs.LATable(ob_u)
    out_u, _ = _render_lav2(renderer, v, ob_u, la_u, LAV2_FULL, PARITY_CPU)
    assert np.array_equal(out_u, GOLD["view5_lav2_cpu_64x36"])


def test_plain_double_bla_parity(renderer, native_libs):
    """Gpu1x64PerturbedBLA <-> Cpu64PerturbedBLA (golden CRC f201db00ade569fc pins the CPU chain)."""
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.OrbitF64(v)
    r = renderer
    lib = r._lib
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert lib.fs_upload_orbit(r._h, 0, T_F64, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
    co = ob.coords()
    for use_bla in (True, False):
        if use_bla:
            assert lib.fs_upload_bla(r._h, T_F64, ob.level_ptrs, ob.level_sizes, ob.num_levels, ob.lm2) == 0
        else:
            assert lib.fs_upload_bla(r._h, T_F64, None, None, 0, 0) == 0
        assert lib.fs_render_bla(r._h, T_F64, co.ctypes.data, v.num_iterations) == 0
        out = r.new_iter_buffer()
        assert r.RenderCurrent(v.num_iterations, out) == 0
        assert r.SyncComputeStream() == 0
        assert np.array_equal(out, _oracle.bla_f64(v, ob, use_bla=use_bla))


# ---- HDRFloat<CudaDblflt> ("2x32 + exponent", GpuHDRx2x32PerturbedLAv2*): no CPU twin in the reference; the checker
# restates the CUDA kernel (oracle/gpu_ref_2x32.cpp, parity unpinned -- see its header and tests/test_2x32_oracle.py)
def _render_2x32(r, v, o2, la2, mode, n_iter=None, bands=None):
    from fractalshark_amd import T_HDR2X32
    w, h = v.width * v.antialiasing, v.height * v.antialiasing
    assert r.InitializeMemory(w, h, v.antialiasing, None, 0, 0, 0, False) == 0
    if bands:
        assert r.SetRowBands(*bands) == 0
    assert r.InitializePerturb(0, o2, 0, None, la2) == 0
    assert r.ClearMemory() == 0
    co = v.coords_perturb_2x32(o2)
    tr = [(float(c["head"]), float(c["tail"]), int(c["e"])) for c in co]
    n = v.num_iterations if n_iter is None else n_iter
    assert r.RenderPerturbLAv2(None, None, None, tr[0], tr[1], tr[2], tr[3], n, T=T_HDR2X32, Mode=mode) == 0
    assert r.SyncComputeStream() == 0
    out = r.new_iter_buffer()
    red = _capi.Reduction()
    assert r.RenderCurrent(n, out, None, red) == 0
    assert r.SyncComputeStream() == 0
    return out, red


@pytest.fixture(scope="module")
def v5_2x32(native_libs):
    v = inputs.View.builtin(5, 64, 36)
    o = inputs.Orbit(v, is64=True)
    la = inputs.LATable(o, use_small_exponents=True)
    return v, inputs.Orbit2x32(o), inputs.LATable2x32(la)


@pytest.mark.parametrize("mode,omode", [(LAV2_FULL, 0), (LAV2_LAO, 2)])
def test_2x32_lav2_matches_restated_cuda_kernel(renderer, v5_2x32, mode, omode):
    v, o2, la2 = v5_2x32
    out, red = _render_2x32(renderer, v, o2, la2, mode)
    ref = _oracle.gpu_lav2_2x32(v, o2, la2, mode=omode)
    assert np.array_equal(out, ref)
    assert red.Sum == int(ref[:36, :64].astype(np.uint64).sum())


def test_2x32_perturbation_only_rows(renderer, v5_2x32):
    v, o2, la2 = v5_2x32
    # rows 8..11 only: perturbation-only runs ~8e4 double-float steps per pixel
    out, _ = _render_2x32(renderer, v, o2, None, LAV2_PO, bands=(8, 4, 36))
    ref = _oracle.gpu_lav2_2x32(v, o2, None, mode=1, rows=(8, 12))
    assert np.array_equal(out[:4, :64], ref[8:12, :64])
    assert renderer.SetRowBands(0, 0, 0) == 0


def test_2x32_view14_deep_zoom_rows(renderer, native_libs):
    """BASELINE config C4's view (2^-21645) at a small size: 116 695-entry orbit, AT-dominated."""
    v = inputs.View.builtin(14, 64, 36, antialiasing=1)
    o = inputs.Orbit(v, is64=True)
    la = inputs.LATable(o, use_small_exponents=True)
    o2, la2 = inputs.Orbit2x32(o), inputs.LATable2x32(la)
    out, _ = _render_2x32(renderer, v, o2, la2, LAV2_FULL)
    ref = _oracle.gpu_lav2_2x32(v, o2, la2, mode=0)
    assert np.array_equal(out, ref)
    assert len(np.unique(out[:36, :64])) > 16


# ---- IterType = uint64_t: 64-bit iteration buffer / record layouts over the 32-bit device counters
def test_uint64_itertype_matches_uint32(renderer, v5_small):
    v, ob, la, _ = v5_small
    r = renderer
    pal = (np.arange(64 * 4, dtype=np.uint32).reshape(64, 4) * 257 % 65536).astype(np.uint16)
    n = v.num_iterations
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    results = {}
    for ib in (4, 8):
        assert r.InitializeMemory(64, 36, 1, pal, 64, 0, 7 + ib, False, iter_bytes=ib) == 0
        la_in = inputs.LATableU64(la) if ib == 8 else la
        assert r.InitializePerturb(0, ob, 0, None, la_in, iter_bytes=ib) == 0
        assert r.ClearMemory() == 0
        assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, n, Mode=LAV2_FULL, parity=PARITY_CPU_GPUSTAGE) == 0
        out = r.new_iter_buffer()
        colors = np.zeros((64 * 40, 4), np.uint16)
        red = _capi.Reduction()
        assert r.RenderCurrent(n, out, colors, red) == 0
        assert r.SyncComputeStream() == 0
        results[ib] = (out, colors, (red.Min, red.Max, red.Sum))
    assert results[8][0].dtype == np.uint64 and results[8][0].itemsize == 8
    assert np.array_equal(results[8][0], results[4][0].astype(np.uint64))
    assert np.array_equal(results[8][1], results[4][1])
    assert results[8][2] == results[4][2]
    # counts that do not fit the 32-bit device counters are refused, not truncated
    big = inputs.LATableU64(la)
    big._stages[0, 1] = 1 << 33
    assert r.InitializePerturb(0, ob, 0, None, big, iter_bytes=8) != 0  # (a stage cannot hold more records than the table)
    # an iteration cap of 2^32 needs IterType = uint64_t: a caller error (hipErrorInvalidValue) with a 4-byte buffer (see the
    # next test and tests/test_gpu_wide_counters.py for 8 bytes)
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=4) == 0
    assert r.InitializePerturb(0, ob, 0, None, la) == 0
    assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, 1 << 32, Mode=LAV2_FULL) == 1


@pytest.mark.parametrize("is64", [False, True])
def test_uint64_itertype_counts_past_2_to_32(renderer, native_libs, is64):
    """GPURenderer::RenderPerturbLAv2<uint64_t, ...> with an iteration cap above 2^32 (LAKernel.cuh:3 is templated on
    IterType; GPU_Render.cu:1204-1300 instantiates uint64_t): 64-bit counters in the kernel.  Interior pixels must come
    back with exactly the cap, which no 32-bit counter can hold; everything is compared with the CPU function
    instantiated for uint64_t (oracle)."""
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v, is64=is64)
    la = inputs.LATable(ob)
    n = (1 << 32) + 12345
    r = renderer
    T = T_HDR64 if is64 else T_HDR32
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=8) == 0
    assert r.InitializePerturb(0, ob, 0, None, inputs.LATableU64(la), iter_bytes=8) == 0
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    for mode, omode in ((LAV2_FULL, 0), (LAV2_LAO, 2)):
        assert r.ClearMemory() == 0
        assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, n, T=T, Mode=mode, parity=PARITY_CPU_GPUSTAGE) == 0
        out = r.new_iter_buffer()
        red = _capi.Reduction()
        assert r.RenderCurrent(n, out, None, red) == 0
        assert r.SyncComputeStream() == 0
        ref = _oracle.lav2_u64(v, ob, la, n, stage_test=1, mode=omode)
        assert out.dtype == np.uint64 and np.array_equal(out, ref)
        if mode == LAV2_FULL:
            assert int(out.max()) == n and int((out[:36, :64] == n).sum()) >= 1  # interior pixels sit at the cap
        assert red.Max == int(ref[:36, :64].max()) and red.Sum == int(ref[:36, :64].sum())
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=4) == 0


# ---- scaled perturbation (GpuHDRx32PerturbedScaled): no CPU twin; checker = restated CUDA kernel, parity unpinned
def test_scaled_hdr32_matches_restated_cuda_kernel(renderer, v5_small):
    v, ob, _, _ = v5_small
    r = renderer
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert r.ClearMemory() == 0
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    r.enable_step_count(True)
    assert r.RenderPerturbBLAScaled(None, ob, ob, None, None, dx, dy, cx, cy, v.num_iterations) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    st = r.read_step_count()
    r.enable_step_count(False)
    ref, rst = _oracle.gpu_scaled_hdr32(v, ob, stats=True)
    assert np.array_equal(out, ref)
    # same work: rescales / full-precision steps / binary32 steps
    assert (st["at_iterations"], st["la_steps"], st["perturb_steps"]) == (rst["rescales"], rst["full_steps"],
                                                                         rst["float_steps"])
    # and it is the same picture as the HDRFloat<float> perturbation path up to binary32 glitches
    hdr = _oracle.bla_hdr32(v, ob, None)
    d = np.abs(out[:36, :64].astype(np.int64) - hdr[:36, :64].astype(np.int64))
    assert (d <= 2).mean() > 0.3 and np.median(d) <= 4


@pytest.mark.parametrize("w,h,aa", [(64, 36, 1), (16, 12, 4)])
def test_scaled_hdr32_view14_deep_zoom(renderer, native_libs, w, h, aa):
    """BASELINE config C4's other form: View 14 (2^-21645) through GpuHDRx32PerturbedScaled (fs_render_scaled), with
    the PerturbExtras::Bad orbit pair of that view (116 695 entries, `bad` entries present) -- at AA 1 and in C4's AA 4
    geometry.  The iteration cap is lowered to 1.8 M: with the view's own cap (2^31 - 2) perturbation-only rendering
    needs 2.1e9 steps for almost every pixel of this view (the first escapes are at 1.53 M), on any hardware."""
    v = inputs.View.builtin(14, w, h, antialiasing=aa)
    ob = inputs.Orbit(v)
    n = 1800000
    r = renderer
    assert r.InitializeMemory(w * aa, h * aa, aa, None, 0, 0, 0, False) == 0
    assert r.ClearMemory() == 0
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob, aa))
    r.enable_step_count(True)
    assert r.RenderPerturbBLAScaled(None, ob, ob, None, None, dx, dy, cx, cy, n) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(n, out) == 0
    assert r.SyncComputeStream() == 0
    st = r.read_step_count()
    r.enable_step_count(False)
    ref, rst = _oracle.gpu_scaled_hdr32(v, ob, aa=aa, stats=True, n_iterations=n)
    assert np.array_equal(out, ref)
    assert rst["rescales"] > 0 and rst["full_steps"] > 0  # the rescale and the `bad`-entry paths are both exercised
    assert (st["at_iterations"], st["la_steps"], st["perturb_steps"]) == (rst["rescales"], rst["full_steps"],
                                                                         rst["float_steps"])


@pytest.mark.parametrize("view_n,w,h,n", [(5, 320, 180, None), (14, 96, 54, 1800000), (3, 128, 72, None)])
def test_scaled_hdr32_tuned_equals_literal(renderer, native_libs, view_n, w, h, n):
    """The tuned scaled kernel (runs of binary32 steps whose outcome tests are implied by a per-entry bound) against the
    statement-for-statement kernel (FS_VARIANT_LITERAL) on larger frames than the oracle is run on: same frame, same
    rescale / full-precision / binary32 step counts."""
    v = inputs.View.builtin(view_n, w, h, antialiasing=1)
    ob = inputs.Orbit(v)
    n = v.num_iterations if n is None else n
    r = renderer
    assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    outs, stats = [], []
    try:
        for variant in (1, 0):  # literal, tuned
            assert r.set_kernel_variant(variant) == 0
            assert r.ClearMemory() == 0
            r.enable_step_count(True)
            assert r.RenderPerturbBLAScaled(None, ob, ob, None, None, dx, dy, cx, cy, n) == 0
            out = r.new_iter_buffer()
            assert r.RenderCurrent(n, out) == 0
            assert r.SyncComputeStream() == 0
            st = r.read_step_count()
            r.enable_step_count(False)
            outs.append(out)
            stats.append((st["at_iterations"], st["la_steps"], st["perturb_steps"]))
            # probes of the tuned kernel (tools/scaled_kernel_probe.py): the binary32 steps taken INSIDE wave-voted runs are
            # a subset of the binary32 steps, and runs exist only where steps do (each probe is counted once)
            assert st["scaled_steps"] <= st["perturb_steps"], (variant, st)
            assert st["scaled_runs"] <= st["scaled_steps"] or st["scaled_steps"] == 0
            if variant == 1:
                assert st["scaled_steps"] == 0 and st["scaled_runs"] == 0  # the literal kernel has no runs
    finally:
        r.set_kernel_variant(0)
    assert np.array_equal(outs[0], outs[1])
    assert stats[0] == stats[1]
    assert stats[0][2] > 0


# ---- LA table built on the device (fs_build_la) == the golden-pinned host builder in its
# single-threaded form (LAReference.cpp:28-210,774-966,1050-1074), bit for bit: every record of every stage, the stage
# table, ATInfo and the UseAT decision -- and the frame rendered from the device-built table equals the fixture
@pytest.mark.parametrize("view_n,is64,small", [(5, False, False), (5, True, False), (1, False, False), (3, True, False),
                                                (3, False, False), (4, False, False), (9, False, False), (9, True, False),
                                                (11, True, False), (11, False, False), (14, True, True),
                                                (14, False, False), (19, False, False), (19, True, False),
                                                (6, False, False)])
def test_la_table_built_on_device_equals_host_builder(renderer, native_libs, view_n, is64, small):
    import ctypes as C
    v = inputs.View.builtin(view_n, 64, 36, antialiasing=1)
    ob = inputs.Orbit(v, is64=is64)
    la = inputs.LATable(ob, host_threads=1, use_small_exponents=small)
    r = renderer
    T = T_HDR64 if is64 else T_HDR32
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert r._lib.fs_upload_orbit(r._h, 0, T, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
    assert r.BuildLAOnDevice(ob, use_small_exponents=small) == 0
    las, stages, at, use_at, is_valid = r.read_la(is64)
    assert is_valid and la.is_valid
    assert stages.shape[0] == la.stage_count and np.array_equal(stages, la.stages())
    assert las.shape[0] == la.count
    host = la.records()
    host = host.view(np.uint8).reshape(la.count, -1)
    bad = np.nonzero((las != host).any(axis=1))[0]
    assert len(bad) == 0, "first differing record %d of %d" % (int(bad[0]), la.count)
    assert use_at == la.use_at
    assert at == bytes((C.c_char * C.sizeof(la.at)).from_address(C.addressof(la.at)))


@pytest.mark.parametrize("is64", [False, True])
def test_la_table_built_on_device_from_a_compressed_orbit(renderer, native_libs, is64):
    """PerturbExtras::SimpleCompression: the orbit is expanded on the device at upload and the table built from it uses
    periodDivisor 8 (LAReference.cpp:12-19) -- must equal the host builder's table for the same compressed orbit, and the
    frame rendered from it the oracle's (golden-pinned chain b956... / 68df...)."""
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v, is64=is64, compression_exp=20)
    assert ob.compressed
    la = inputs.LATable(ob, host_threads=1)
    r = renderer
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, ob, 0, None, None) == 0  # waypoints only, no table
    assert r.BuildLAOnDevice(ob) == 0
    las, stages, at, use_at, is_valid = r.read_la(is64)
    assert np.array_equal(stages, la.stages()) and las.shape[0] == la.count
    assert las.tobytes() == la.records().tobytes() and use_at == la.use_at
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    T = T_HDR64 if is64 else T_HDR32
    assert r.ClearMemory() == 0
    assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, v.num_iterations, T=T, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    assert np.array_equal(out, _oracle.lav2_hdr32(v, ob, la, stage_test=0))


# ---- ... and in its MULTI-THREADED form (CreateLAFromOrbitMT, LAReference.cpp:215-770: what FractalShark's CPU builder makes of
# an orbit of 100 000 entries and more on a host with two or more hardware threads): fs_build_la_mt(host_threads) == the host
# builder's replay of that variant with the same thread count, bit for bit.  On the built-in views the pieces join cleanly and
# the table equals the single-threaded one; the crafted orbits put two adjacent deep minima right behind a worker's first index
# (Orbit.scale_entries), where the worker's first record and the scan arriving from the left disagree and the tables differ.
def _check_la_mt(r, ob, is64, threads, expect_differs, compressed=False):
    import ctypes as C
    la = inputs.LATable(ob, host_threads=threads)
    st = inputs.LATable(ob, host_threads=1)
    differs = la.count != st.count or la.records().tobytes() != st.records().tobytes()
    assert differs == expect_differs
    T = T_HDR64 if is64 else T_HDR32
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    if compressed:
        assert r.InitializePerturb(1, ob, 0, None, None) == 0
    else:
        assert r._lib.fs_upload_orbit(r._h, 0, T, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
    assert r.BuildLAOnDevice(ob, host_threads=threads, host_fallback=False) == 0
    las, stages, at, use_at, is_valid = r.read_la(is64)
    assert is_valid and la.is_valid
    assert stages.shape[0] == la.stage_count and np.array_equal(stages, la.stages())
    assert las.shape[0] == la.count
    host = la.records().view(np.uint8).reshape(la.count, -1)
    bad = np.nonzero((las != host).any(axis=1))[0]
    assert len(bad) == 0, "first differing record %d of %d" % (int(bad[0]), la.count)
    assert use_at == la.use_at
    assert at == bytes((C.c_char * C.sizeof(la.at)).from_address(C.addressof(la.at)))
    # host_threads = 1 through the same entry point is the single-threaded table
    assert r.BuildLAOnDevice(ob, host_threads=1, host_fallback=False) == 0
    las1 = r.read_la(is64)[0]
    assert las1.shape[0] == st.count and las1.tobytes() == st.records().tobytes()


@pytest.mark.parametrize("view_n,is64,threads,compression", [
    (14, False, 2, None),                                # 116 695 entries: two pieces whatever the host has
    (19, False, 2, None), (19, False, 3, None), (19, False, 8, None), (19, True, 5, None), (19, True, 8, None),
    (19, False, 64, None),                               # 412 729 entries: at most 8 pieces
    (6, False, 4, None), (6, False, 16, None), (6, True, 9, None),  # 457 977 entries: at most 9
    (19, False, 8, 20),                                  # SimpleCompression: expanded at upload, periodDivisor 8
])
def test_la_table_built_on_device_equals_host_builder_multithreaded(renderer, native_libs, view_n, is64, threads, compression):
    v = inputs.View.builtin(view_n, 64, 36, antialiasing=1)
    ob = inputs.Orbit(v, is64=is64) if compression is None else inputs.Orbit(v, is64=is64, compression_exp=compression)
    _check_la_mt(renderer, ob, is64, threads, expect_differs=False, compressed=compression is not None)


@pytest.mark.parametrize("is64,threads,edits", [
    # (worker k, offset behind its first index, 2^e of that entry, 2^e of the next one)
    (False, 3, [(2, 9, -30, -50)]),
    (False, 8, [(1, 10, -30, -50)]),
    (False, 8, [(1, 10, -20, -24)]),
    (False, 8, [(3, 4, -30, -50)]),
    (False, 8, [(6, 7, -20, -24)]),
    (False, 8, [(1, 10, -30, -50), (3, 4, -20, -24), (6, 7, -30, -50)]),   # three joins off in one table
    (True, 8, [(1, 10, -30, -50), (3, 4, -20, -24), (6, 7, -30, -50)]),
])
def test_la_multithreaded_table_where_it_differs_from_the_single_threaded_one(renderer, native_libs, is64, threads, edits):
    v = inputs.View.builtin(19, 64, 36, antialiasing=1)
    ob = inputs.Orbit(v, is64=is64)
    max_ref = ob.count - 1
    idx, ex = [], []
    for k, a, ea, eb in edits:
        begin = max_ref * k // threads
        idx += [begin + a, begin + a + 1]
        ex += [ea, eb]
    assert ob.scale_entries(idx, ex) == len(idx)
    _check_la_mt(renderer, ob, is64, threads, expect_differs=True)


@pytest.mark.parametrize("seed,count,threads", [(5, 20000, 8), (2, 20000, 3), (1, 2000, 5)])
def test_la_multithreaded_table_of_an_orbit_with_random_minima(renderer, native_libs, seed, count, threads):
    """Thousands of artificial period boundaries (half of them in adjacent pairs): long chains, many stages."""
    v = inputs.View.builtin(19, 64, 36, antialiasing=1)
    ob = inputs.Orbit(v, is64=False)
    rng = np.random.default_rng(seed)
    idx = rng.integers(2, ob.count - 2, count).astype(np.uint64)
    idx = np.concatenate([idx, idx[:count // 2] + 1])
    ex = -rng.integers(5, 60, idx.size).astype(np.int32)
    ob.scale_entries(idx, ex)
    la = inputs.LATable(ob, host_threads=threads)
    st = inputs.LATable(ob, host_threads=1)
    _check_la_mt(renderer, ob, False, threads, expect_differs=la.records().tobytes() != st.records().tobytes())


def _tiny_view(cx, cy, w, n, W=64, H=36):
    from decimal import Decimal, getcontext
    getcontext().prec = 50
    cx, cy, w = Decimal(cx), Decimal(cy), Decimal(w)
    h = w * H / W
    return inputs.View(str(cx - w / 2), str(cy - h / 2), str(cx + w / 2), str(cy + h / 2), W, H, num_iterations=n)


@pytest.mark.parametrize("name,args,is64", [
    ("view2", None, False),                                                         # 59 entries, no period: two records, not valid
    ("period3_bulb_50", ("-0.1225", "0.7448", "1e-4", 50), False),                  # 51 entries, periods found: a valid 2-stage table
    ("period3_bulb_50", ("-0.1225", "0.7448", "1e-4", 50), True),
    ("period3_centre", ("-0.122561166876654", "0.744861766619744", "1e-6", 60), False),  # 4 entries
    ("period2_centre", ("-1.0", "0.0", "1e-5", 40), False),                         # 3 entries: the smallest orbit with a step to fold
    ("main_cardioid_30", ("-0.2", "0.1", "1e-5", 30), True),                        # 31 entries, no period
    ("seahorse_63", ("-0.75", "0.1", "1e-4", 63), False),                           # 36 entries (the orbit escapes)
])
def test_la_build_on_device_for_orbits_of_at_most_64_entries(renderer, native_libs, name, args, is64):
    """LowBound = 64 (LAReference.h:56): an orbit this short either yields periods (a normal, small table) or CreateLAFromOrbit
    keeps one record over the whole orbit plus the closing one and returns false -- the table is then NOT valid
    (LAReference.cpp:135-140, :1002-1005) and the kernels ignore it.  Both outcomes on the device == the host builder's:
    records, stage table, validity, UseAT -- and the frame rendered with the device's table == the oracle's with the host's."""
    v = inputs.View.builtin(2, 64, 36, antialiasing=1) if args is None else _tiny_view(*args)
    ob = inputs.Orbit(v, is64=is64)
    assert 3 <= ob.count <= 65
    la = inputs.LATable(ob, host_threads=1)
    r = renderer
    T = T_HDR64 if is64 else T_HDR32
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert r._lib.fs_upload_orbit(r._h, 0, T, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
    assert r.BuildLAOnDevice(ob, host_fallback=False) == 0
    las, stages, at, use_at, is_valid = r.read_la(is64)
    assert is_valid == la.is_valid and use_at == la.use_at
    assert stages.shape[0] == la.stage_count and np.array_equal(stages, la.stages())
    assert las.shape[0] == la.count and las.tobytes() == la.records().tobytes()
    if la.is_valid:
        import ctypes as C
        assert at == bytes((C.c_char * C.sizeof(la.at)).from_address(C.addressof(la.at)))
    assert r.ClearMemory() == 0
    assert r.RenderPerturbLAv2(None, None, None, *_pairs(v.coords_perturb(ob)), v.num_iterations, T=T,
                               Mode=LAV2_FULL, parity=PARITY_CPU) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    assert np.array_equal(out, _oracle.lav2_hdr32(v, ob, la, stage_test=0))  # (dispatches on the orbit's type)


def test_frame_from_device_built_la_table(renderer, v5_small):
    """View 5 64x36 rendered from the device-built table == the golden fixture (host table, golden-pinned oracle)."""
    v, ob, la, _ = v5_small
    r = renderer
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert r._lib.fs_upload_orbit(r._h, 0, T_HDR32, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
    assert r.BuildLAOnDevice(ob) == 0
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    for parity, key in ((PARITY_CPU, "view5_lav2_cpu_64x36"), (PARITY_CPU_GPUSTAGE, "view5_lav2_gpustage_64x36")):
        assert r.ClearMemory() == 0
        assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL, parity=parity) == 0
        out = r.new_iter_buffer()
        assert r.RenderCurrent(v.num_iterations, out) == 0
        assert r.SyncComputeStream() == 0
        assert np.array_equal(out, GOLD[key])


# ---- BLA table built on the device (BLAS::Init) == the golden-pinned host builder, bit for bit
@pytest.mark.parametrize("view_n,is64", [(5, False), (19, False), (5, True)])
def test_bla_table_built_on_device_equals_host_builder(renderer, native_libs, view_n, is64):
    v = inputs.View.builtin(view_n, 64, 36, antialiasing=1)
    ob = inputs.Orbit(v, is64=is64)
    host = inputs.BLATable(ob)
    r = renderer
    T = T_HDR64 if is64 else T_HDR32
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert r._lib.fs_upload_orbit(r._h, 0, T, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
    assert r.BuildBLAOnDevice(ob) == 0
    assert r._lib.fs_bla_num_levels(r._h) == host.num_levels and r._lib.fs_bla_lm2(r._h) == host.lm2
    dev = r.read_bla_levels(is64)
    assert [len(a) for a in dev] == host.sizes()
    for l, a in enumerate(dev):
        assert np.array_equal(a, host.level(l)), "level %d differs" % l
    # and a render with the device-built table is the oracle's render
    co = v.coords_perturb(ob)
    assert r.ClearMemory() == 0
    assert r._lib.fs_render_bla(r._h, T, co.ctypes.data, v.num_iterations) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    assert np.array_equal(out, _oracle.bla_hdr32(v, ob, host))


# ---- threading contract: four renderers on one device driven from four threads, and a progressive
# RenderCurrent on the display stream while an iteration kernel is in flight on the compute stream
def test_four_renderers_concurrently_and_progressive_readback(native_libs, v5_small):
    import threading
    v, ob, la, _ = v5_small
    ref = _oracle.lav2_hdr32(v, ob, la, stage_test=0)
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    results, errors = [None] * 4, []

    def worker(i):
        try:
            r = GPURenderer(0)
            assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
            assert r.InitializePerturb(i + 1, ob, 0, None, la) == 0
            for _ in range(3):
                assert r.ClearMemory() == 0
                assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL,
                                           parity=PARITY_CPU) == 0
                # progressive read-back on the display stream while the kernel may still be running: must not fail and
                # must return either untouched (0) or final values, never garbage
                part = r.new_iter_buffer()
                assert r.RenderCurrent(v.num_iterations, part, None, None, progressive=True) == 0
                assert r.SyncDisplayStream() == 0
                assert np.all((part == 0) | (part == ref))
                assert r.SyncComputeStream() == 0
            out = r.new_iter_buffer()
            assert r.RenderCurrent(v.num_iterations, out) == 0
            assert r.SyncComputeStream() == 0
            results[i] = out
            r.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for out in results:
        assert np.array_equal(out, ref)


# ---- Gpu1x32 / Gpu2x32 / Gpu2x64 direct kernels (no CPU twin; checker = restated CUDA kernels, oracle/gpu_ref_lp.cpp)
@pytest.mark.parametrize("kind,ip", [("1x32", 1), ("1x32", 4), ("1x32", 16), ("2x32", 1), ("2x32", 8), ("2x64", 1),
                                     ("4x32", 1), ("4x64", 1)])
def test_low_precision_direct_kernels(renderer, native_libs, kind, ip):
    from fractalshark_amd import T_2X32, T_2X64, T_4X32, T_4X64, T_F32
    v = inputs.View.builtin(0, 70, 37)  # ragged size: padding columns / rows stay zero
    r = renderer
    assert r.InitializeMemory(70, 37, 1, None, 0, 0, 0, False) == 0
    assert r.ClearMemory() == 0
    T = {"1x32": T_F32, "2x32": T_2X32, "2x64": T_2X64, "4x32": T_4X32, "4x64": T_4X64}[kind]
    assert r.RenderLowPrecision(None, v.coords_direct_lp(kind), v.num_iterations, ip, T=T) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    ref = _oracle.gpu_direct_lp(v, kind, ip)
    assert np.array_equal(out, ref)
    # same picture as the pinned Cpu64 render: the GPU kernels sample row R at maxY - dy*(R+1) and start from z = 0
    # (one extra iteration, rounded up to a multiple of iteration_precision)
    cpu = _oracle.direct_f64(v)[:37, :70].astype(np.int64)
    a = out[:36, :70].astype(np.int64)
    d = a - np.minimum(cpu[1:37] + 1, v.num_iterations)
    assert ((d >= 0) & (d < ip + 1)).mean() > 0.97, np.unique(d, return_counts=True)
    if kind in ("1x32", "2x32"):  # iteration_precision values the reference does not instantiate launch nothing
        assert r.ClearMemory() == 0
        assert r.RenderLowPrecision(None, v.coords_direct_lp(kind), v.num_iterations, 3, T=T) == 0
        z = r.new_iter_buffer()
        assert r.RenderCurrent(v.num_iterations, z) == 0
        assert r.SyncComputeStream() == 0
        assert not z.any()


def test_scaled_f64_matches_restated_cuda_kernel(renderer, native_libs):
    """Gpu1x32PerturbedScaled (T = double): View 5 is in double range (2^-148), so the plain-double orbit works."""
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.OrbitF64(v)
    r = renderer
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert r.ClearMemory() == 0
    co = ob.coords()
    assert r.RenderPerturbBLAScaled(None, ob, ob, None, None, co[0], co[1], co[2], co[3], v.num_iterations, T=T_F64) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, out) == 0
    assert r.SyncComputeStream() == 0
    ref = _oracle.gpu_scaled_f64(v, ob)
    assert np.array_equal(out, ref)
    # same picture as the plain-double perturbation path (Cpu64PerturbedBLA without BLA) up to binary32 glitches
    dbl = _oracle.bla_f64(v, ob, use_bla=False)
    d = np.abs(out[:36, :64].astype(np.int64) - dbl[:36, :64].astype(np.int64))
    assert np.median(d) <= 16 and (d <= 2).mean() > 0.25


@pytest.mark.parametrize("w,h,cap", [(70, 37, 15), (70, 37, 17), (33, 9, 300), (64, 36, 1)])
def test_scaled_hdr32_tuned_equals_literal_small_caps_and_ragged_frames(renderer, native_libs, w, h, cap):
    """Iteration limits around the shortest run (16 steps) and frames that are not a multiple of the 8 x 8 tile: the
    tuned scaled kernel's run-length votes and its four-entries-ahead loads against the statement-for-statement kernel
    and the oracle."""
    v = inputs.View.builtin(5, w, h, antialiasing=1)
    ob = inputs.Orbit(v)
    r = renderer
    assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
    dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
    outs = []
    try:
        for variant in (1, 0):
            assert r.set_kernel_variant(variant) == 0
            assert r.ClearMemory() == 0
            assert r.RenderPerturbBLAScaled(None, ob, ob, None, None, dx, dy, cx, cy, cap) == 0
            out = r.new_iter_buffer()
            assert r.RenderCurrent(cap, out) == 0
            assert r.SyncComputeStream() == 0
            outs.append(out)
    finally:
        r.set_kernel_variant(0)
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[1], _oracle.gpu_scaled_hdr32(v, ob, n_iterations=cap))
    assert not outs[1][h:, :].any() and not outs[1][:, w:].any()  # padding rows / columns stay zero


def test_scaled_f64_tuned_equals_literal(renderer, native_libs):
    v = inputs.View.builtin(5, 320, 180)
    ob = inputs.OrbitF64(v)
    r = renderer
    assert r.InitializeMemory(320, 180, 1, None, 0, 0, 0, False) == 0
    co = ob.coords()
    outs = []
    try:
        for variant in (1, 0):  # literal, tuned
            assert r.set_kernel_variant(variant) == 0
            assert r.ClearMemory() == 0
            assert r.RenderPerturbBLAScaled(None, ob, ob, None, None, co[0], co[1], co[2], co[3], v.num_iterations,
                                            T=T_F64) == 0
            out = r.new_iter_buffer()
            assert r.RenderCurrent(v.num_iterations, out) == 0
            assert r.SyncComputeStream() == 0
            outs.append(out)
    finally:
        r.set_kernel_variant(0)
    assert np.array_equal(outs[0], outs[1]) and outs[0].any()


def test_frame_from_an_orbit_loaded_from_an_im_file(renderer, native_libs, tmp_path):
    """SURVEY 8(f) row 3, end to end: View 5's HDRFloat<double> orbit saved as an Imagina .im file (993 bytes for 16 046
    entries), the view and the orbit loaded back from it, an LA table built from the loaded orbit, the frame rendered
    -- against the frame from the orbit GMP computed.  The reloaded orbit equals the original to ~1e-13 and the view to a
    double's precision, so the frames agree except where an iteration count sits on a rounding edge."""
    v = inputs.View.builtin(5, 64, 64, antialiasing=1)
    ob = inputs.Orbit(v, is64=True)
    p = tmp_path / "v5.im"
    ob.save_im(p)
    assert os.path.getsize(p) < 4096
    w = inputs.View.load_im(p, 64, 64)
    qb = inputs.Orbit.load_im(p, w)
    assert (qb.count, qb.period) == (ob.count, ob.period)
    a, _ = _render_lav2(renderer, v, ob, inputs.LATable(ob), LAV2_FULL, PARITY_CPU_GPUSTAGE)
    b, _ = _render_lav2(renderer, w, qb, inputs.LATable(qb), LAV2_FULL, PARITY_CPU_GPUSTAGE)
    a, b = a[:64, :64].astype(np.int64), b[:64, :64].astype(np.int64)
    assert (a == b).mean() > 0.9 and np.median(np.abs(a - b)) == 0
    assert abs(int(a.sum()) - int(b.sum())) < 0.01 * a.sum()


# ---- more built-in views: different depths (2^-60 ... 2^-2400), periods (59 ... 52 860), stage counts (1 ... 12)
@pytest.mark.parametrize("view_n", [2, 3, 9, 11])
@pytest.mark.parametrize("is64", [False, True])
def test_other_views_lav2_and_bla_parity(renderer, native_libs, view_n, is64):
    v = inputs.View.builtin(view_n, 64, 36, antialiasing=1)
    ob = inputs.Orbit(v, is64=is64)
    la = inputs.LATable(ob)
    for parity, st in ((PARITY_CPU, 0), (PARITY_CPU_GPUSTAGE, 1)):
        out, red = _render_lav2(renderer, v, ob, la, LAV2_FULL, parity)
        ref = _oracle.lav2_hdr32(v, ob, la, stage_test=st)
        assert np.array_equal(out, ref), (view_n, is64, parity)
        assert red.Sum == int(ref[:36, :64].astype(np.uint64).sum())
    if view_n != 9:  # perturbation only / BLA (view 9 needs 2e7 literal steps on the CPU side: covered by LAv2 above)
        out, _ = _render_lav2(renderer, v, ob, la, LAV2_PO, PARITY_CPU)
        assert np.array_equal(out, _oracle.bla_hdr32(v, ob, None))
        bla = inputs.BLATable(ob)
        r = renderer
        dx, dy, cx, cy = _pairs(v.coords_perturb(ob))
        assert r.RenderPerturbBLA(None, ob, bla, None, None, dx, dy, cx, cy, v.num_iterations) == 0
        out = r.new_iter_buffer()
        assert r.RenderCurrent(v.num_iterations, out) == 0
        assert r.SyncComputeStream() == 0
        assert np.array_equal(out, _oracle.bla_hdr32(v, ob, bla))


def test_other_views_2x32(renderer, native_libs):
    for view_n in (3, 11):
        v = inputs.View.builtin(view_n, 64, 36, antialiasing=1)
        o = inputs.Orbit(v, is64=True)
        la = inputs.LATable(o, use_small_exponents=True)
        o2, la2 = inputs.Orbit2x32(o), inputs.LATable2x32(la)
        out, _ = _render_2x32(renderer, v, o2, la2, LAV2_FULL)
        assert np.array_equal(out, _oracle.gpu_lav2_2x32(v, o2, la2, mode=0)), view_n


# ---- round 4: the packed X/Y perturbation step of k_lav2_2x32 (pt_step_pk) on orbits it has not seen: three centres (one ON
# the real axis, where parts of dz and of the orbit are exact zeros and the step must hand over to the literal code), zoom
# widths from 1e-8 to 1e-40 -- exponent gaps from a few binades to beyond the 120 at which an operand is dropped
_X2_CENTRES = [("-0.5482057480704757084582125675467330293766992786373239", "-0.5775708389036038428051089822018505586755517268027721"),
               ("-1.7685736563152709932817429153295447129341", "0.0"),
               ("-0.1528465308235274786391493323577", "1.0397032701234428320367513768879")]


@pytest.mark.parametrize("centre", [0, 1, 2])
@pytest.mark.parametrize("width", ["1e-8", "1e-14", "1e-22", "1e-31", "1e-40"])
def test_2x32_generated_views_full_and_perturbation_only(renderer, native_libs, centre, width):
    from decimal import Decimal, getcontext
    getcontext().prec = 80
    W, H = 48, 27
    cx, cy, w = Decimal(_X2_CENTRES[centre][0]), Decimal(_X2_CENTRES[centre][1]), Decimal(width)
    h = w * H / W
    v = inputs.View(str(cx - w / 2), str(cy - h / 2), str(cx + w / 2), str(cy + h / 2), W, H, num_iterations=30000)
    o = inputs.Orbit(v, is64=True)
    la = inputs.LATable(o, use_small_exponents=True)
    o2, la2 = inputs.Orbit2x32(o), inputs.LATable2x32(la)
    out, _ = _render_2x32(renderer, v, o2, la2, LAV2_FULL)
    assert np.array_equal(out, _oracle.gpu_lav2_2x32(v, o2, la2, mode=0)), (centre, width, "full")
    out, _ = _render_2x32(renderer, v, o2, None, LAV2_PO, n_iter=4000)
    assert np.array_equal(out, _oracle.gpu_lav2_2x32(v, o2, None, mode=1, n_iterations=4000)), (centre, width, "po")


# ---- non-HDR LAv2: Gpu1x32 / Gpu1x64 / Gpu2x32 PerturbedLAv2[PO|LAO] (Fractal's AUTO choice for zoom 1e4 .. 1e34)
def _render_plain(r, v, pin, mode, n_iter=None, bands=None):
    w, h = v.width * v.antialiasing, v.height * v.antialiasing
    assert r.InitializeMemory(w, h, v.antialiasing, None, 0, 0, 0, False) == 0
    if bands:
        assert r.SetRowBands(*bands) == 0
    assert r.InitializePerturbPlain(0, pin) == 0
    assert r.ClearMemory() == 0
    n = v.num_iterations if n_iter is None else n_iter
    assert r.RenderPerturbLAv2Plain(pin, n, Mode=mode) == 0
    assert r.SyncComputeStream() == 0
    out = r.new_iter_buffer()
    red = _capi.Reduction()
    assert r.RenderCurrent(n, out, None, red) == 0
    assert r.SyncComputeStream() == 0
    return out, red


@pytest.mark.parametrize("kind", ["f32", "f64", "2x32"])
@pytest.mark.parametrize("width", ["1e-6", "1e-12", "1e-20", "1e-28"])
def test_plain_lav2_parity(renderer, native_libs, kind, width):
    from test_plain_oracle import shallow_view
    v = shallow_view(width)
    pin = inputs.PlainInputs(v, kind)
    for mode, omode in ((LAV2_FULL, 0), (LAV2_PO, 1), (LAV2_LAO, 2)):
        out, red = _render_plain(renderer, v, pin, mode)
        ref = _oracle.gpu_lav2_plain(v, pin, mode=omode)
        assert np.array_equal(out, ref), (kind, width, mode, int((out != ref).sum()))
        assert red.Sum == int(ref[:36, :64].astype(np.uint64).sum())


@pytest.mark.parametrize("kind,width", [("f32", "1e-5"), ("f64", "1e-10"), ("2x32", "1e-10")])
def test_plain_lav2_from_an_im_file_orbit(renderer, native_libs, tmp_path, kind, width):
    """Round 4: ".im" files of the non-ExtendedRange types.  The orbit rebuilt from a file's waypoints (and the table built from
    it) renders to what the restated kernel makes of the same inputs -- and that frame is close to the saved orbit's."""
    from test_plain_oracle import shallow_view
    v = shallow_view(width)
    src = inputs.PlainInputs(v, "f32" if kind == "f32" else "f64")
    p = tmp_path / "plain.im"
    src.save_im(p)
    w = inputs.View.load_im(p, v.width, v.height)
    pin = inputs.PlainInputs.load_im(p, w, kind=kind)
    assert (pin.count, pin.period) == (src.count, src.period)
    out, _ = _render_plain(renderer, w, pin, LAV2_FULL)
    assert np.array_equal(out, _oracle.gpu_lav2_plain(w, pin, mode=0)), kind
    direct, _ = _render_plain(renderer, v, inputs.PlainInputs(v, kind), LAV2_FULL)
    a, b = out[:36, :64].astype(np.int64), direct[:36, :64].astype(np.int64)
    # an orbit good to the compression tolerance (1e-10 relative in binary64, 1e-3 in binary32): the same picture
    assert (np.abs(a - b) <= 2).mean() > (0.5 if kind == "f32" else 0.9)


def test_plain_lav2_odd_sizes_bands_and_antialiasing(renderer, native_libs):
    from test_plain_oracle import shallow_view
    v = shallow_view("1e-12", W=37, H=21)
    v.antialiasing = 2
    for kind in ("f32", "2x32"):
        pin = inputs.PlainInputs(v, kind)
        out, _ = _render_plain(renderer, v, pin, LAV2_FULL)
        ref = _oracle.gpu_lav2_plain(v, pin, aa=2, mode=0)
        assert np.array_equal(out, ref), kind


def test_quad_direct_kernels_on_a_deep_view(renderer, native_libs):
    """Gpu4x32 / Gpu4x64 where the extra words matter: a 1e-20-wide view (beyond binary64, inside quad-float's ~29 digits)."""
    from fractalshark_amd import T_4X32, T_4X64
    from test_plain_oracle import shallow_view
    v = shallow_view("1e-20", n_iter=12000, W=32, H=16)
    r = renderer
    assert r.InitializeMemory(32, 16, 1, None, 0, 0, 0, False) == 0
    outs = {}
    for kind, T in (("4x32", T_4X32), ("4x64", T_4X64)):
        assert r.ClearMemory() == 0
        assert r.RenderLowPrecision(None, v.coords_direct_lp(kind), v.num_iterations, 1, T=T) == 0
        out = r.new_iter_buffer()
        assert r.RenderCurrent(v.num_iterations, out) == 0
        assert r.SyncComputeStream() == 0
        assert np.array_equal(out, _oracle.gpu_direct_lp(v, kind, 1)), kind
        outs[kind] = out[:16, :32].astype(np.int64)
    assert len(np.unique(outs["4x64"])) > 20  # a real picture, not a constant
    assert (np.abs(outs["4x32"] - outs["4x64"]) <= 2).mean() > 0.9


# ---- the scaled quiet runs on inputs that stress their acceptance tests
def _axis_view(width, W=64, H=36, n_iter=60000):
    """A view centred ON the real axis (imaginary centre exactly 0, orbit and part of dz purely real): parts of dz and
    of the orbit are exact zeros, which the scaled runs must leave to the exponent-tracking path."""
    from decimal import Decimal, getcontext
    getcontext().prec = 80
    cx = Decimal("-1.7865720822115618956924187301180679424")  # a real-axis minibrot neighbourhood
    w = Decimal(width)
    h = w * H / W
    return inputs.View(str(cx - w / 2), str(-h / 2), str(cx + w / 2), str(h / 2), W, H, num_iterations=n_iter)


@pytest.mark.parametrize("width", ["1e-20", "1e-30"])
def test_scaled_runs_on_a_real_axis_view(renderer, native_libs, width):
    v = _axis_view(width)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    for parity, st in ((PARITY_CPU, 0), (PARITY_CPU_GPUSTAGE, 1)):
        out, _ = _render_lav2(renderer, v, ob, la, LAV2_FULL, parity)
        assert np.array_equal(out, _oracle.lav2_hdr32(v, ob, la, stage_test=st)), (width, parity)
    out, _ = _render_lav2(renderer, v, ob, la, LAV2_PO, PARITY_CPU)  # scalar-HDRFloat kernel
    assert np.array_equal(out, _oracle.bla_hdr32(v, ob, None)), width


@pytest.mark.parametrize("view_n", [3, 9, 11])
def test_three_variants_agree_on_other_views(renderer, native_libs, view_n):
    v = inputs.View.builtin(view_n, 96, 54, antialiasing=1)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    outs = []
    try:
        for variant in (1, 2, 0):
            assert renderer.set_kernel_variant(variant) == 0
            outs.append(_render_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU)[0])
    finally:
        renderer.set_kernel_variant(0)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


# ---- PerturbExtras::SimpleCompression for the remaining numeric types (Gpu1x32 / Gpu1x64 / Gpu2x32 / GpuHDRx2x32
# PerturbedRCLAv2*): waypoints are expanded on the device in the kernel's own arithmetic (kernels_decompress.hip)
@pytest.mark.parametrize("kind,width,cexp", [("f32", "1e-6", 10), ("f64", "1e-12", 20), ("f64", "1e-28", 20),
                                             ("2x32", "1e-12", 20), ("2x32", "1e-20", 16)])
def test_plain_simple_compression_parity(renderer, native_libs, kind, width, cexp):
    from test_plain_oracle import shallow_view
    v = shallow_view(width)
    pin = inputs.PlainInputs(v, kind, compression_exp=cexp)
    assert pin.compressed and 1 < pin.compressed_count < pin.count
    # what the reference's kernel rebuilds: for float / double the host RuntimeDecompressor's orbit, for CudaDblflt the
    # double-float rebuild of the converted waypoints
    full = _oracle.decompress_p2x32(pin) if kind == "2x32" else pin.orbit()
    for mode, omode in ((LAV2_FULL, 0), (LAV2_PO, 1), (LAV2_LAO, 2)):
        out, red = _render_plain(renderer, v, pin, mode)
        ref = _oracle.gpu_lav2_plain(v, pin, mode=omode, orbit_entries=full)
        assert np.array_equal(out, ref), (kind, width, mode, int((out != ref).sum()))
        assert red.Sum == int(ref[:36, :64].astype(np.uint64).sum())
    # the compressed upload is a different orbit from the uncompressed one (last bits), and the frames can tell
    if kind == "f32":
        u = inputs.PlainInputs(v, kind)
        assert not np.array_equal(u.orbit()["x"], full["x"])


@pytest.mark.parametrize("view_n", [5, 14])
def test_hdr2x32_simple_compression_parity(renderer, native_libs, view_n):
    v = inputs.View.builtin(view_n, 64, 36, antialiasing=1)
    o = inputs.Orbit(v, is64=True, compression_exp=20)
    la = inputs.LATable(o, use_small_exponents=True)
    o2, la2 = inputs.Orbit2x32(o), inputs.LATable2x32(la)
    assert o2.compressed and o2.compressed_count < o2.count
    full = _oracle.decompress_hdr2x32(o2)
    out, _ = _render_2x32(renderer, v, o2, la2, LAV2_FULL)
    ref = _oracle.gpu_lav2_2x32(v, o2, la2, mode=0, orbit_entries=full)
    assert np.array_equal(out, ref)
    assert len(np.unique(out[:36, :64])) > 16


def test_compressed_upload_rejects_missing_constants(renderer, native_libs):
    from test_plain_oracle import shallow_view
    from fractalshark_amd import T_F32
    pin = inputs.PlainInputs(shallow_view("1e-6"), "f32", compression_exp=10)
    assert renderer.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    e = renderer._lib.fs_upload_orbit_compressed(renderer._h, 0, T_F32, 4, pin.compressed_data_ptr, pin.compressed_count,
                                                 pin.count, pin.period, None, None)
    assert e != 0


def test_short_orbit_runs_and_oracle(renderer, native_libs):
    """An orbit shorter than a full scaled run (122 entries < 256): run lengths fall back to 64 / 16 by wave vote.  The
    three kernel variants agree and equal the CPU oracle."""
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    cx, cy = Decimal("-0.1528465308235274786391493323577"), Decimal("1.0397032701234428320367513768879")
    w = Decimal("1e-22")
    h = w * 36 / 64
    v = inputs.View(str(cx - w / 2), str(cy - h / 2), str(cx + w / 2), str(cy + h / 2), 64, 36, num_iterations=50000)
    ob = inputs.Orbit(v)
    assert ob.count < 256
    la = inputs.LATable(ob)
    outs = []
    try:
        for mode in (LAV2_FULL, LAV2_PO):
            for variant in (1, 2, 0):
                assert renderer.set_kernel_variant(variant) == 0
                outs.append(_render_lav2(renderer, v, ob, la, mode, PARITY_CPU)[0])
    finally:
        renderer.set_kernel_variant(0)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.array_equal(outs[3], outs[4]) and np.array_equal(outs[3], outs[5])
    assert np.array_equal(outs[0], _oracle.lav2_hdr32(v, ob, la, stage_test=0))
