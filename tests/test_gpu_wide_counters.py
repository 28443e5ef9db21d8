"""GPU: IterType = uint64_t counting in every render entry point (the reference templates each kernel on IterType and
instantiates uint64_t: GPU_Render.cu:849-991 Render, :1204-1300 RenderPerturbLAv2, :1380-1436 RenderPerturbBLAScaled,
:1610-1692 RenderPerturbBLA).

Two kinds of checks:
  * every entry point / numeric type with its 64-bit counting instantiation FORCED at the view's own cap
    (FS_VARIANT_WIDE_COUNTERS) against the 32-bit kernels (which the rest of the suite compares with the oracle): same
    frame, into 4-byte and into 8-byte buffers;
  * iteration caps above 2^32 on frames whose pixels all escape: same counts as under the view's own cap.  A never-escaping
    pixel costs 4.3e9 dependent steps (tens of minutes) wherever it is not carried by AT, so counts BEYOND 2^32 are
    exercised where AT does carry it: the HDRFloat<float | double> LAv2 kernels, against the CPU function instantiated for
    uint64_t (tests/test_gpu_parity.py::test_uint64_itertype_counts_past_2_to_32)."""
import numpy as np
import pytest

import _oracle
from fractalshark_amd import (GPURenderer, LAV2_FULL, LAV2_LAO, LAV2_PO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_2X32, T_2X64,
                              T_4X32, T_4X64, T_F32, T_F64, T_HDR2X32, T_HDR32, T_HDR64, _capi, inputs)

pytestmark = pytest.mark.gpu
BIG = (1 << 32) + 12345
MAX32 = (1 << 32) - 1


def _pairs(co):
    return [(float(c["m"]), int(c["e"])) for c in co]


@pytest.fixture(scope="module")
def renderer(native_libs):
    assert GPURenderer.TestCudaIsWorking() != 0, "no usable HIP device: the product path has no CPU fallback"
    r = GPURenderer(0)
    yield r
    r.set_kernel_variant(0)
    r.close()


def _frame(r, n):
    out = r.new_iter_buffer()
    assert r.RenderCurrent(n, out) == 0
    assert r.SyncComputeStream() == 0
    return out


def _both(r, w, h, aa, setup, render, n, iter_bytes=4):
    """Frame of `render` with the default (32-bit counting) kernels and with the 64-bit counting ones forced."""
    frames = []
    try:
        for wide in (False, True):
            assert r.set_kernel_variant(0, wide_counters=wide) == 0
            assert r.InitializeMemory(w, h, aa, None, 0, 0, 0, False, iter_bytes=iter_bytes) == 0
            setup()
            assert r.ClearMemory() == 0
            assert render() == 0
            frames.append(_frame(r, n))
    finally:
        r.set_kernel_variant(0)
    return frames


@pytest.mark.parametrize("iter_bytes", [4, 8])
@pytest.mark.parametrize("is64", [False, True])
def test_forced_wide_lav2_and_bla_hdr(renderer, native_libs, iter_bytes, is64):
    r = renderer
    v = inputs.View.builtin(5, 70, 37)
    ob = inputs.Orbit(v, is64=is64)
    la = inputs.LATable(ob)
    la_up = inputs.LATableU64(la) if iter_bytes == 8 else la
    bla = inputs.BLATable(ob)
    T = T_HDR64 if is64 else T_HDR32
    co = _pairs(v.coords_perturb(ob))
    n = v.num_iterations
    for mode, parity in ((LAV2_FULL, PARITY_CPU), (LAV2_FULL, PARITY_CPU_GPUSTAGE), (LAV2_LAO, PARITY_CPU_GPUSTAGE),
                         (LAV2_PO, PARITY_CPU), (LAV2_PO, PARITY_CPU_GPUSTAGE)):
        a, b = _both(r, 70, 37, 1, lambda: r.InitializePerturb(0, ob, 0, None, la_up, iter_bytes=iter_bytes),
                     lambda: r.RenderPerturbLAv2(None, None, None, *co, n, T=T, Mode=mode, parity=parity), n, iter_bytes)
        assert np.array_equal(a, b), (mode, parity, int((a != b).sum()))
    a, b = _both(r, 70, 37, 1, lambda: 0, lambda: r.RenderPerturbBLA(None, ob, bla, None, None, *co, n), n, iter_bytes)
    assert np.array_equal(a, b)
    assert np.array_equal(b[:37, :70].astype(np.uint64), _oracle.bla_hdr32(v, ob, bla)[:37, :70])


def test_forced_wide_plain_bla_scaled_and_direct(renderer, native_libs):
    r = renderer
    v = inputs.View.builtin(5, 64, 36)
    n = v.num_iterations
    # Gpu1x64PerturbedBLA (plain double)
    of = inputs.OrbitF64(v)
    cof = of.coords()

    def bla_f64():
        lib = r._lib
        assert lib.fs_upload_orbit(r._h, 0, T_F64, 4, of.data_ptr, of.count, of.count, of.period) == 0
        assert lib.fs_upload_bla(r._h, T_F64, of.level_ptrs, of.level_sizes, of.num_levels, of.lm2) == 0
        return lib.fs_render_bla(r._h, T_F64, cof.ctypes.data, n)
    a, b = _both(r, 64, 36, 1, lambda: 0, bla_f64, n)
    assert np.array_equal(a, b) and np.array_equal(b, _oracle.bla_f64(v, of))
    # scaled kernels, both T
    ob = inputs.Orbit(v)
    co = _pairs(v.coords_perturb(ob))
    a, b = _both(r, 64, 36, 1, lambda: 0, lambda: r.RenderPerturbBLAScaled(None, ob, ob, None, None, *co, n), n)
    assert np.array_equal(a, b) and np.array_equal(b, _oracle.gpu_scaled_hdr32(v, ob))
    a, b = _both(r, 64, 36, 1, lambda: 0,
                 lambda: r.RenderPerturbBLAScaled(None, of, of, None, None, cof[0], cof[1], cof[2], cof[3], n, T=T_F64), n)
    assert np.array_equal(a, b) and np.array_equal(b, _oracle.gpu_scaled_f64(v, of))
    # direct kernels with a CPU twin
    v0 = inputs.View.builtin(0, 70, 37)
    dx, dy, minx, maxy = v0.coords_direct_f64()
    a, b = _both(r, 70, 37, 1, lambda: 0, lambda: r.Render(None, minx, maxy, dx, dy, v0.num_iterations, T=T_F64),
                 v0.num_iterations)
    assert np.array_equal(a, b) and np.array_equal(b, _oracle.direct_f64(v0))
    for is64 in (False, True):
        dxh, dyh, mxh, myh = _pairs(v0.coords_direct_hdr(is64))
        a, b = _both(r, 70, 37, 1, lambda: 0,
                     lambda: r.Render(None, mxh, myh, dxh, dyh, v0.num_iterations, T=T_HDR64 if is64 else T_HDR32),
                     v0.num_iterations)
        assert np.array_equal(a, b) and np.array_equal(b, _oracle.direct_hdr(v0, is64))
    # low-precision direct kernels
    for kind, ip, T in (("1x32", 1, T_F32), ("1x32", 8, T_F32), ("2x32", 4, T_2X32), ("2x64", 1, T_2X64), ("4x32", 1, T_4X32),
                        ("4x64", 1, T_4X64)):
        a, b = _both(r, 70, 37, 1, lambda: 0,
                     lambda: r.RenderLowPrecision(None, v0.coords_direct_lp(kind), v0.num_iterations, ip, T=T),
                     v0.num_iterations)
        assert np.array_equal(a, b), (kind, ip)
        assert np.array_equal(b, _oracle.gpu_direct_lp(v0, kind, ip)), (kind, ip)


def test_forced_wide_2x32_plain_and_compressed(renderer, native_libs):
    from test_plain_oracle import shallow_view
    r = renderer
    v = inputs.View.builtin(5, 64, 36)
    o = inputs.Orbit(v, is64=True)
    la = inputs.LATable(o, use_small_exponents=True)
    o2, la2 = inputs.Orbit2x32(o), inputs.LATable2x32(la)
    tr = [(float(c["head"]), float(c["tail"]), int(c["e"])) for c in v.coords_perturb_2x32(o2)]
    n = v.num_iterations
    for mode, omode in ((LAV2_FULL, 0), (LAV2_PO, 1), (LAV2_LAO, 2)):
        a, b = _both(r, 64, 36, 1, lambda: r.InitializePerturb(0, o2, 0, None, la2),
                     lambda: r.RenderPerturbLAv2(None, None, None, *tr, n, T=T_HDR2X32, Mode=mode), n)
        assert np.array_equal(a, b), mode
        if mode != LAV2_PO:  # (the PO oracle run of this frame takes minutes; the 32-bit kernel is its checked twin)
            assert np.array_equal(b, _oracle.gpu_lav2_2x32(v, o2, la2, mode=omode))
    vs = shallow_view("1e-12")
    for kind in ("f32", "f64", "2x32"):
        pin = inputs.PlainInputs(vs, kind)
        for mode, omode in ((LAV2_FULL, 0), (LAV2_PO, 1), (LAV2_LAO, 2)):
            a, b = _both(r, 64, 36, 1, lambda: r.InitializePerturbPlain(0, pin),
                         lambda: r.RenderPerturbLAv2Plain(pin, vs.num_iterations, Mode=mode), vs.num_iterations)
            assert np.array_equal(a, b), (kind, mode)
            assert np.array_equal(b, _oracle.gpu_lav2_plain(vs, pin, mode=omode)), (kind, mode)
    # compressed orbit decompressed in the kernel
    oc = inputs.Orbit(v, compression_exp=20)
    lac = inputs.LATable(oc)
    co = _pairs(v.coords_perturb(oc))
    try:
        assert r.set_compressed_orbit_mode(True) == 0
        a, b = _both(r, 64, 36, 1, lambda: r.InitializePerturb(0, oc, 0, None, lac),
                     lambda: r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU), n)
    finally:
        r.set_compressed_orbit_mode(False)
    assert np.array_equal(a, b) and np.array_equal(b, _oracle.lav2_hdr32(v, oc, lac, stage_test=0))


@pytest.mark.parametrize("kind", ["hdr32", "hdr64", "f64"])
def test_bla_with_a_cap_above_2_to_32(renderer, native_libs, kind):
    """RenderPerturbBLA<uint64_t, T> with a cap above 2^32 on a frame whose pixels all escape: the counts are those of
    the largest 32-bit cap.  (A never-escaping pixel is not affordable here: its dz does not stay inside the table's
    validity radii, so it takes most of its 2^32 iterations as single steps -- tens of minutes for one pixel.  The 64-bit
    counting instantiation itself is compared with the oracle, interior pixels included, at the view's own cap in
    test_forced_wide_*; counts beyond 2^32 are exercised where AT carries the interior, below.)"""
    r = renderer
    v = inputs.View.builtin(9, 64, 36, antialiasing=1)
    ob32 = inputs.Orbit(v)
    ref = _oracle.bla_hdr32(v, ob32, inputs.BLATable(ob32))
    if int(ref.max()) >= v.num_iterations:
        pytest.skip("this frame has never-escaping pixels")
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=8) == 0
    frames = []
    for n in (MAX32, BIG):
        assert r.ClearMemory() == 0
        if kind == "f64":
            of = inputs.OrbitF64(v)
            lib = r._lib
            assert lib.fs_upload_orbit(r._h, 0, T_F64, 8, of.data_ptr, of.count, of.count, of.period) == 0
            assert lib.fs_upload_bla(r._h, T_F64, of.level_ptrs, of.level_sizes, of.num_levels, of.lm2) == 0
            assert lib.fs_render_bla(r._h, T_F64, of.coords().ctypes.data, n) == 0
        else:
            ob = inputs.Orbit(v, is64=kind == "hdr64")
            bla = inputs.BLATable(ob)
            assert r.RenderPerturbBLA(None, ob, bla, None, None, *_pairs(v.coords_perturb(ob)), n) == 0
        frames.append(_frame(r, n))
    assert frames[1].dtype == np.uint64 and np.array_equal(frames[0], frames[1])
    assert int(frames[1].max()) < v.num_iterations * 4
    if kind == "hdr32":
        assert np.array_equal(frames[1][:36, :64], ref[:36, :64].astype(np.uint64))
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=4) == 0


def test_2x32_and_plain_lav2_with_a_cap_above_2_to_32(renderer, native_libs):
    """RenderPerturbLAv2<uint64_t, ...> for HDRFloat<CudaDblflt> and the non-HDR types with a cap above 2^32, on frames
    whose pixels all escape (checked with the oracle at the view's own cap first): same counts as under that cap.
    (Counting BEYOND 2^32 is exercised by test_uint64_itertype_counts_past_2_to_32 on the HDRFloat<float | double>
    kernels, whose AT carries View 5's never-escaping pixels in seconds; with these types' tables a never-escaping pixel
    falls back to single steps, 4.3e9 of them.)"""
    from test_plain_oracle import shallow_view
    r = renderer
    v = inputs.View.builtin(9, 64, 36, antialiasing=1)
    o = inputs.Orbit(v, is64=True)
    la = inputs.LATable(o, use_small_exponents=True)
    o2, la2 = inputs.Orbit2x32(o), inputs.LATable2x32(la)
    ref = _oracle.gpu_lav2_2x32(v, o2, la2, mode=0)
    if int(ref.max()) < v.num_iterations:
        tr = [(float(c["head"]), float(c["tail"]), int(c["e"])) for c in v.coords_perturb_2x32(o2)]
        assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=8) == 0
        assert r.InitializePerturb(0, o2, 0, None, la2) == 0
        assert r.ClearMemory() == 0
        assert r.RenderPerturbLAv2(None, None, None, *tr, BIG, T=T_HDR2X32, Mode=LAV2_FULL) == 0
        out = _frame(r, BIG)
        assert out.dtype == np.uint64 and np.array_equal(out[:36, :64], ref[:36, :64].astype(np.uint64))
    vs = shallow_view("1e-12")
    ran = 0
    for kind in ("f32", "f64", "2x32"):
        pin = inputs.PlainInputs(vs, kind)
        ref = _oracle.gpu_lav2_plain(vs, pin, mode=0)
        if int(ref.max()) >= vs.num_iterations:
            continue
        assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=8) == 0
        assert r.InitializePerturbPlain(0, pin) == 0
        assert r.ClearMemory() == 0
        assert r.RenderPerturbLAv2Plain(pin, BIG, Mode=LAV2_FULL) == 0
        out = _frame(r, BIG)
        assert np.array_equal(out[:36, :64], ref[:36, :64].astype(np.uint64)), kind
        ran += 1
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=4) == 0
    if ran == 0 and int(ref.max()) >= vs.num_iterations:
        pytest.skip("every candidate frame has never-escaping pixels")


def test_caps_above_2_to_32_need_an_8_byte_buffer_and_are_served_everywhere(renderer, native_libs):
    """No entry point refuses an iteration cap for its size any more: with IterType = uint64_t every one renders; with a
    4-byte buffer such a cap is a caller error (hipErrorInvalidValue = 1), not FS_ERR_UNSUPPORTED."""
    r = renderer
    v = inputs.View.builtin(9, 64, 36, antialiasing=1)  # a view whose 64 x 36 frame has no never-escaping pixel
    ob = inputs.Orbit(v)
    co = _pairs(v.coords_perturb(ob))
    ref = _oracle.bla_hdr32(v, ob, None)
    if int(ref.max()) >= v.num_iterations:
        pytest.skip("this frame has never-escaping pixels: a step-by-step kernel would need 2^32 steps for them")
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=4) == 0
    assert r.RenderPerturbBLAScaled(None, ob, ob, None, None, *co, BIG) == 1
    assert r.RenderPerturbBLA(None, ob, None, None, None, *co, BIG) == 1
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=8) == 0
    # perturbation only (scalar kernel) and the scaled kernel take every iteration: all pixels escape, counts unchanged
    assert r.ClearMemory() == 0
    assert r.RenderPerturbBLA(None, ob, None, None, None, *co, BIG) == 0
    assert np.array_equal(_frame(r, BIG)[:36, :64], ref[:36, :64].astype(np.uint64))
    assert r.ClearMemory() == 0
    assert r.RenderPerturbBLAScaled(None, ob, ob, None, None, *co, BIG) == 0
    sc = _frame(r, BIG)
    assert np.array_equal(sc[:36, :64], _oracle.gpu_scaled_hdr32(v, ob)[:36, :64].astype(np.uint64))
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=4) == 0
