"""GPU: the hand-written BLA kernel (csrc/kernels_bla_fast.hip, the default of fs_render_bla for HDRFloat<float> with a table)
against the compiled kernel it replaces (fs_set_kernel_variant 2) and the CPU oracle (Cpu32PerturbedBLAHDR), where its
special cases live: iteration caps that end pixels inside a jump or a step (the statement drops a lane from its running
mask), orbits so short that jumps leave them (poisoned step counts) and rebases at the orbit's end, ragged frames (lanes
that never run), row bands, pixels on the real axis (exact zeros: the literal order), and the slow exits in general.
The reference's golden CRC-64s, the 40-view sweep and the BASELINE frame run through the same kernel (test_gpu_goldens.py,
test_gpu_sweep.py, test_gpu_full_size.py)."""
import numpy as np
import pytest

import _oracle
from fractalshark_amd import GPURenderer, T_HDR32, inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def renderer(native_libs):
    assert GPURenderer.TestCudaIsWorking() != 0, "no usable HIP device: the product path has no CPU fallback"
    r = GPURenderer(0)
    yield r
    r.set_kernel_variant(0)
    r.close()


def _frames(r, v, ob, bla, n, bands=None):
    """-> {variant: iteration buffer} for the default (hand-written) and the compiled kernel."""
    w, h = v.width, v.height
    assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
    if bands:
        assert r.SetRowBands(*bands) == 0
    lib = r._lib
    assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
    assert lib.fs_upload_bla(r._h, T_HDR32, bla.level_ptrs, bla.level_sizes, bla.num_levels, bla.lm2) == 0
    co = v.coords_perturb(ob)
    out = {}
    for variant in (2, 0):
        assert r.set_kernel_variant(variant) == 0
        assert r.ClearMemory() == 0
        assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, n) == 0
        buf = r.new_iter_buffer()
        assert r.RenderCurrent(n, buf) == 0
        assert r.SyncComputeStream() == 0
        out[variant] = buf
    r.set_kernel_variant(0)
    return out


@pytest.mark.parametrize("view_n,w,h", [(19, 70, 37), (5, 70, 37), (1, 96, 54), (3, 64, 36), (11, 64, 36)])
@pytest.mark.parametrize("cap", [1, 2, 3, 7, 64, 1000, 50000, None])
def test_iteration_caps_against_the_oracle_and_the_compiled_kernel(renderer, native_libs, view_n, w, h, cap):
    v = inputs.View.builtin(view_n, w, h, antialiasing=1)
    ob = inputs.Orbit(v)
    bla = inputs.BLATable(ob)
    n = v.num_iterations if cap is None else cap
    if cap is None and view_n == 19:
        n = 2_000_000  # (the oracle's share: the view's own 113 M cap is test_gpu_full_size.py's)
    f = _frames(renderer, v, ob, bla, n)
    assert np.array_equal(f[0], f[2]), (view_n, cap)
    ref = _oracle.bla_hdr32(v, ob, bla, n_iterations=n)
    assert np.array_equal(f[0], ref), (view_n, cap)
    assert int(f[0][:h, :w].max()) <= n


def test_row_bands_and_ragged_frames(renderer, native_libs):
    v = inputs.View.builtin(19, 100, 75, antialiasing=1)
    ob = inputs.Orbit(v)
    bla = inputs.BLATable(ob)
    n = 300_000
    ref = _oracle.bla_hdr32(v, ob, bla, n_iterations=n)
    for rank in range(3):
        f = _frames(renderer, v, ob, bla, n, bands=(rank * 8, 8, 24))
        assert np.array_equal(f[0], f[2]), rank
        k = 0
        for a in range(rank * 8, 75, 24):
            b = min(a + 8, 75)
            assert np.array_equal(f[0][k:k + (b - a), :100], ref[a:b, :100]), (rank, a)
            k += b - a


def test_real_axis_and_tiny_frames(renderer, native_libs):
    """A centre on the real axis (dc.im and dz.im are exact zeros for a whole row: the literal order, every trip) and frames
    smaller than a tile."""
    from decimal import Decimal, getcontext
    getcontext().prec = 80
    cx, wd = Decimal("-1.7685736563152709932817429153295447129341"), Decimal("1e-22")
    W, H = 33, 9  # (odd height: the middle row lies ON the axis)
    hgt = wd * H / W
    v = inputs.View(str(cx - wd / 2), str(-hgt / 2), str(cx + wd / 2), str(hgt / 2), W, H, num_iterations=50000)
    ob = inputs.Orbit(v)
    bla = inputs.BLATable(ob)
    f = _frames(renderer, v, ob, bla, 50000)
    assert np.array_equal(f[0], f[2])
    assert np.array_equal(f[0], _oracle.bla_hdr32(v, ob, bla, n_iterations=50000))
    for w, h in ((1, 1), (3, 2), (8, 8), (9, 1)):
        v2 = inputs.View.builtin(5, w, h, antialiasing=1)
        ob2 = inputs.Orbit(v2)
        bla2 = inputs.BLATable(ob2)
        f2 = _frames(renderer, v2, ob2, bla2, 20000)
        assert np.array_equal(f2[0], f2[2]), (w, h)
        assert np.array_equal(f2[0], _oracle.bla_hdr32(v2, ob2, bla2, n_iterations=20000)), (w, h)


def test_1080p_equals_the_compiled_kernel_and_sampled_oracle_rows(renderer, native_libs):
    v = inputs.View.builtin(19, 1920, 1080, antialiasing=1)
    ob = inputs.Orbit(v)
    bla = inputs.BLATable(ob)
    f = _frames(renderer, v, ob, bla, v.num_iterations)
    assert np.array_equal(f[0], f[2])
    _oracle.set_row_step(270)
    try:
        ref = _oracle.bla_hdr32(v, ob, bla, rows=(100, 1080))
    finally:
        _oracle.set_row_step(1)
    for y in range(100, 1080, 270):
        assert np.array_equal(f[0][y], ref[y]), y


@pytest.mark.parametrize("view_n,w,h,n", [(19, 320, 180, 2_000_000), (5, 100, 75, None), (19, 33, 9, 300_000)])
def test_workgroup_pooling_variant_is_identical(renderer, native_libs, view_n, w, h, n):
    """FS_VARIANT_BLA_POOL (A/B, off by default because it measures slower): every 32 trips the running pixels of a workgroup's
    four waves are re-packed into as few waves as possible through LDS -- a pixel's state moves between lanes and waves and its
    count is written by whichever lane finishes it.  Same frame as the default kernel and the oracle, ragged frames included."""
    v = inputs.View.builtin(view_n, w, h, antialiasing=1)
    ob = inputs.Orbit(v)
    bla = inputs.BLATable(ob)
    n = v.num_iterations if n is None else n
    r = renderer
    assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
    lib = r._lib
    assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
    assert lib.fs_upload_bla(r._h, T_HDR32, bla.level_ptrs, bla.level_sizes, bla.num_levels, bla.lm2) == 0
    co = v.coords_perturb(ob)
    frames = []
    try:
        for pool in (False, True):
            assert r.set_kernel_variant(0, bla_pool=pool) == 0
            assert r.ClearMemory() == 0
            assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, n) == 0
            buf = r.new_iter_buffer()
            assert r.RenderCurrent(n, buf) == 0
            assert r.SyncComputeStream() == 0
            frames.append(buf)
    finally:
        r.set_kernel_variant(0)
    assert np.array_equal(frames[0], frames[1])
    assert np.array_equal(frames[1], _oracle.bla_hdr32(v, ob, bla, n_iterations=n))
