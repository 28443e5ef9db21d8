"""GPU: "pixels in the order of the previous frame's counts" (csrc/kernels_order.hip) for the HDRFloat<double> and
HDRFloat<CudaDblflt> LAv2 kernels: the second frame of a view is launched with lane s rendering the pixel that ranked s-th by
count in the first -- a permutation of which lane renders which pixel, so every frame must be the first frame bit for bit (and
the oracle's rows), with row bands, after a change of the inputs (the stale order is a permutation of another view's ranking:
still every pixel exactly once), and with the A/B switch (FS_VARIANT_NATURAL_TILE_ORDER) or fs_forget_tile_costs the frames run
in the tile mapping again."""
import numpy as np
import pytest

import _oracle
from fractalshark_amd import (GPURenderer, LAV2_FULL, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR2X32, T_HDR64, inputs)

pytestmark = pytest.mark.gpu
W = H = 1024  # 2^20 elements: the smallest frame the order is made for


def _pairs(co):
    return [(float(c["m"]), int(c["e"])) for c in co]


@pytest.fixture(scope="module")
def renderer(native_libs):
    assert GPURenderer.TestCudaIsWorking() != 0, "no usable HIP device: the product path has no CPU fallback"
    r = GPURenderer(0)
    yield r
    r.set_kernel_variant(0)
    r.close()


def _frame(r, co, n, T, parity):
    assert r.ClearMemory() == 0
    assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T, Mode=LAV2_FULL, parity=parity) == 0
    sampled = r.last_frame_sampled_tile_order()
    out = r.new_iter_buffer()
    assert r.RenderCurrent(n, out) == 0
    assert r.SyncComputeStream() == 0
    _frame.sampled.append(sampled)
    return out, r.last_frame_tile_ordered()


_frame.sampled = []  # (per frame: was it a first frame launched in the order of a sampled PerformAT count, kernels_tile_sample.hip)


@pytest.mark.parametrize("view_n,parity,st", [(14, PARITY_CPU_GPUSTAGE, 1), (5, PARITY_CPU, 0)])
def test_hdr64_frames_in_count_order_are_the_first_frame(renderer, native_libs, view_n, parity, st):
    r = renderer
    v = inputs.View.builtin(view_n, W, H, antialiasing=1)
    ob = inputs.Orbit(v, is64=True)
    la = inputs.LATable(ob)
    co = _pairs(v.coords_perturb(ob))
    n = v.num_iterations if view_n == 14 else 20000
    assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
    assert r.forget_tile_costs() == 0
    # an order is made for a view that comes twice: the first frame runs as it is, the second records and sorts, the third and
    # later ones run ordered
    del _frame.sampled[:]
    first, ordered0 = _frame(r, co, n, T_HDR64, parity)
    second, ordered1 = _frame(r, co, n, T_HDR64, parity)
    third, ordered2 = _frame(r, co, n, T_HDR64, parity)
    fourth, ordered3 = _frame(r, co, n, T_HDR64, parity)
    # (round 6) the view's FIRST frame runs its tiles in the order of a sampled PerformAT count when the table has an AT; the later ones do not
    assert _frame.sampled[1:] == [False, False, False] and _frame.sampled[0] == bool(la.use_at)
    # (round 6: a frame whose table has an AT makes its own order from the AT pass -- ordered from the first frame on)
    inframe = False  # (FSMI355_C4_INFRAME_ORDER=1, an A/B that is off: DESIGN.md 7)
    assert (ordered0, ordered1, ordered2, ordered3) == ((True, True, True, True) if inframe else (False, False, True, True))
    assert np.array_equal(second, first) and np.array_equal(third, first) and np.array_equal(fourth, first)
    _oracle.set_row_step(255)
    try:
        ref = _oracle.lav2_hdr32(v, ob, la, rows=(3, H), stage_test=st, n_iterations=n)
    finally:
        _oracle.set_row_step(1)
    for y in range(3, H, 255):
        assert np.array_equal(second[y], ref[y]), y
    # other coordinates with the same geometry: the recorded order does not match -> tile mapping, and a new order after it
    co2 = list(co)
    co2[2] = (co[2][0] * 0.5, co[2][1])
    a, oa = _frame(r, co2, n, T_HDR64, parity)
    b, ob_ = _frame(r, co2, n, T_HDR64, parity)
    b2, ob2 = _frame(r, co2, n, T_HDR64, parity)
    assert (oa, ob_, ob2) == ((True, True, True) if inframe else (False, False, True)) and np.array_equal(a, b) and np.array_equal(a, b2)
    # the A/B switch and fs_forget_tile_costs
    assert r.set_kernel_variant(0, natural_tile_order=True) == 0
    del _frame.sampled[:]
    c, oc = _frame(r, co2, n, T_HDR64, parity)
    assert _frame.sampled == [False]  # (FS_VARIANT_NATURAL_TILE_ORDER: the tile mapping's own order, no sampled one either)
    assert r.set_kernel_variant(0) == 0
    assert oc is False and np.array_equal(c, a)
    assert r.forget_tile_costs() == 0
    d, od = _frame(r, co2, n, T_HDR64, parity)
    assert od is inframe and np.array_equal(d, a)


def test_hdr64_count_order_with_row_bands(renderer, native_libs):
    r = renderer
    v = inputs.View.builtin(14, 2048, 1536, antialiasing=1)
    ob = inputs.Orbit(v, is64=True)
    la = inputs.LATable(ob)
    co = _pairs(v.coords_perturb(ob))
    n = v.num_iterations
    assert r.InitializeMemory(2048, 1536, 1, None, 0, 0, 0, False) == 0
    assert r.SetRowBands(8, 8, 16) == 0  # the second of two ranks: 768 rows x 2048 = 1.5 M elements
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
    assert r.forget_tile_costs() == 0
    first, o0 = _frame(r, co, n, T_HDR64, PARITY_CPU_GPUSTAGE)
    mid, om = _frame(r, co, n, T_HDR64, PARITY_CPU_GPUSTAGE)
    second, o1 = _frame(r, co, n, T_HDR64, PARITY_CPU_GPUSTAGE)
    assert (o0, om, o1) == (False, False, True)
    assert np.array_equal(first, second) and np.array_equal(first, mid)
    ref = _oracle.lav2_hdr32(v, ob, la, rows=(8, 16), stage_test=1)
    assert np.array_equal(second[0:8, :2048], ref[8:16, :2048])


def test_2x32_frames_in_count_order_are_the_first_frame(renderer, native_libs):
    r = renderer
    v = inputs.View.builtin(14, W, H, antialiasing=1)
    ob = inputs.Orbit(v, is64=True)
    la = inputs.LATable(ob, use_small_exponents=True)
    o2, la2 = inputs.Orbit2x32(ob), inputs.LATable2x32(la)
    co = [(float(c["head"]), float(c["tail"]), int(c["e"])) for c in v.coords_perturb_2x32(o2)]
    assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, o2, 0, None, la2) == 0
    assert r.forget_tile_costs() == 0
    del _frame.sampled[:]
    first, o0 = _frame(r, co, v.num_iterations, T_HDR2X32, PARITY_CPU)
    mid, om = _frame(r, co, v.num_iterations, T_HDR2X32, PARITY_CPU)
    second, o1 = _frame(r, co, v.num_iterations, T_HDR2X32, PARITY_CPU)
    assert (o0, om, o1) == (False, False, True)
    assert _frame.sampled == [True, False, False]  # (the first frame: tiles in the order of a sampled PerformAT count)
    assert np.array_equal(first, second) and np.array_equal(first, mid)
    ref = _oracle.gpu_lav2_2x32(v, o2, la2, rows=(500, 504))
    assert np.array_equal(second[500:504], ref[500:504])


def test_at_order_is_not_trusted_across_row_bands_and_tables(renderer, native_libs):
    """Advisor finding of round 5: the AT pass's order had no key of its own.  Build it for a whole 2048 x 1536 frame, shrink the
    row bands, render the new view's first frames with a table WITHOUT an AT (so the pixel order is rebuilt for the smaller buffer
    while the AT order is not), then upload the table with its AT again: the next frame must not run the AT pass in the old, larger
    permutation (pixels without an AT result then kept stale ones).  Every frame == the first frame of its configuration."""
    import ctypes as C
    r = renderer
    v = inputs.View.builtin(14, 2048, 1536, antialiasing=1)
    ob = inputs.Orbit(v, is64=True)
    la = inputs.LATable(ob)
    assert la.use_at
    co = _pairs(v.coords_perturb(ob))
    n = v.num_iterations
    lib = r._lib

    def upload_la(use_at):
        return lib.fs_upload_la(r._h, 0, T_HDR64, 4, la.las_ptr, la.count, la.stages_ptr, la.stage_count, 1, 1 if use_at else 0,
                                C.addressof(la.at))

    assert r.InitializeMemory(2048, 1536, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
    assert r.forget_tile_costs() == 0
    whole = [_frame(r, co, n, T_HDR64, PARITY_CPU_GPUSTAGE) for _ in range(4)]
    assert [o for _, o in whole] == [False, False, True, True]
    assert all(np.array_equal(whole[0][0], f) for f, _ in whole)
    # the second of two ranks; a poisoned iteration buffer would show a pixel the frame's kernel skipped
    assert r.SetRowBands(8, 8, 16) == 0
    assert upload_la(False) == 0
    no_at = [_frame(r, co, n, T_HDR64, PARITY_CPU_GPUSTAGE) for _ in range(3)]
    assert [o for _, o in no_at] == [False, False, True]
    assert upload_la(True) == 0
    with_at = [_frame(r, co, n, T_HDR64, PARITY_CPU_GPUSTAGE) for _ in range(3)]
    assert r.forget_tile_costs() == 0
    fresh, _ = _frame(r, co, n, T_HDR64, PARITY_CPU_GPUSTAGE)
    for f, _ in with_at:
        assert np.array_equal(f, fresh)
    # the banded frame's rows are the whole frame's rows 8..15, 24..31, ...
    assert np.array_equal(fresh[0:8, :2048], whole[0][0][8:16, :2048])
    assert np.array_equal(fresh[8:16, :2048], whole[0][0][24:32, :2048])
    # (without AT the same pixels come out of the LA stages alone -- another arithmetic: only checked against itself)
    assert all(np.array_equal(no_at[0][0], f) for f, _ in no_at)
    assert r.SetRowBands(0, 0, 0) == 0
