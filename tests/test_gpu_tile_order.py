"""GPU tests of "longest tiles first, self-recorded" (fs_render_lav2 of the tuned HDRFloat<float> kernel).

A frame records one cost word per 8 x 8 tile; the next frame of the same geometry / row bands / orbit generation is launched
in descending cost order.  Which wave renders which tile must change no pixel: cold and warm frames are compared with each
other and with the CPU oracle, the recorded costs with the oracle's own iteration counts, the launch order with a host-side
sort of the same costs."""
import numpy as np
import pytest

import _oracle
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_LAO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR32, inputs

pytestmark = pytest.mark.gpu


def _pairs(co):
    return [(float(c["m"]), int(c["e"])) for c in co]


@pytest.fixture(scope="module")
def renderer(native_libs):
    assert GPURenderer.TestCudaIsWorking() != 0, "no usable HIP device: the product path has no CPU fallback"
    r = GPURenderer(0)
    yield r
    r.set_kernel_variant(0)
    r.close()


def _frame(r, v, ob, parity, mode=LAV2_FULL):
    n = v.num_iterations
    assert r.ClearMemory() == 0
    assert r.RenderPerturbLAv2(None, None, None, *_pairs(v.coords_perturb(ob)), n, T=T_HDR32, Mode=mode, parity=parity) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(n, out) == 0
    assert r.SyncComputeStream() == 0
    return out


def _setup(r, v, ob, la, bands=None, gen=1):
    assert r.InitializeMemory(v.width, v.height, 1, None, 0, 0, 0, False) == 0
    if bands:
        assert r.SetRowBands(*bands) == 0
    assert r.InitializePerturb(gen, ob, 0, None, la) == 0


def _host_order(cost):
    """The order the device sort must produce: 256 cost classes (exponent and three mantissa bits of the cost as a float, 8
    per octave), highest first, stable inside a class."""
    f = cost.astype(np.float32)
    q = (f.view(np.uint32) >> 20).astype(np.int64) - (127 << 3)
    q = np.where(cost == 0, 0, np.clip(q, 0, 255))
    cls = 255 - q
    return np.argsort(cls, kind="stable").astype(np.uint32)


@pytest.mark.parametrize("w,h", [(960, 544), (1001, 517)])
@pytest.mark.parametrize("parity,st", [(PARITY_CPU, 0), (PARITY_CPU_GPUSTAGE, 1)])
def test_cold_and_warm_frames_are_identical_and_equal_the_oracle(renderer, native_libs, w, h, parity, st):
    r = renderer
    v = inputs.View.builtin(5, w, h)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    _setup(r, v, ob, la, gen=10 + st)
    assert r.forget_tile_costs() == 0
    cold = _frame(r, v, ob, parity)
    assert not r.last_frame_tile_ordered()
    cost_cold = r.read_tile_costs()
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    assert cost_cold is not None and cost_cold.size == tiles_x * tiles_y
    warm = _frame(r, v, ob, parity)
    assert r.last_frame_tile_ordered()
    order = r.read_tile_order(cost_cold.size)
    cost_warm = r.read_tile_costs()
    assert np.array_equal(cold, warm), "the launch order changed %d pixels" % int((cold != warm).sum())
    # the cost of a tile does not depend on when it ran
    assert np.array_equal(cost_cold, cost_warm)
    # the order is the stable class sort of the recorded costs: a permutation, longest first
    assert np.array_equal(np.sort(order), np.arange(cost_cold.size, dtype=np.uint32))
    assert np.array_equal(order, _host_order(cost_cold))
    # a third frame is warm too, and equal again
    again = _frame(r, v, ob, parity)
    assert r.last_frame_tile_ordered()
    assert np.array_equal(again, cold)
    # against the oracle on sampled rows (whole frame for the small one would take minutes)
    rows = list(range(3, h, max(1, h // 12)))
    _oracle.set_row_step(max(1, h // 12))
    try:
        ref = _oracle.lav2_hdr32(v, ob, la, rows=(3, h), stage_test=st)
    finally:
        _oracle.set_row_step(1)
    for y in rows:
        assert np.array_equal(warm[y, :w], ref[y, :w]), "row %d differs from the oracle" % y
    # the recorded cost of a tile is bounded by its pixels' counts: the longest lane's perturbation steps (+ what ran before
    # the loop), never more than the largest iteration count of the tile (+ 8 per LA step, none in the literal CPU direction)
    if st == 0:
        it = np.zeros((tiles_y * 8, tiles_x * 8), np.uint64)
        it[:h, :w] = warm[:h, :w]
        tile_max = it.reshape(tiles_y, 8, tiles_x, 8).max(axis=(1, 3)).reshape(-1)
        assert (cost_cold.astype(np.uint64) <= tile_max + 1).all()
        assert (cost_cold > 0).all()


def test_changes_that_make_the_next_frame_cold(renderer, native_libs):
    r = renderer
    v = inputs.View.builtin(5, 640, 512)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    _setup(r, v, ob, la, gen=21)
    r.forget_tile_costs()
    a = _frame(r, v, ob, PARITY_CPU)
    assert not r.last_frame_tile_ordered()
    b = _frame(r, v, ob, PARITY_CPU)
    assert r.last_frame_tile_ordered()
    # a new orbit generation: cold
    assert r.InitializePerturb(22, ob, 0, None, la) == 0
    c = _frame(r, v, ob, PARITY_CPU)
    assert not r.last_frame_tile_ordered()
    d = _frame(r, v, ob, PARITY_CPU)
    assert r.last_frame_tile_ordered()
    # FS_VARIANT_NATURAL_TILE_ORDER: never ordered, nothing recorded
    assert r.set_kernel_variant(0, natural_tile_order=True) == 0
    e = _frame(r, v, ob, PARITY_CPU)
    assert not r.last_frame_tile_ordered()
    assert r.read_tile_costs() is None
    assert r.set_kernel_variant(0) == 0
    f = _frame(r, v, ob, PARITY_CPU)
    assert not r.last_frame_tile_ordered()  # nothing valid was left to order by
    g = _frame(r, v, ob, PARITY_CPU)
    assert r.last_frame_tile_ordered()
    # another geometry: cold
    v2 = inputs.View.builtin(5, 648, 512)
    _setup(r, v2, ob, la, gen=22)
    h2 = _frame(r, v2, ob, PARITY_CPU)
    assert not r.last_frame_tile_ordered()
    for x in (b, c, d, e, f, g):
        assert np.array_equal(a, x)
    assert h2.shape != a.shape or not np.array_equal(h2, a)


def test_row_bands_keep_their_own_costs_and_reassemble(renderer, native_libs):
    """Each rank of the 8-way split records and reuses the costs of ITS bands; the reassembled warm frame equals the
    whole-frame render."""
    r = renderer
    w, h = 1280, 1024
    v = inputs.View.builtin(5, w, h)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    _setup(r, v, ob, la, gen=31)
    r.forget_tile_costs()
    whole = _frame(r, v, ob, PARITY_CPU)
    world, band = 4, 8
    out = np.zeros_like(whole)
    for rank in range(world):
        _setup(r, v, ob, la, bands=(rank * band, band, world * band), gen=31)
        cold = _frame(r, v, ob, PARITY_CPU)
        assert not r.last_frame_tile_ordered()
        warm = _frame(r, v, ob, PARITY_CPU)
        assert r.last_frame_tile_ordered()
        assert np.array_equal(cold, warm)
        k = 0
        for start in range(rank * band, h, world * band):
            n = min(band, h - start)
            out[start:start + n] = warm[k:k + n]
            k += n
    assert np.array_equal(out[:h, :w], whole[:h, :w])


def test_lao_mode_records_and_reuses(renderer, native_libs):
    r = renderer
    v = inputs.View.builtin(5, 640, 512)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    _setup(r, v, ob, la, gen=41)
    r.forget_tile_costs()
    a = _frame(r, v, ob, PARITY_CPU_GPUSTAGE, mode=LAV2_LAO)
    b = _frame(r, v, ob, PARITY_CPU_GPUSTAGE, mode=LAV2_LAO)
    assert r.last_frame_tile_ordered()
    assert np.array_equal(a, b)
