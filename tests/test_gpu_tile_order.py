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


def test_perturbation_only_probe_order_is_reused_for_a_repeated_frame(renderer, native_libs):
    """fs_render_bla without a table (C2's path): the probe launch's tile order is a pure function of its inputs, so a
    repeated frame reuses it -- also when the orbit is re-uploaded with generation 0, as RenderPerturbBLA does on every
    call -- and anything that changes the inputs (coordinates, iteration limit, orbit, fs_forget_tile_costs) probes again.
    Every frame equals the first and the CPU function's rows."""
    r = renderer
    w, h = 640, 416  # 80 x 52 = 4160 tiles: enough for the reordering to switch on
    v = inputs.View.builtin(5, w, h)
    n = (1 << 18) + 8192  # the reordering needs a limit of 2^18 or more, for both limits used below
    ob = inputs.Orbit(v)
    co = _pairs(v.coords_perturb(ob))
    assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
    assert r.forget_tile_costs() == 0

    def frame(coords=co, cap=n, orbit=ob):
        assert r.ClearMemory() == 0
        assert r.RenderPerturbBLA(None, orbit, None, None, None, *coords, cap) == 0
        out = r.new_iter_buffer()
        assert r.RenderCurrent(cap, out) == 0
        assert r.SyncComputeStream() == 0
        return out[:h, :w].copy()

    first = frame()
    assert not r.last_frame_tile_ordered()
    second = frame()  # same orbit uploaded again (generation 0): recognised by its fingerprint
    assert r.last_frame_tile_ordered()
    assert np.array_equal(first, second)
    ref = _oracle.bla_hdr32(v, ob, None, rows=(204, 212), n_iterations=n)
    assert np.array_equal(first[204:212], ref[204:212, :w])
    # a lower iteration limit, other coordinates, a forgotten order: each probes again, and the frame after is warm
    frame(cap=n - 4096)
    assert not r.last_frame_tile_ordered()
    frame(cap=n - 4096)
    assert r.last_frame_tile_ordered()
    moved = [co[0], co[1], (co[2][0] * 1.0000001, co[2][1]), co[3]]
    frame(coords=moved)
    assert not r.last_frame_tile_ordered()
    assert np.array_equal(frame(), first)
    assert not r.last_frame_tile_ordered()  # the coordinates changed back: probed again
    assert np.array_equal(frame(), first) and r.last_frame_tile_ordered()
    assert r.forget_tile_costs() == 0
    assert np.array_equal(frame(), first) and not r.last_frame_tile_ordered()
    # another orbit (View 3) at the same geometry: probed again
    v3 = inputs.View.builtin(3, w, h)
    ob3 = inputs.Orbit(v3)
    frame(coords=_pairs(v3.coords_perturb(ob3)), orbit=ob3)
    assert not r.last_frame_tile_ordered()
    # natural order: no probe, never "ordered"; same pixels
    r.set_kernel_variant(0x800)
    assert np.array_equal(frame(), first) and not r.last_frame_tile_ordered()
    r.set_kernel_variant(0)
