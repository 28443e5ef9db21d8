"""GPU tests of the two A/B kernel variants that are off by default (north_star rows N1 / N2):

  FS_VARIANT_LDS_ORBIT  k_lav2_hdr32_fast<kLds>: orbit entries of the scaled runs staged through LDS (LDS-DMA double buffer)
  FS_VARIANT_REFILL     k_perturb_scalar<kRefill>: persistent launch, finished lanes refilled from a pixel queue
                        (wave-ballot compaction)

Both must be bit-identical to the default kernels (and therefore to the oracle) on every frame; they are selected per
renderer with fs_set_kernel_variant, so one process can compare them.  Also here: the executed-work counters of the
headline kernel (the numerator of bench.py's roofline) against the oracle's own counts."""
import numpy as np
import pytest

import _oracle
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_LAO, LAV2_PO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR32, T_HDR64, _capi, inputs

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


def _lav2(r, v, ob, la, mode, parity, n_iter=None, bands=None):
    w, h = v.width * v.antialiasing, v.height * v.antialiasing
    assert r.InitializeMemory(w, h, v.antialiasing, None, 0, 0, 0, False) == 0
    if bands:
        assert r.SetRowBands(*bands) == 0
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
    assert r.ClearMemory() == 0
    n = v.num_iterations if n_iter is None else n_iter
    assert r.RenderPerturbLAv2(None, None, None, *_pairs(v.coords_perturb(ob)), n, T=T_HDR64 if ob.is64 else T_HDR32,
                               Mode=mode, parity=parity) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(n, out) == 0
    assert r.SyncComputeStream() == 0
    return out


def _bla(r, v, ob, bla, n_iter=None):
    w, h = v.width * v.antialiasing, v.height * v.antialiasing
    assert r.InitializeMemory(w, h, v.antialiasing, None, 0, 0, 0, False) == 0
    assert r.ClearMemory() == 0
    n = v.num_iterations if n_iter is None else n_iter
    assert r.RenderPerturbBLA(None, ob, bla, None, None, *_pairs(v.coords_perturb(ob)), n) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(n, out) == 0
    assert r.SyncComputeStream() == 0
    return out


def test_variant_selection_is_validated(renderer):
    r = renderer
    assert r._lib.fs_set_kernel_variant(r._h, 3) != 0            # no such base variant
    assert r._lib.fs_set_kernel_variant(r._h, 0x2000) != 0       # no such flag
    assert r._lib.fs_set_kernel_variant(r._h, 0x1000) == 0       # FS_VARIANT_BLA_POOL
    assert r._lib.fs_set_kernel_variant(r._h, 0x800) == 0        # FS_VARIANT_NATURAL_TILE_ORDER
    assert r._lib.fs_set_kernel_variant(r._h, 0x100 | 0x200) == 0
    assert r._lib.fs_set_kernel_variant(r._h, 0) == 0


# ---- N1: orbit entries through LDS
@pytest.mark.parametrize("w,h", [(64, 36), (37, 21), (320, 180)])
@pytest.mark.parametrize("parity,st", [(PARITY_CPU, 0), (PARITY_CPU_GPUSTAGE, 1)])
def test_lds_orbit_variant_equals_default_and_oracle_view5(renderer, native_libs, w, h, parity, st):
    v = inputs.View.builtin(5, w, h)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    ref = _oracle.lav2_hdr32(v, ob, la, stage_test=st)
    try:
        for mode, omode in ((LAV2_FULL, 0), (LAV2_LAO, 2)):
            assert renderer.set_kernel_variant(0) == 0
            base = _lav2(renderer, v, ob, la, mode, parity)
            assert renderer.set_kernel_variant(0, lds_orbit=True) == 0
            lds = _lav2(renderer, v, ob, la, mode, parity)
            assert np.array_equal(lds, base), (w, h, parity, mode, int((lds != base).sum()))
            if mode == LAV2_FULL:
                assert np.array_equal(lds, ref)
    finally:
        renderer.set_kernel_variant(0)


def test_lds_orbit_variant_1080p_rows_and_bands(renderer, native_libs):
    """A frame large enough for long wave-uniform scaled runs (where the LDS pipeline actually streams chunks), whole and
    as the second of three interleaved row bands."""
    v = inputs.View.builtin(5, 960, 540)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    try:
        for bands in (None, (8, 8, 24)):
            got = []
            for lds_on in (False, True):
                assert renderer.set_kernel_variant(0, lds_orbit=lds_on) == 0
                renderer.enable_step_count(True)
                out = _lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU, bands=bands)
                got.append((out, renderer.read_step_count()))
                renderer.enable_step_count(False)
            (base, st0), (lds, st1) = got
            assert np.array_equal(lds, base), int((lds != base).sum())
            # same executed work, same careful (exit-tested) steps -- and the variant really ran scaled runs through its LDS
            # pipeline (this is not a fall-back path being compared)
            for k in ("perturb_steps", "careful_steps", "at_iterations", "pixels"):
                assert st0[k] == st1[k], (k, st0[k], st1[k])
            assert st1["scaled_steps"] > 0 and st1["scaled_runs"] > 0
    finally:
        renderer.enable_step_count(False)
        renderer.set_kernel_variant(0)
    rows = list(range(3, 540, 67))
    full = _lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU)
    _oracle.set_row_step(67)
    try:
        ref = _oracle.lav2_hdr32(v, ob, la, rows=(3, 540), stage_test=0)
    finally:
        _oracle.set_row_step(1)
    assert all(np.array_equal(full[y], ref[y]) for y in rows)


@pytest.mark.parametrize("view_n", [3, 9, 11])
def test_lds_orbit_variant_other_views(renderer, native_libs, view_n):
    v = inputs.View.builtin(view_n, 96, 54, antialiasing=1)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    try:
        assert renderer.set_kernel_variant(0) == 0
        base = _lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU)
        assert renderer.set_kernel_variant(0, lds_orbit=True) == 0
        assert np.array_equal(_lav2(renderer, v, ob, la, LAV2_FULL, PARITY_CPU), base)
    finally:
        renderer.set_kernel_variant(0)


# ---- N2: persistent launch with lane refill
@pytest.mark.parametrize("view_n,w,h", [(19, 64, 36), (19, 37, 21), (19, 200, 120), (5, 64, 36), (1, 96, 54)])
@pytest.mark.parametrize("is64", [False, True])
def test_refill_variant_equals_default_and_oracle(renderer, native_libs, view_n, w, h, is64):
    if is64 and (w, h) == (200, 120):
        pytest.skip("the larger frame is covered in HDRFloat<float>")
    v = inputs.View.builtin(view_n, w, h, antialiasing=1)
    ob = inputs.Orbit(v, is64=is64)
    bla = inputs.BLATable(ob)
    try:
        assert renderer.set_kernel_variant(0) == 0
        base = _bla(renderer, v, ob, bla)
        assert renderer.set_kernel_variant(0, refill=True) == 0
        ref = _bla(renderer, v, ob, bla)
        assert np.array_equal(ref, base), (view_n, w, h, is64, int((ref != base).sum()))
        if w * h <= 64 * 36:
            oracle = _oracle.bla_hdr32(v, ob, bla)  # (follows the orbit's type)
            assert np.array_equal(ref, oracle)
    finally:
        renderer.set_kernel_variant(0)


def test_refill_variant_counts_every_pixel_once(renderer, native_libs):
    """The queue hands out every pixel of a ragged frame exactly once (pixel counter of the instrumented build) and the
    lane slots it occupies are fewer than the one-tile-per-wave launch's (that is the point of the compaction)."""
    v = inputs.View.builtin(19, 203, 117, antialiasing=1)
    ob = inputs.Orbit(v)
    bla = inputs.BLATable(ob)
    r = renderer
    try:
        got = {}
        for name, refill in (("tiles", False), ("refill", True)):
            assert r.set_kernel_variant(0, refill=refill) == 0
            r.enable_step_count(True)
            out = _bla(r, v, ob, bla)
            st = r.read_step_count()
            r.enable_step_count(False)
            got[name] = (out, st)
            assert st["pixels"] == 203 * 117
        assert np.array_equal(got["tiles"][0], got["refill"][0])
        for k in ("perturb_steps", "la_steps"):
            assert got["tiles"][1][k] == got["refill"][1][k]
    finally:
        r.enable_step_count(False)
        r.set_kernel_variant(0)


# ---- long tiles first (perturbation only)
@pytest.mark.parametrize("w,h,bands", [(512, 512, None), (520, 517, None), (512, 1024, (8, 8, 16))])
def test_long_tiles_first_changes_no_pixel(renderer, native_libs, w, h, bands):
    """fs_render_bla without BLA reorders the launch of a frame's 8 x 8 tiles from a probe of their centre pixels once the
    frame has 4096 tiles and the iteration limit is 2^18 or more (C2's regime).  Same frame with the reordering on (default)
    and off (FS_VARIANT_NATURAL_TILE_ORDER): identical buffers, pixels at the limit included, on a square frame, a ragged
    one and one rank's row bands; the rows around the frame's busiest tile against the oracle."""
    v = inputs.View.builtin(5, w, h, antialiasing=1)
    ob = inputs.Orbit(v)
    n = 300000  # above every pixel's escape time that escapes below it; the interior pixels stop here
    r = renderer
    got = {}
    try:
        for name, natural in (("reordered", False), ("natural", True)):
            assert r.set_kernel_variant(0, natural_tile_order=natural) == 0
            assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
            if bands:
                assert r.SetRowBands(*bands) == 0
            assert r.ClearMemory() == 0
            assert r.RenderPerturbBLA(None, ob, None, None, None, *_pairs(v.coords_perturb(ob)), n) == 0
            out = r.new_iter_buffer()
            assert r.RenderCurrent(n, out) == 0
            assert r.SyncComputeStream() == 0
            got[name] = out.copy()
        assert np.array_equal(got["reordered"], got["natural"])
        img = got["reordered"][:r.local_rows, :w]
        assert img.max() <= n
        if not bands:
            y = int(np.unravel_index(int(img.argmax()), img.shape)[0])
            y0, y1 = max(0, y - 1), min(h, y + 2)
            ref = _oracle.bla_hdr32(v, ob, None, rows=(y0, y1), n_iterations=n)
            assert np.array_equal(img[y0:y1], ref[y0:y1, :w])
    finally:
        r.set_kernel_variant(0)


# ---- the roofline numerator of the headline kernel
@pytest.mark.parametrize("parity,st", [(PARITY_CPU, 0), (PARITY_CPU_GPUSTAGE, 1)])
def test_lav2_hdr32_step_counters_equal_the_oracles(renderer, native_libs, parity, st):
    """bench.py's roofline.achieved = perturb_steps x 18 flop / kernel time, with perturb_steps read from the
    instrumented build of k_lav2_hdr32_fast.  Here: AT iterations, LA steps, perturbation steps and pixels of a 256 x 256
    View 5 frame equal the CPU function's own counts, and reconcile with the frame (sum of the iteration counts)."""
    v = inputs.View.builtin(5, 256, 256)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    ref, ost = _oracle.lav2_hdr32(v, ob, la, stage_test=st, stats=True)
    r = renderer
    r.set_kernel_variant(0)
    try:
        r.enable_step_count(True)
        out = _lav2(r, v, ob, la, LAV2_FULL, parity)
        got = r.read_step_count()
    finally:
        r.enable_step_count(False)
    assert np.array_equal(out, ref)
    for k in ("at_iterations", "la_steps", "perturb_steps", "pixels"):
        assert got[k] == ost[k], (k, got[k], ost[k])
    assert got["careful_steps"] + got["scaled_steps"] <= got["perturb_steps"]
    assert got["lane_slots"] >= got["perturb_steps"]
    if st == 0:
        # no LA step is taken in the CPU direction at View 5: every counted iteration is an AT block or a perturbation
        # step, and an escaping pixel's last step is executed but not counted (`break` before ++iterations)
        assert got["la_steps"] == 0
        escaped = int((out[:256, :256] < v.num_iterations).sum())
        at_len = int(la.at.StepLength)
        if la.use_at:
            assert int(out[:256, :256].astype(np.uint64).sum()) == got["at_iterations"] * at_len + got["perturb_steps"] - escaped


# ---- the device-native BLA table (FsBlaRec + ladder keys) against the reference-layout lookup
@pytest.mark.parametrize("view_n,w,h", [(19, 64, 36), (19, 203, 117), (5, 64, 36), (1, 96, 54), (9, 96, 54)])
def test_native_bla_table_equals_reference_layout_lookup(renderer, native_libs, view_n, w, h):
    """The HDRFloat<float> BLA kernel reads its table in a device-native form by default (48-byte records, the four r2 a
    lookup probes next as 64-bit integer keys in one ladder entry); variant 1 keeps the lookup on the reference-layout
    records (44-byte AoS, extended-exponent compares).  Same frames, same jumps, same steps -- with the host-built table
    and with the table built on the device."""
    v = inputs.View.builtin(view_n, w, h, antialiasing=1)
    ob = inputs.Orbit(v)
    bla = inputs.BLATable(ob)
    r = renderer
    got = {}
    try:
        for name, variant in (("native", 0), ("reference_layout", 1)):
            assert r.set_kernel_variant(variant) == 0
            r.enable_step_count(True)
            out = _bla(r, v, ob, bla)
            got[name] = (out, r.read_step_count())
            r.enable_step_count(False)
        # device-built table (fs_build_bla), native form made from it
        assert r.set_kernel_variant(0) == 0
        assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
        assert r._lib.fs_upload_orbit(r._h, 0, T_HDR32, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
        assert r.BuildBLAOnDevice(ob) == 0
        assert r._lib.fs_render_bla(r._h, T_HDR32, v.coords_perturb(ob).ctypes.data, v.num_iterations) == 0
        dev = r.new_iter_buffer()
        assert r.RenderCurrent(v.num_iterations, dev) == 0
        assert r.SyncComputeStream() == 0
    finally:
        r.enable_step_count(False)
        r.set_kernel_variant(0)
    assert np.array_equal(got["native"][0], got["reference_layout"][0]), int((got["native"][0] != got["reference_layout"][0]).sum())
    assert np.array_equal(dev, got["native"][0])
    for k in ("perturb_steps", "la_steps", "pixels"):
        assert got["native"][1][k] == got["reference_layout"][1][k], k
    if w * h <= 64 * 36:
        assert np.array_equal(got["native"][0], _oracle.bla_hdr32(v, ob, bla))


# ---- the step-counting instantiation of the perturbation-only kernel (k_perturb_scalar<float, false, true>): the build
# bench.py's roofline numerator for C2 is read from, and the one tests/test_isa_asm_registers.py's liveness analysis leaves out
# (path-insensitive: it cannot rule out one compiler-allocated scalar).  Dynamic evidence instead: its frame is the plain
# kernel's and the oracle's, its counts are those of the statement-for-statement variant, and they reconcile with the frame.
@pytest.mark.parametrize("w,h", [(256, 144), (200, 120)])
def test_counting_instantiation_of_the_perturbation_only_kernel(renderer, native_libs, w, h):
    v = inputs.View.builtin(5, w, h, antialiasing=1)
    ob = inputs.Orbit(v)
    r = renderer
    n = v.num_iterations
    got = {}
    try:
        for name, variant, counting in (("plain", 0, False), ("counting", 0, True), ("literal_counting", 1, True)):
            assert r.set_kernel_variant(variant) == 0
            r.enable_step_count(counting)
            out = _bla(r, v, ob, None)
            got[name] = (out, r.read_step_count() if counting else None)
            r.enable_step_count(False)
    finally:
        r.enable_step_count(False)
        r.set_kernel_variant(0)
    ref = _oracle.bla_hdr32(v, ob, None)
    assert np.array_equal(got["plain"][0], ref)
    assert np.array_equal(got["counting"][0], ref)
    assert np.array_equal(got["literal_counting"][0], ref)
    st, lit = got["counting"][1], got["literal_counting"][1]
    assert st["pixels"] == lit["pixels"] == w * h
    frame = ref[:h, :w].astype(np.uint64)
    escaped = int((frame < n).sum())
    capped = int((frame >= n).sum())
    assert capped > 0  # interior pixels: the waves this kernel's frame time is made of
    # every counted iteration is an executed step; an escaping pixel's last step is executed and not counted; a pixel the cycle
    # test ends early (both variants: the test sits at the rebase, outside the parts that differ) has executed fewer steps than
    # it reports -- so the two variants must agree with each other exactly and with the frame up to the capped pixels
    total = int(frame.sum()) + escaped
    assert st["perturb_steps"] == lit["perturb_steps"], (st["perturb_steps"], lit["perturb_steps"])
    assert total - capped * n <= st["perturb_steps"] <= total
    assert st["scaled_runs"] > 0 and st["la_steps"] > 0  # (this path's slot for the steps inside scaled runs)
