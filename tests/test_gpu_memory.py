"""GPU: memory behaviour of the host state machine (SURVEY.md section 8(b) "Async allocation ... ordering on the compute
stream is relied on"; Perturb.cuh:51-61 / GPU_LAReference.h:93-113: input tables fall back to page-locked host memory
when the device is out of memory).

The out-of-memory path is reached by fault injection: FSMI355_FAIL_INPUT_ALLOC=1 in the environment when fs_create runs
makes every input-table allocation of that renderer behave as if the device were full."""
import os

import numpy as np
import pytest

import _oracle
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_PO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR32, T_HDR64, _capi, inputs

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))


def _pairs(co):
    return [(float(c["m"]), int(c["e"])) for c in co]


@pytest.fixture()
def starved_renderer(native_libs, monkeypatch):
    assert GPURenderer.TestCudaIsWorking() != 0
    monkeypatch.setenv("FSMI355_FAIL_INPUT_ALLOC", "1")
    r = GPURenderer(0)
    monkeypatch.delenv("FSMI355_FAIL_INPUT_ALLOC")
    yield r
    r.close()


def _frame(r, n):
    out = r.new_iter_buffer()
    red = _capi.Reduction()
    assert r.RenderCurrent(n, out, None, red) == 0
    assert r.SyncComputeStream() == 0
    return out, red


def test_frames_render_from_pinned_host_memory_when_the_device_is_full(starved_renderer):
    r = starved_renderer
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    bla = inputs.BLATable(ob)
    n = v.num_iterations
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert r.host_fallback_bytes == 0  # frame buffers never fall back
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
    assert r.host_fallback_bytes >= ob.count * 16 + la.count * 68  # orbit (prepared + companions) and the LA table
    co = _pairs(v.coords_perturb(ob))
    for parity, st in ((PARITY_CPU, 0), (PARITY_CPU_GPUSTAGE, 1)):
        assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR32, Mode=LAV2_FULL, parity=parity) == 0
        out, red = _frame(r, n)
        assert np.array_equal(out, _oracle.lav2_hdr32(v, ob, la, stage_test=st)), parity
        assert red.Sum == int(out[:36, :64].astype(np.uint64).sum())
    # perturbation only (the scaled runs read the companion arrays through the scalar cache: host memory too)
    assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR32, Mode=LAV2_PO, parity=PARITY_CPU) == 0
    assert np.array_equal(_frame(r, n)[0], _oracle.bla_hdr32(v, ob, None))
    # BLA: orbit + all table levels in host memory
    before = r.host_fallback_bytes
    assert r.RenderPerturbBLA(None, ob, bla, None, None, *co, n) == 0
    assert np.array_equal(_frame(r, n)[0], GOLD["view5_bla_64x36"])
    assert r.host_fallback_bytes > before
    # tables built on the device land in (and are built from) host memory as well
    assert r.InitializePerturb(2, ob, 0, None, None) == 0
    assert r.BuildLAOnDevice(ob, host_fallback=False) == 0
    assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
    assert np.array_equal(_frame(r, n)[0], GOLD["view5_lav2_cpu_64x36"])
    assert r.BuildBLAOnDevice(ob) == 0
    assert r._lib.fs_render_bla(r._h, T_HDR32, v.coords_perturb(ob).ctypes.data, n) == 0
    assert np.array_equal(_frame(r, n)[0], GOLD["view5_bla_64x36"])


def test_compressed_orbit_and_hdr64_from_pinned_host_memory(starved_renderer):
    r = starved_renderer
    v = inputs.View.builtin(5, 64, 36)
    o = inputs.Orbit(v, is64=True, compression_exp=20)
    la = inputs.LATable(o)
    assert o.compressed
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, o, 0, None, la) == 0
    assert r.host_fallback_bytes > 0
    assert r.RenderPerturbLAv2(None, None, None, *_pairs(v.coords_perturb(o)), v.num_iterations, T=T_HDR64, Mode=LAV2_FULL,
                               parity=PARITY_CPU) == 0
    assert np.array_equal(_frame(r, v.num_iterations)[0], _oracle.lav2_hdr32(v, o, la, stage_test=0))


def test_a_normal_renderer_never_touches_the_fallback_and_reuses_its_table_memory(native_libs):
    """Two orbits of different lengths through the same renderer, each with a device-built LA and BLA table, twice over:
    results stay right while the arena, the LA buffers and the BLA block are reused (second round allocates nothing new
    that the first did not -- not observable from here, but a stale pointer or a short buffer would show in the frames)."""
    r = GPURenderer(0)
    try:
        for _ in range(2):
            for view_n in (5, 19, 5):
                v = inputs.View.builtin(view_n, 64, 36)
                ob = inputs.Orbit(v)
                assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, True) == 0
                assert r._lib.fs_upload_orbit(r._h, 0, T_HDR32, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
                assert r.BuildLAOnDevice(ob, host_fallback=False) == 0
                la = inputs.LATable(ob)
                co = _pairs(v.coords_perturb(ob))
                assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL,
                                           parity=PARITY_CPU_GPUSTAGE) == 0
                assert np.array_equal(_frame(r, v.num_iterations)[0], _oracle.lav2_hdr32(v, ob, la, stage_test=1)), view_n
                assert r.BuildBLAOnDevice(ob) == 0
                assert r._lib.fs_render_bla(r._h, T_HDR32, v.coords_perturb(ob).ctypes.data, v.num_iterations) == 0
                assert np.array_equal(_frame(r, v.num_iterations)[0], _oracle.bla_hdr32(v, ob, inputs.BLATable(ob))), view_n
        assert r.host_fallback_bytes == 0
    finally:
        r.close()


def test_second_upload_of_a_large_orbit_renders_the_same_frame(native_libs):
    """Round 3's regression: with the stream-ordered allocator (hipMallocAsync / hipFreeAsync) the blocks of the SECOND upload
    of a large orbit came back from the pool with stale contents under ROCm 7.2.0 (tools/microbench/async_alloc_probe.hip;
    DESIGN.md 3.2) and every pixel of the second frame was wrong -- whatever the kernel variant.  View 17's 1.5-million-entry
    orbit (25 MB prepared + 49 MB of companions + its LA table) uploaded three times into one renderer: the tuned, the
    no-scaled-runs and the literal kernel give one and the same frame, in LAv2 Full and perturbation only."""
    r = GPURenderer(0)
    try:
        v = inputs.View.builtin(17, 64, 36)
        ob = inputs.Orbit(v)
        assert ob.count > 1 << 20
        la = inputs.LATable(ob)
        co = _pairs(v.coords_perturb(ob))
        frames = {}
        for mode, n in ((LAV2_FULL, v.num_iterations), (LAV2_PO, min(v.num_iterations, 20000))):
            for k, variant in enumerate((1, 0, 2, 0)):
                assert r.set_kernel_variant(variant) == 0
                assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
                assert r.InitializePerturb(0, ob, 0, None, la) == 0  # generation 0: uploaded again every time
                assert r.ClearMemory() == 0
                assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR32, Mode=mode, parity=PARITY_CPU) == 0
                out, _ = _frame(r, n)
                frames.setdefault(mode, out.copy())
                assert np.array_equal(out, frames[mode]), (mode, k, variant)
        assert r.host_fallback_bytes == 0
    finally:
        r.set_kernel_variant(0)
        r.close()


def test_idle_blocks_of_every_renderer_on_the_device_can_be_taken_back(native_libs):
    """A renderer keeps the blocks it frees for its next request (fs_idle_device_bytes).  What the out-of-memory path of ANY
    renderer on the device does before it gives up -- free the idle blocks of all of them (fs_release_idle_device_memory) --
    is run here by hand between two frames of two renderers: the idle bytes go to zero, both renderers keep rendering the
    right frame (nothing that is in use was touched), and the blocks come back with the next round of uploads."""
    a, b = GPURenderer(0), GPURenderer(0)
    try:
        v = inputs.View.builtin(5, 64, 36)
        ob = inputs.Orbit(v)
        la = inputs.LATable(ob)
        co = _pairs(v.coords_perturb(ob))
        want = _oracle.lav2_hdr32(v, ob, la, stage_test=0)

        def frame(r, gen):
            assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, True) == 0
            assert r.InitializePerturb(gen, ob, 0, None, la) == 0
            assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL,
                                       parity=PARITY_CPU) == 0
            return _frame(r, v.num_iterations)[0]

        for gen in (1, 2, 3):  # new generations: orbit and table are uploaded again, the old buffers are freed -> kept idle
            assert np.array_equal(frame(a, gen), want)
            assert np.array_equal(frame(b, gen), want)
        idle = a.idle_device_bytes() + b.idle_device_bytes()
        assert idle > 0
        assert GPURenderer.release_idle_device_memory(0) >= idle
        assert a.idle_device_bytes() == 0 and b.idle_device_bytes() == 0
        assert np.array_equal(frame(a, 3), want)  # (cached generation: renders from what is resident)
        assert np.array_equal(frame(b, 4), want)
        assert np.array_equal(frame(a, 5), want)
        assert a.host_fallback_bytes == 0 and b.host_fallback_bytes == 0
    finally:
        a.close()
        b.close()
