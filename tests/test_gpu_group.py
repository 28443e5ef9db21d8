"""GPU: the multi-GPU tiler behind the C ABI (fs_group_*).  The test box has ONE GPU, so the members of the group share
device 0 and the gather runs over its peer-copy transport (the RCCL transport needs distinct devices; its entry points
are checked on the CPU side and the 8-GPU run is the driver's): N members render their interleaved bands concurrently,
the slices are gathered and re-ordered on the device, and the frame, its reduction and the padding must equal the
single-renderer frame."""
import numpy as np
import pytest

from fractalshark_amd import (GPURenderer, GPURendererGroup, LAV2_FULL, PARITY_CPU, PARITY_CPU_GPUSTAGE, _capi, inputs)

pytestmark = pytest.mark.gpu


def _pairs(co):
    return [(float(c["m"]), int(c["e"])) for c in co]


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("w,h", [(64, 36), (100, 75)])
def test_group_frame_equals_single_renderer(native_libs, world, w, h):
    v = inputs.View.builtin(5, w, h)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    n = v.num_iterations
    r = GPURenderer(0)
    assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
    assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, n, Mode=LAV2_FULL, parity=PARITY_CPU_GPUSTAGE) == 0
    ref = r.new_iter_buffer()
    rred = _capi.Reduction()
    assert r.RenderCurrent(n, ref, None, rred) == 0
    assert r.SyncComputeStream() == 0
    r.close()

    g = GPURendererGroup([0] * world)
    assert g.size == world and g.transport == 1  # shared device -> peer copies
    assert g.InitializeMemory(w, h, 1) == 0
    assert g.InitializePerturb(1, ob, la) == 0
    for parity in (PARITY_CPU_GPUSTAGE, PARITY_CPU_GPUSTAGE):  # twice: buffers are reused
        assert g.ClearMemory() == 0
        assert g.RenderPerturbLAv2(dx, dy, cx, cy, n, Mode=LAV2_FULL, parity=parity) == 0
        out = g.new_iter_buffer()
        red = _capi.Reduction()
        assert g.RenderCurrent(n, out, red) == 0
        assert g.Sync() == 0
        assert out.shape == ref.shape
        assert np.array_equal(out, ref)
        assert (red.Min, red.Max, red.Sum) == (rred.Min, rred.Max, rred.Sum)
    assert g.gather_ms() >= 0.0
    g.close()


def test_group_bla_path(native_libs):
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v)
    bla = inputs.BLATable(ob)
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))
    g = GPURendererGroup([0, 0, 0])
    assert g.InitializeMemory(64, 36, 1) == 0
    assert g.InitializePerturb(0, ob, None) == 0
    assert g.UploadBLA(bla) == 0
    assert g.RenderPerturbBLA(dx, dy, cx, cy, v.num_iterations) == 0
    out = g.new_iter_buffer()
    assert g.RenderCurrent(v.num_iterations, out) == 0
    assert g.Sync() == 0
    assert np.array_equal(out, gold["view5_bla_64x36"])
    g.close()


def _single_frames(v, ob, la, parities):
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    r = GPURenderer(0)
    assert r.InitializeMemory(v.width, v.height, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
    out = []
    for parity in parities:
        assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL, parity=parity) == 0
        buf = r.new_iter_buffer()
        assert r.RenderCurrent(v.num_iterations, buf) == 0
        assert r.SyncComputeStream() == 0
        out.append(buf)
    r.close()
    return out


@pytest.mark.parametrize("world", [2, 5])
def test_group_frames_issued_back_to_back_without_a_sync(native_libs, world):
    """fs_group_render_current is asynchronous: a host may issue frame N+1 before frame N has been gathered.  Two DIFFERENT
    frames (the two parity modes give different counts) are rendered and gathered back to back into separate host buffers
    with a single sync at the end; a sender that overwrote a gather slot the previous reassembly was still reading would mix
    them (the peer-copy transport orders the copy behind that reassembly with an event)."""
    v = inputs.View.builtin(5, 320, 180)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    order = [PARITY_CPU, PARITY_CPU_GPUSTAGE] * 3
    refs = _single_frames(v, ob, la, order[:2])
    assert not np.array_equal(refs[0], refs[1])
    g = GPURendererGroup([0] * world)
    assert g.InitializeMemory(320, 180, 1) == 0
    assert g.InitializePerturb(1, ob, la) == 0
    outs = [g.new_iter_buffer() for _ in order]
    for parity, out in zip(order, outs):
        assert g.RenderPerturbLAv2(dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL, parity=parity) == 0
        assert g.RenderCurrent(v.num_iterations, out) == 0
    assert g.Sync() == 0
    for k, out in enumerate(outs):
        assert np.array_equal(out, refs[k % 2]), k
    g.close()


@pytest.mark.parametrize("world", [2, 8])
def test_group_pipelined_loop_two_frames_in_flight(native_libs, world):
    """The pipelined host loop of include/fsmi355.h: render k; render_current k into host buffer k % 2; wait_current(1).
    When wait_current(1) returns, frame k-1 must be complete in ITS host buffer although frame k is still in flight; the
    frames alternate between two different results, and the members run longest-tiles-first from their second frame on."""
    w, h = 960, 544
    v = inputs.View.builtin(5, w, h)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    order = [PARITY_CPU, PARITY_CPU_GPUSTAGE] * 4
    refs = _single_frames(v, ob, la, order[:2])
    assert not np.array_equal(refs[0], refs[1])
    g = GPURendererGroup([0] * world)
    assert g.InitializeMemory(w, h, 1) == 0
    assert g.InitializePerturb(1, ob, la) == 0
    host = [g.new_iter_buffer() for _ in range(2)]
    reds = [_capi.Reduction(), _capi.Reduction()]
    assert g.WaitCurrent(0) == 0  # nothing posted yet: returns at once
    for k, parity in enumerate(order):
        assert g.RenderPerturbLAv2(dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL, parity=parity) == 0
        assert g.RenderCurrent(v.num_iterations, host[k % 2], reds[k % 2]) == 0
        assert g.WaitCurrent(1) == 0
        if k >= 1:
            got = host[(k - 1) % 2]
            assert np.array_equal(got, refs[(k - 1) % 2]), "frame %d was not complete when wait_current(1) returned" % (k - 1)
            assert reds[(k - 1) % 2].Sum == int(refs[(k - 1) % 2][:h, :w].astype(np.uint64).sum())
    assert g.WaitCurrent(0) == 0
    assert np.array_equal(host[(len(order) - 1) % 2], refs[(len(order) - 1) % 2])
    assert g.WaitCurrent(2) != 0  # only two frames are tracked
    if world == 2:  # (with 8 members a member's share of this frame is below the tile-order threshold)
        assert g.renderer(1).last_frame_tile_ordered()
    assert g.Sync() == 0
    g.close()


@pytest.mark.parametrize("transport", [0, 1])
def test_group_over_distinct_devices(native_libs, transport):
    """The group on real, distinct GPUs: transport 0 = RCCL (ncclCommInitAll, grouped ncclSend / ncclRecv on the members'
    compute streams over xGMI), transport 1 = hipMemcpyPeerAsync.  Needs >= 2 devices in this process; the round's test box
    has one, the driver's 8-GPU node has eight."""
    n = GPURenderer.device_count()
    if n < 2:
        pytest.skip("needs >= 2 HIP devices (this box has %d)" % n)
    world = min(n, 8)
    v = inputs.View.builtin(5, 320, 180)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    order = [PARITY_CPU, PARITY_CPU_GPUSTAGE, PARITY_CPU]
    refs = _single_frames(v, ob, la, order[:2])
    g = GPURendererGroup(list(range(world)), transport=transport)
    assert g.size == world
    assert g.transport == transport, "RCCL did not come up (ncclCommInitAll) -- the group fell back to peer copies"
    assert g.InitializeMemory(320, 180, 1) == 0
    assert g.InitializePerturb(1, ob, la) == 0
    outs = [g.new_iter_buffer() for _ in order]
    for parity, out in zip(order, outs):
        assert g.RenderPerturbLAv2(dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL, parity=parity) == 0
        assert g.RenderCurrent(v.num_iterations, out) == 0
    assert g.Sync() == 0
    for k, out in enumerate(outs):
        assert np.array_equal(out, refs[k % 2]), (transport, k)
    assert g.gather_ms() >= 0.0
    g.close()


# ---- RenderCurrent's colour half and its progressive form through the group (GPU_Render.cu:556-581, 1695-1805)
def _group_golden(world, name):
    import golden_cases as gc
    import _oracle
    from fractalshark_amd import T_F64, T_HDR32
    case = [c for c in gc.CASES if c[0] == name][0]
    _, view_n, alg, aa, crc64 = case
    v, ob, table = gc.build_inputs(inputs, view_n, alg, aa)
    pal = _oracle.default_palette(8)
    g = GPURendererGroup([0] * world)
    assert g.InitializeMemory(gc.W * aa, gc.H * aa, aa, pal, len(pal), 0, 1) == 0
    n = v.num_iterations
    if alg == "Cpu64":
        dx, dy, minx, maxy = v.coords_direct_f64(aa)

        def render():
            return g.Render(minx, maxy, dx, dy, n, T=T_F64)
    elif "V2" in alg:
        assert g.InitializePerturb(1, ob, table) == 0
        co = _pairs(v.coords_perturb(ob))

        def render():
            return g.RenderPerturbLAv2(*co, n, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU)
    else:
        assert g.InitializePerturb(0, ob, None) == 0
        assert g.UploadBLA(table) == 0
        co = _pairs(v.coords_perturb(ob))

        def render():
            return g.RenderPerturbBLA(*co, n)
    return g, render, v, aa, crc64, gc


@pytest.mark.parametrize("name", ["view5-cpu-bla-v2", "view5-cpu32-bla-hdr", "view0-cpu64-aa4"])
def test_group_colours_reproduce_the_reference_golden_crc(native_libs, name):
    """Three members row-tile a golden case; fs_group_render_current_colors returns the iteration buffer AND the Color16
    buffer of the whole frame (antialias + palette on device 0 behind the row order).  Both must encode, through the
    reference's own PNG writer, to the CRC-64 literal of FractalSharkTest/TestRenderGoldens.cpp:84-97 (AA 4: bands of 8 rows
    never split an antialiasing group)."""
    import json
    import os
    import _oracle
    g, render, v, aa, crc64, gc = _group_golden(3, name)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_crc.json")))[name]
    try:
        for _ in range(2):  # twice: the second frame goes through the other buffer set
            assert g.ClearMemory() == 0
            assert render() == 0
            it = g.new_iter_buffer()
            colors = g.new_color_buffer()
            assert len(colors) == gc.W * gc.H
            red = _capi.Reduction()
            assert g.RenderCurrent(v.num_iterations, it, red, color_buffer=colors) == 0
            assert g.Sync() == 0
            assert gc.buffer_crc32(it) == gold["iter_buffer_crc32"]
            assert red.Sum == gold["iter_sum"]
            assert _oracle.pin_lib() is not None
            assert _oracle.png_crc64(it, gc.W, gc.H, aa, v.num_iterations) == crc64
            assert _oracle.png_crc64_rgba16(colors.reshape(gc.H, gc.W, 4), gc.W, gc.H) == crc64
    finally:
        g.close()


def test_group_progressive_snapshot(native_libs):
    """progressive = true: a snapshot on the display streams that waits for no kernel and leaves the two frames in flight alone.
    (1) taken after the frame has finished it IS the frame, colours included; (2) taken while a long frame is being rendered it
    returns at once with a mixture of finished pixels and zeros, and the frame that was in flight is still delivered complete."""
    import _oracle
    g, render, v, aa, crc64, gc = _group_golden(3, "view5-cpu-bla-v2")
    try:
        assert g.ClearMemory() == 0
        assert render() == 0
        it = g.new_iter_buffer()
        assert g.RenderCurrent(v.num_iterations, it) == 0
        assert g.Sync() == 0
        # (1) a snapshot of the finished frame.  After RenderCurrent member 0 renders into the OTHER set: render the frame
        # again so that every member's current buffer holds it
        assert render() == 0
        assert g.Sync() == 0
        snap, colors, red = g.new_iter_buffer(), g.new_color_buffer(), _capi.Reduction()
        assert g.RenderCurrent(v.num_iterations, snap, red, color_buffer=colors, progressive=True) == 0
        assert g.SyncDisplay() == 0
        assert np.array_equal(snap, it)
        assert red.Sum == int(it[:gc.H, :gc.W].astype(np.uint64).sum())
        assert _oracle.png_crc64_rgba16(colors.reshape(gc.H, gc.W, 4), gc.W, gc.H) == crc64
        assert g.WaitCurrent(0) == 0  # the snapshot did not count as a posted frame: still the first RenderCurrent
    finally:
        g.close()
    # (2) in flight: a frame that takes long enough to be caught half way
    w, h = 1920, 1080
    v2 = inputs.View.builtin(5, w, h)
    ob = inputs.Orbit(v2)
    la = inputs.LATable(ob)
    co = _pairs(v2.coords_perturb_hdr32(ob))
    g = GPURendererGroup([0] * 3)
    try:
        assert g.InitializeMemory(w, h, 1) == 0
        assert g.InitializePerturb(1, ob, la) == 0
        assert g.ClearMemory() == 0
        assert g.Sync() == 0
        assert g.RenderPerturbLAv2(*co, v2.num_iterations, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
        full = g.new_iter_buffer()
        snap = g.new_iter_buffer()
        assert g.RenderCurrent(v2.num_iterations, snap, progressive=True) == 0
        assert g.SyncDisplay() == 0
        assert g.RenderCurrent(v2.num_iterations, full) == 0
        assert g.Sync() == 0
        done = snap[:h, :w] != 0
        assert (full[:h, :w] != 0).all()
        assert np.array_equal(snap[:h, :w][done], full[:h, :w][done])  # what the snapshot holds are final counts
        assert not done.all(), "the snapshot waited for the kernels"
    finally:
        g.close()


def test_group_reinitialised_after_an_odd_number_of_frames(native_libs):
    """A second InitializeMemory (a resize) after ONE RenderCurrent: the buffer-set rotation starts again at set 0, and
    fs_group_wait_current must wait on the set the frames of the new geometry actually go through (it used to count posted
    frames across the re-initialisation and wait on the other set's never-recorded event)."""
    v = inputs.View.builtin(5, 320, 180)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    co = _pairs(v.coords_perturb_hdr32(ob))
    refs = _single_frames(v, ob, la, [PARITY_CPU, PARITY_CPU_GPUSTAGE])
    g = GPURendererGroup([0] * 3)
    try:
        assert g.InitializeMemory(64, 36, 1) == 0
        v0 = inputs.View.builtin(5, 64, 36)
        ob0 = inputs.Orbit(v0)
        assert g.InitializePerturb(1, ob0, inputs.LATable(ob0)) == 0
        assert g.RenderPerturbLAv2(*_pairs(v0.coords_perturb_hdr32(ob0)), v0.num_iterations, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
        assert g.RenderCurrent(v0.num_iterations, g.new_iter_buffer()) == 0  # ONE frame: an odd count
        assert g.Sync() == 0
        assert g.InitializeMemory(320, 180, 1) == 0
        assert g.WaitCurrent(0) == 0  # nothing of the new geometry posted yet
        assert g.InitializePerturb(2, ob, la) == 0
        host = [g.new_iter_buffer() for _ in range(2)]
        order = [PARITY_CPU, PARITY_CPU_GPUSTAGE] * 3
        for k, parity in enumerate(order):
            assert g.RenderPerturbLAv2(*co, v.num_iterations, Mode=LAV2_FULL, parity=parity) == 0
            assert g.RenderCurrent(v.num_iterations, host[k % 2]) == 0
            assert g.WaitCurrent(1) == 0
            if k >= 1:
                assert np.array_equal(host[(k - 1) % 2], refs[(k - 1) % 2]), k - 1
        assert g.WaitCurrent(0) == 0
        assert np.array_equal(host[(len(order) - 1) % 2], refs[(len(order) - 1) % 2])
    finally:
        g.close()


# ---- round 6: the direct host path (every member copies its own bands to the host over its own link)

@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("w,h", [(64, 36), (100, 75), (320, 180)])
def test_direct_host_path_frame_equals_gather_path(native_libs, world, w, h):
    """fs_group_set_host_path(1): the frame every member copies band by band into the caller's buffer (one 2-D copy per member, a
    cut last band by itself) equals the gathered frame, padding included, with and without the reduction (which keeps its
    gather) -- heights that are and are not multiples of the band, worlds that leave some members without a band."""
    v = inputs.View.builtin(5, w, h)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    n = v.num_iterations
    ref, = _single_frames(v, ob, la, [PARITY_CPU_GPUSTAGE])
    g = GPURendererGroup([0] * world)
    assert g.InitializeMemory(w, h, 1) == 0
    assert g.InitializePerturb(1, ob, la) == 0
    assert g.SetHostPath(True) == 0
    for with_reduction in (False, True, False):
        assert g.RenderPerturbLAv2(dx, dy, cx, cy, n, Mode=LAV2_FULL, parity=PARITY_CPU_GPUSTAGE) == 0
        out = g.new_iter_buffer()
        out[:] = 0xDEADBEEF
        red = _capi.Reduction()
        assert g.RenderCurrent(n, out, red if with_reduction else None) == 0
        assert g.Sync() == 0
        assert np.array_equal(out[:h], ref[:h])
        if with_reduction:
            assert red.Sum == int(ref[:h, :w].astype(np.uint64).sum())
    g.close()


@pytest.mark.parametrize("world", [2, 8])
def test_direct_host_path_pipelined_two_frames_in_flight(native_libs, world):
    """The pipelined loop under the direct path: a member renders frame k+1 into its other slice while its copy of frame k is
    still leaving the first one; wait_current(1) returns when EVERY member's bands of frame k-1 are in that frame's host buffer.
    Alternating results, colours asked for on some frames (gather + direct copies side by side)."""
    w, h = 960, 544
    v = inputs.View.builtin(5, w, h)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    order = [PARITY_CPU, PARITY_CPU_GPUSTAGE] * 5
    refs = _single_frames(v, ob, la, order[:2])
    g = GPURendererGroup([0] * world)
    assert g.InitializeMemory(w, h, 1) == 0
    assert g.InitializePerturb(1, ob, la) == 0
    assert g.SetHostPath(True) == 0
    host = [g.new_iter_buffer() for _ in range(2)]
    reds = [_capi.Reduction(), _capi.Reduction()]
    for k, parity in enumerate(order):
        assert g.RenderPerturbLAv2(dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL, parity=parity) == 0
        host[k % 2][:] = 0
        assert g.RenderCurrent(v.num_iterations, host[k % 2], reds[k % 2] if k % 3 == 0 else None) == 0
        assert g.WaitCurrent(1) == 0
        if k >= 1:
            assert np.array_equal(host[(k - 1) % 2][:h], refs[(k - 1) % 2][:h]), k - 1
            if (k - 1) % 3 == 0:
                assert reds[(k - 1) % 2].Sum == int(refs[(k - 1) % 2][:h, :w].astype(np.uint64).sum())
    assert g.WaitCurrent(0) == 0
    assert np.array_equal(host[(len(order) - 1) % 2][:h], refs[(len(order) - 1) % 2][:h])
    assert g.Sync() == 0
    g.close()


@pytest.mark.parametrize("aa,world", [(1, 3), (3, 2)])
def test_copy_bands_to_host_of_one_renderer(native_libs, aa, world):
    """fs_copy_bands_to_host on plain renderers with row bands (what a rank process of bench.py --host-path direct calls): `world`
    renderers with interleaved bands (24-row bands for antialiasing 3) fill ONE host frame that equals the single-renderer frame."""
    from fractalshark_amd import tiling
    w, h = 96 * aa, 60 * aa
    v = inputs.View.builtin(5, w // aa, h // aa, antialiasing=aa)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    dx, dy, cx, cy = _pairs(v.coords_perturb_hdr32(ob))
    r = GPURenderer(0)
    assert r.InitializeMemory(w, h, aa, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(1, ob, 0, None, la) == 0
    assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
    ref = r.new_iter_buffer()
    assert r.RenderCurrent(v.num_iterations, ref) == 0
    assert r.SyncComputeStream() == 0
    whole = np.full_like(ref, 0xFFFFFFFF)
    assert r.CopyBandsToHost(whole.ctypes.data) == 0  # without bands: the plain copy of the whole buffer
    assert r.SyncComputeStream() == 0
    assert np.array_equal(whole, ref)
    band = tiling.band_height(aa)
    frame = np.full_like(ref, 0xFFFFFFFF)
    for k in range(world):
        assert r.SetRowBands(k * band, band, world * band) == 0
        assert r.RenderPerturbLAv2(None, None, None, dx, dy, cx, cy, v.num_iterations, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
        assert r.CopyBandsToHost(frame.ctypes.data) == 0
        assert r.SyncComputeStream() == 0
    assert np.array_equal(frame[:h], ref[:h])
    r.close()
