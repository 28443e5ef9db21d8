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
