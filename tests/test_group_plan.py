"""CPU-only: the multi-GPU tiler behind the C ABI (fs_group_*, csrc/group.cpp).  Its plan (which rank owns which
rows, where each frame row lands in the gathered buffer) is a pure host function and must equal the Python tiler the
torchrun path of bench.py uses (fractalshark_amd/tiling.py, itself covered by the world-2 gloo test); and the RCCL entry
points the gather needs must be resolvable from the RCCL this image ships."""
import ctypes as C

import numpy as np
import pytest

from fractalshark_amd import _capi, tiling


@pytest.mark.parametrize("height", [36, 37, 64, 1080, 2160, 4320])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("band", [8, 24])
def test_group_plan_equals_python_tiler(native_libs, height, world, band):
    lib = _capi.render_lib()
    idx = np.zeros(height, np.uint32)
    mx = C.c_uint32(0)
    for rank in range(world):
        lr = C.c_uint32(0)
        lib.fs_group_plan(height, world, band, rank, C.byref(lr), C.byref(mx), idx.ctypes.data)
        assert lr.value == tiling.local_rows(height, rank, world, band)
    assert mx.value == tiling.max_local_rows(height, world, band)
    assert np.array_equal(idx.astype(np.int64), tiling.reassemble_index(height, world, band))
    # a permutation into distinct slots: no two frame rows share a gathered row
    assert len(np.unique(idx)) == height


def test_rccl_entry_points_resolve():
    """group.cpp resolves RCCL with dlopen at first use; the names it asks for must exist in the image's RCCL."""
    lib = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            lib = C.CDLL(name)
            break
        except OSError:
            continue
    if lib is None:
        pytest.skip("no RCCL on this host")
    for sym in ("ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv",
                "ncclGetErrorString"):
        assert hasattr(lib, sym), sym
