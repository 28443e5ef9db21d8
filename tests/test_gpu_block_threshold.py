"""The wave-uniform block threshold of the tuned HDRFloat<float> LAv2 loop (FS_BT_T in csrc/scaled_runs.hpp), evaluated on the device by
the loop's own macro and held against its definition:  T = -1 ("never") when the wave's largest max|dc| exceeds the block bound --
which a "never" bound, the most negative integer, always is -- and min(bound - largest scale shift, bits(2^14)) otherwise, without
wrap-around.  The first form of the macro replaced the "never" bound by -2^30 before subtracting: positive again for scale shifts
below -2^30 (|dz| < 2^-152), found by tools/block_bound_check.py on the deep views and by nothing else."""
import ctypes as C

import numpy as np
import pytest

from fractalshark_amd import GPURenderer

pytestmark = pytest.mark.gpu
H = 0x46800000
NEVER = -(1 << 31)


def model(bound, shift, dc):
    if dc > bound:
        return -1
    return min(bound - shift, H)


def test_block_threshold_corners_and_random(native_libs):
    rng = np.random.default_rng(5)
    shifts = [-254 << 23, -200 << 23, -129 << 23, -(1 << 30) - 1, -(1 << 30), -128 << 23, -127 << 23, -1 << 23, 0, 1 << 23,
              64 << 23, 127 << 23]
    bounds = [NEVER, 0, 1, 0x00800000, 0x33800000, 0x3E800000, 0x3F800000, H - 1, H, H + 1, 0x7F000000, 0x7F7FFFFF]
    dcs = [0, 1, 0x00800000, 0x33800000, 0x3F800000, 0x7F000000, 0x7F800000]
    cases = [(b, s, d) for b in bounds for s in shifts for d in dcs]
    for _ in range(4000):
        b = NEVER if rng.random() < 0.1 else int(rng.integers(0, 0x7F800000))
        s = int(rng.integers(-254, 128)) << 23
        d = int(rng.integers(0, 0x7F800001))
        cases.append((b, s, d))
    bw = np.array([c[0] for c in cases], dtype=np.int32)
    sh = np.array([c[1] for c in cases], dtype=np.int32)
    dc = np.array([c[2] for c in cases], dtype=np.int32)
    out = np.zeros(len(cases), dtype=np.int32)
    r = GPURenderer(0)
    rc = r._lib.fs_test_block_threshold(r._h, bw.ctypes.data_as(C.c_void_p), sh.ctypes.data_as(C.c_void_p),
                                        dc.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), len(cases))
    assert rc == 0
    want = np.array([model(*c) for c in cases], dtype=np.int64)
    bad = np.nonzero(out.astype(np.int64) != want)[0]
    assert bad.size == 0, [(cases[i], int(out[i]), int(want[i])) for i in bad[:8]]
    # the property the loop relies on: a "never" bound yields a threshold no bit pattern of max|w| (>= 0) can pass
    assert (out[bw == NEVER] < 0).all()
