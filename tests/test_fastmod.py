"""CPU: the remainder shortcut of the palette lookup (fast_mod, fractalshark_amd/csrc/kernels_current.hip) restated in
numpy with the same binary32 / uint32 operations, checked against the exact remainder over its whole contract
(n < 2^32, 2^8 <= d < 2^24).  The GPU side is pinned by the colour tests (tests/test_gpu_goldens.py step 3)."""
import numpy as np
import pytest


def fast_mod(n, d):
    n = n.astype(np.uint32)
    inv = np.float32(1.0 / float(d))
    q = (n.astype(np.float32) * inv).astype(np.uint32)
    q = np.maximum(q, np.uint32(2)) - np.uint32(2)
    r = n - ((q.astype(np.uint64) * np.uint64(d)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    for k in (4, 2, 1):
        t = r - np.uint32(k * d)
        r = np.where(t < r, t, r)
    return r


@pytest.mark.parametrize("d", [256, 257, 1792, 7 << 8, 7 << 12, 65535, 65536, 1000003, (1 << 24) - 1])
def test_fast_mod_matches_exact_remainder(d):
    rng = np.random.default_rng(d)
    n = np.concatenate([rng.integers(0, 1 << 32, 2_000_000, dtype=np.uint64),
                        np.arange(0, 70000, dtype=np.uint64),
                        (1 << 32) - 1 - np.arange(0, 70000, dtype=np.uint64),
                        # around every multiple of d near the top and the bottom of the range
                        (np.arange(1, 4000, dtype=np.uint64) * d)[:, None].repeat(3, 1).ravel() +
                        np.tile(np.array([-1, 0, 1], np.int64), 3999).astype(np.uint64),
                        ((1 << 32) // d * d - np.arange(0, 3000, dtype=np.uint64) * d)])
    n = n[n < (1 << 32)].astype(np.uint32)
    with np.errstate(over="ignore"):
        got = fast_mod(n, d)
    assert np.array_equal(got, n % np.uint32(d))
