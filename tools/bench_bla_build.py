"""BLA table construction: device (fs_build_bla) vs the host builder, on the View 19 orbit (BASELINE config C5) and on a
synthetic orbit of the size the non-periodic View 19 orbit has (7.85 M entries: the real entries tiled).

Prints one JSON line.  Algorithmic bytes per build = orbit read (16 B/entry) + every level written once (44 B/record) +
every level but the last read once by the next merge; the device time is the HIP-event time around all launches of one
build (fs_last_kernel_ms)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, T_HDR32, inputs  # noqa: E402


def level_sizes(n_entries):
    m = n_entries - 1
    epl = []
    while m > 1:
        epl.append(m)
        m = (m + 1) >> 1
    epl.append(m)
    return epl


def algorithmic_bytes(n_entries):
    epl = level_sizes(n_entries)
    b = 16 * n_entries
    for l in range(2, len(epl)):
        b += 44 * epl[l]              # written once
        if l + 1 < len(epl):
            b += 44 * epl[l]          # read by the next merge
    return b


def main():
    v = inputs.View.builtin(19, 64, 36, antialiasing=1)
    ob = inputs.Orbit(v)
    t0 = time.perf_counter()
    host = inputs.BLATable(ob)
    host_s = time.perf_counter() - t0
    r = GPURenderer(0)
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
    out = {"orbit_entries": ob.count, "host_build_ms": round(host_s * 1e3, 3), "host_threads": 1}

    def device_build(entries_ptr, count, reps=5):
        assert r._lib.fs_upload_orbit(r._h, 0, T_HDR32, 4, entries_ptr, count, count, 0) == 0
        mr = ob.max_radius()
        ks, ws = [], []
        for _ in range(reps):
            t1 = time.perf_counter()
            assert r._lib.fs_build_bla(r._h, T_HDR32, mr.ctypes.data) == 0
            ws.append((time.perf_counter() - t1) * 1e3)
            ks.append(r.last_kernel_ms())
        return min(ks), min(ws)

    k, w = device_build(ob.data_ptr, ob.count)
    ab = algorithmic_bytes(ob.count)
    out.update({"device_kernels_ms": round(k, 4), "device_call_ms_incl_alloc": round(w, 3),
                "algorithmic_bytes": ab, "achieved_GBps": round(ab / (k * 1e-3) / 1e9, 1),
                "levels": r._lib.fs_bla_num_levels(r._h)})
    dev = r.read_bla_levels(False)
    out["bit_identical_to_host"] = all(np.array_equal(a, host.level(l)) for l, a in enumerate(dev))
    # synthetic: the real entries tiled to the size of the non-periodic View 19 orbit
    big = np.tile(ob.entries(), 19)[: 7_850_000]
    k2, w2 = device_build(big.ctypes.data, len(big), reps=3)
    ab2 = algorithmic_bytes(len(big))
    out["synthetic"] = {"orbit_entries": len(big), "device_kernels_ms": round(k2, 4),
                        "device_call_ms_incl_alloc": round(w2, 3), "algorithmic_bytes": ab2,
                        "achieved_GBps": round(ab2 / (k2 * 1e-3) / 1e9, 1), "hbm_peak_GBps": 8000,
                        "table_bytes": sum(44 * n for n in level_sizes(len(big))[2:])}
    print(json.dumps(out))
    r.close()


if __name__ == "__main__":
    main()
