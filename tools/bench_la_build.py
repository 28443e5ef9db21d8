#!/usr/bin/env python3
"""LA table construction: device (fs_build_la) vs the host builder, per view.  The device number is the wall time of the
whole call (kernels + the per-stage read-backs of a few words), best of 3; the host number is libfsinputs' builder
(single-threaded = the table the device reproduces; and with the box's threads = the reference's multi-threaded stage 0).

  python tools/bench_la_build.py [--views 5 19 6 7]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, T_HDR32, T_HDR64, inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, nargs="*", default=[5, 19, 6, 7])
ap.add_argument("--is64", action="store_true")
args = ap.parse_args()
r = GPURenderer(0)
assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
for n in args.views:
    v = inputs.View.builtin(n, 64, 36, antialiasing=1)
    t0 = time.perf_counter()
    ob = inputs.Orbit(v, is64=args.is64)
    t_orbit = time.perf_counter() - t0
    t0 = time.perf_counter()
    la1 = inputs.LATable(ob, host_threads=1)
    t_host1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    la16 = inputs.LATable(ob, host_threads=16)
    t_host16 = time.perf_counter() - t0
    T = T_HDR64 if args.is64 else T_HDR32
    assert r._lib.fs_upload_orbit(r._h, 0, T, 4, ob.data_ptr, ob.count, ob.count, ob.period) == 0
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        assert r.BuildLAOnDevice(ob) == 0
        best = min(best, time.perf_counter() - t0)
    las, stages, at, use_at, _ = r.read_la(args.is64)
    same = las.tobytes() == la1.records().tobytes()
    best_mt = 1e9
    for _ in range(3):  # stage 0 as CreateLAFromOrbitMT makes it on a 16-thread host (fs_build_la_mt)
        t0 = time.perf_counter()
        assert r.BuildLAOnDevice(ob, host_threads=16) == 0
        best_mt = min(best_mt, time.perf_counter() - t0)
    same_mt = r.read_la(args.is64)[0].tobytes() == la16.records().tobytes()
    print(json.dumps({"view": n, "type": "hdr64" if args.is64 else "hdr32", "orbit_entries": ob.count,
                      "la_records": int(las.shape[0]), "stages": int(stages.shape[0]),
                      "device_build_ms": round(best * 1e3, 3), "host_build_1thread_ms": round(t_host1 * 1e3, 3),
                      "host_build_16thread_replay_ms": round(t_host16 * 1e3, 3),
                      "device_build_mt16_ms": round(best_mt * 1e3, 3), "bit_identical_to_host_16thread": bool(same_mt),
                      "gmp_orbit_s": round(t_orbit, 3), "bit_identical_to_host_1thread": bool(same),
                      "records_if_multithreaded_host": la16.count}), flush=True)
r.close()
