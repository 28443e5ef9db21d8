#!/usr/bin/env python3
"""Whole BASELINE frames as the CPU ORACLE renders them -> tests/golden/frame_crcs.json (build container, CPU only, hours).

For every bench workload (bench.WORKLOADS: C1, C3 and its GPU-stage-test secondary, C2, C5, C4 in its three forms) this renders the
WHOLE frame with the oracle (tests/_oracle.workload_rows: the same dispatch bench.py's cpu_baseline leg uses, on the inputs
bench.make_inputs builds) and records, per `workload|parity|iteration cap`:

  crc32      zlib CRC-32 of the frame's valid region: rows 0..H-1, columns 0..W-1, uint32 little-endian, row-major
  sum        sum of the iteration counts of that region (what ReductionResults.Sum holds)
  band_rows  rows per band, and band_crc32 = the CRC-32 of each band of rows by itself (to localise a difference)
  source     "oracle" -- no GPU was involved in making these numbers

The pattern is the reference's own: FractalSharkTest/TestRenderGoldens.cpp:84-97 pins whole frames by a CRC of the image.
What renders them: oracle/cpu_ref.cpp (Fractal.cpp:2545-2678 LAv2, :2266-2470 BLA / single-step) pinned by the reference's
twelve golden CRC-64s, and -- for the two forms without a CPU twin (c4_2x32, c4_scaled) -- the restated CUDA kernels
(oracle/gpu_ref_2x32.cpp, cpu_ref.cpp's scaled kernel), which nothing in the reference pins ("parity unpinned").

Resumable: progress is kept in --state (default /tmp/fs_frame_crcs_state.json) band by band; rerun to continue.

  python tests/golden/make_frame_crcs.py [--only c5_bla,c3_lav2:cpu_gpustage,...] [--threads 7]
"""
import argparse
import json
import os
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OUT = os.path.join(HERE, "frame_crcs.json")
# cheapest first (so that an interrupted run has the most frames): (workload, parity)
JOBS = [("c1_direct", "cpu"), ("c5_bla", "cpu"), ("c3_lav2", "cpu_gpustage"), ("c4_scaled", "cpu"), ("c4_hdr64", "cpu_gpustage"),
        ("c3_lav2", "cpu"), ("c4_2x32", "cpu"), ("c2_po", "cpu")]
BANDS = 32


def band_rows_of(H):
    return ((H + BANDS - 1) // BANDS + 7) // 8 * 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--threads", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--state", default="/tmp/fs_frame_crcs_state.json")
    args = ap.parse_args()
    import numpy as np

    import _oracle
    import bench
    jobs = JOBS
    if args.only:
        want = [tuple((x.split(":") + [None])[:2]) for x in args.only.split(",") if x]
        jobs = [(w, p) for (w, p) in JOBS if (w, p) in want or (w, None) in want]
    try:
        table = json.load(open(OUT))
    except (OSError, ValueError):
        table = {}
    table["_comment"] = (
        "Whole frames of the bench workloads as the CPU ORACLE renders them (tests/golden/make_frame_crcs.py; no GPU involved). "
        "Key: workload|parity|iteration cap. crc32 = zlib CRC-32 of rows 0..H-1 x columns 0..W-1 of the iteration buffer, uint32 "
        "little-endian, row-major; sum = sum of those counts; band_crc32[i] = CRC-32 of rows [i*band_rows, (i+1)*band_rows) alone.")
    try:
        state = json.load(open(args.state))
    except (OSError, ValueError):
        state = {}
    _oracle.lib()
    for wl, parity in jobs:
        t0 = time.time()
        inp = bench.make_inputs(wl, parity=parity)
        W, H, n_iter = inp["W"], inp["H"], inp["n_iter"]
        key = "%s|%s|%d" % (inp["key"], parity, n_iter)
        if key in table and table[key].get("source") == "oracle":
            print("have", key, flush=True)
            continue
        br = band_rows_of(H)
        st = state.get(key) or {"next_row": 0, "crc": 0, "sum": 0, "bands": [], "seconds": 0.0}
        print("start %s at row %d of %d (inputs %.1f s)" % (key, st["next_row"], H, time.time() - t0), flush=True)
        while st["next_row"] < H:
            y0 = st["next_row"]
            y1 = min(H, y0 + br)
            t1 = time.time()
            buf = _oracle.workload_rows(inp, y0, y1, threads=args.threads)
            rows = np.ascontiguousarray(buf[y0:y1, :W]).astype("<u4", copy=False)
            del buf
            raw = rows.tobytes()
            st["crc"] = zlib.crc32(raw, st["crc"]) & 0xFFFFFFFF
            st["bands"].append(zlib.crc32(raw) & 0xFFFFFFFF)
            st["sum"] = int(st["sum"]) + int(rows.astype(np.uint64).sum())
            st["next_row"] = y1
            st["seconds"] += time.time() - t1
            state[key] = st
            with open(args.state + ".tmp", "w") as f:
                json.dump(state, f)
            os.replace(args.state + ".tmp", args.state)
            print("  %s rows %d..%d  %.1f s (total %.0f s)" % (wl, y0, y1, time.time() - t1, st["seconds"]), flush=True)
        table[key] = {"crc32": "%08x" % st["crc"], "sum": int(st["sum"]), "width": W, "height": H, "band_rows": br,
                      "band_crc32": ["%08x" % c for c in st["bands"]], "source": "oracle",
                      "oracle_function": ("direct_f64 (CalcCpuHDR<uint32_t,double,double>)" if inp.get("is_direct") else
                                          "gpu_lav2_2x32 (restated CUDA kernel, parity unpinned)" if inp["is2x32"] else
                                          "gpu_scaled_hdr32 (restated CUDA kernel, parity unpinned)" if inp["is_scaled"] else
                                          "lav2_hdr%d stage_test=%d" % (64 if inp["is64"] else 32, 0 if parity == "cpu" else 1)
                                          if inp["is_lav2"] else "bla_hdr32" + ("" if inp["bla"] is not None else " (no table)")),
                      "cpu_seconds_wall": round(st["seconds"], 1), "threads": args.threads}
        with open(OUT + ".tmp", "w") as f:
            json.dump(table, f, indent=1)
        os.replace(OUT + ".tmp", OUT)
        print("done", key, table[key]["crc32"], table[key]["sum"], "%.0f s" % st["seconds"], flush=True)
    # every job of the full list present: the tests stop tolerating a missing frame
    try:
        table = json.load(open(OUT))
    except (OSError, ValueError):
        table = {}
    want = set()
    for wl, parity in JOBS:
        inp = bench.make_inputs(wl, parity=parity)
        want.add("%s|%s|%d" % (inp["key"], parity, inp["n_iter"]))
    if want <= set(table):
        table["_complete"] = True
        with open(OUT + ".tmp", "w") as f:
            json.dump(table, f, indent=1)
        os.replace(OUT + ".tmp", OUT)
        print("all %d frames present" % len(want), flush=True)


if __name__ == "__main__":
    main()
