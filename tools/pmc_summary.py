#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 output directories written by tools/profile_round4.sh / tools/pmc_passes.sh:

  python tools/pmc_summary.py <tag> <kernel name substring> [--out profiles/<tag>_traffic.json --key <workload key>]

reads gpurun_out/pmc_<tag>_*/**/*_counter_collection.csv (one counter group per directory) and, for the kernel whose name
contains the substring, prints the per-launch average of every counter; FETCH_SIZE / WRITE_SIZE are KiB (MI355X_MICROARCH.md
"HBM / rocprofv3": 1 unit = 1 KiB of L2 <-> fabric traffic; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, so
traffic_bytes = (2 x FETCH + WRITE) x 1024 is the upper estimate the bench line carries).  Also copies the rows of that kernel
into profiles/<tag>_<group>.csv (small, tracked)."""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("kernel")
    ap.add_argument("--out", default=None)
    ap.add_argument("--key", default=None)
    ap.add_argument("--algorithmic-bytes", type=int, default=None)
    ap.add_argument("--note", default="")
    ap.add_argument("--largest-grid-only", action="store_true",
                    help="keep only the launches with the largest Grid_Size among the matches (drops C2's tile-order probe launch, "
                         "the same kernel over one pixel per tile)")
    a = ap.parse_args()
    sums, counts = {}, {}
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_*" % a.tag))):
        if not os.path.isdir(d):
            continue
        group = os.path.basename(d)[len("pmc_%s_" % a.tag):]
        rows = []
        # (gpurun merges every call's files into the same directory: only the NEWEST run's counters are this kernel's)
        files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
        for f in files[-1:]:
            with open(f) as fh:
                rd = csv.DictReader(fh)
                for row in rd:
                    if a.kernel in row["Kernel_Name"]:
                        rows.append(row)
        if not rows:
            continue
        if a.largest_grid_only:
            big = max(int(row["Grid_Size"]) for row in rows)
            rows = [row for row in rows if int(row["Grid_Size"]) == big]
        # one row per (dispatch, counter); a dispatch's value may be split over several rows (dimensions): sum per dispatch
        per = {}
        for row in rows:
            k = (row["Counter_Name"], row["Dispatch_Id"])
            per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
        for (name, _), v in per.items():
            sums[name] = sums.get(name, 0.0) + v
            counts[name] = counts.get(name, 0) + 1
        keep = os.path.join(ROOT, "profiles", "%s_%s.csv" % (a.tag, group))
        with open(keep, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    if not sums:
        sys.exit("no rows for kernel %r under gpurun_out/pmc_%s_*" % (a.kernel, a.tag))
    avg = {k: sums[k] / counts[k] for k in sums}
    out = {"kernel_substring": a.kernel, "launches_averaged": counts}
    out.update({("%s_KiB" % k if k in ("FETCH_SIZE", "WRITE_SIZE") else k): v for k, v in avg.items()})
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        out["traffic_bytes"] = int((2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)
        out["traffic_bytes_as_reported"] = int((avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)
    if a.algorithmic_bytes:
        out["algorithmic_bytes"] = a.algorithmic_bytes
    if a.note:
        out["note"] = a.note
    print(json.dumps(out, indent=1))
    if a.out and a.key:
        path = os.path.join(ROOT, a.out)
        table = json.load(open(path)) if os.path.exists(path) else {}
        table[a.key] = out
        json.dump(table, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
