#!/usr/bin/env python3
"""gpurun_out/{prof,pmc}_<tag>_<workload>* (tools/profile_round5.sh) -> profiles/<tag>_*: per workload the rocprofv3 kernel
statistics (CSV, as written), the bench line of the same command, the dominant kernel's rows of every PMC pass, and ONE
profiles/<tag>_traffic.json with a key per workload (what bench.py's roofline.traffic reads).

  python tools/profile_round5_collect.py r05"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# the instantiation the TIMED frames launch (the step-counting launch of every run is another instantiation: kStats = true)
KERNEL = {"c3_lav2": "k_lav2_hdr32_fast<0, false, true, false, false", "c2_po": "k_perturb_scalar<float, false, false, false",
          "c5_bla": "k_bla_hdr32_fast", "c4_hdr64": "k_lav2_lit<double, 0, false",
          "c4_2x32": "k_lav2_2x32<0, false", "c4_scaled": "k_scaled_hdr32_fast<false>"}


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "profiles")
    for wl in bench.WORKLOADS:
        line = os.path.join(ROOT, "gpurun_out", "prof_%s_%s.json" % (tag, wl))
        if not os.path.exists(line) or not open(line).read().strip():
            print("no run for", wl)
            continue
        d = json.loads(open(line).read().strip().splitlines()[-1])
        shutil.copy(line, os.path.join(out, "%s_%s_under_rocprof_bench.json" % (tag, wl)))
        stats = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, wl), "**", "*kernel_stats.csv"), recursive=True)
        if stats:  # (gpurun merges every call's files into the same directory: the newest run is the one the bench line is from)
            shutil.copy(max(stats, key=os.path.getmtime), os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, wl)))
        key = d["config"]["workload"]
        alg = d["roofline"].get("algorithmic_bytes")
        if alg is None:  # compulsory traffic of a VALU-bound frame: inputs once + the iteration buffer once
            alg = int(d["config"].get("algorithmic_bytes", 0)) or None
        cmd = [sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "%s_%s" % (tag, wl), KERNEL[wl],
               "--out", "profiles/%s_traffic.json" % tag, "--key", key, "--largest-grid-only"]
        if alg:
            cmd += ["--algorithmic-bytes", str(alg)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(wl, key, d["roofline"]["kernel_ms"], "ms;", r.stdout.strip().splitlines()[-1] if r.returncode else "pmc ok")


if __name__ == "__main__":
    main()
