#!/bin/bash
# One bench.py line per BASELINE config on this GPU, each WITH its CPU leg (cpu_baseline + cpu_sample_rows_bit_exact: the sampled
# rows of the same frame checked against the oracle in the same run) -> gpurun_out/<tag>_all.jsonl
tag=${1:-rXX}
cd ${GRAFT_REPO_ROOT:-$(pwd)} || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_all.jsonl
: > $out
python3 bench.py --no-build >> $out 2>/dev/null
for wl in c1_direct c2_po c5_bla c4_hdr64 c4_2x32 c4_scaled; do
  python3 bench.py --no-build --workload $wl --steps 10 --warmup 1 --no-secondary >> $out 2>/dev/null
done
python3 - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    print(d["config"]["workload"], d["ms_per_step"], d["value"], d["unit"], d["roofline"]["kernel_ms"], d.get("frame_checksum"),
          "rows bit-exact:", d.get("cpu_sample_rows_bit_exact"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
