#!/bin/bash
# A/B of the floor-test form on the GPU box: builds the library per variant (outside any profiler) and runs the bench
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04j
mkdir -p $O
for v in "0 24 44" "0 20 44" "0 28 44" "0 24 48" "1 20 56" "1 28 56"; do
  set -- $v
  export FS_FL_EVERY=$1 FS_FL_SHIFT=$2 FS_FL_FLOOR_EXP=$3
  python3 -c 'from fractalshark_amd import _build; _build.build_all()' > $O/build_$1_$2_$3.log 2>&1 || { echo "build failed $v"; tail -5 $O/build_$1_$2_$3.log; continue; }
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-cold > $O/bench_$1_$2_$3.json 2> $O/bench_$1_$2_$3.err
  python3 - <<PY
import json
d=json.load(open("$O/bench_$1_$2_$3.json"))
print("EVERY=$1 SHIFT=$2 FLOOR=2^-$3 kernel_ms", d["roofline"]["kernel_ms"], "ms/step", d["ms_per_step"], "runs", d["roofline"]["stat7_rank0"], "careful", d["roofline"]["careful_step_fraction_rank0"], "checksum ok", d["frame_checksum_equals_committed_n1"])
PY
done
