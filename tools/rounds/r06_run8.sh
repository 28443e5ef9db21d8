#!/bin/bash
# round 6, GPU call 8: LA loop with the step length one step ahead, same-box A/B
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06h
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hdr64_fast.py -x -q 2>&1 | tail -3
B="timeout 600 python bench.py --workload c4_hdr64 --steps 10 --warmup 1 --no-cpu"
for rep in 1 2; do
  $B > $O/c4_pipe_$rep.json 2> $O/c4_pipe_$rep.err
  FSMI355_LIB=$PWD/build/ab/libfsmi355_h64nopipe.so $B > $O/c4_nopipe_$rep.json 2> $O/c4_nopipe_$rep.err
done
for f in $O/c4_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print({k: d.get(k) for k in ("ms_per_step", "kernel_ms_warm", "kernel_parts_ms_warm", "frame_crc32_equals_oracle_frame")}, "cold kernel", d["roofline"].get("kernel_ms"))
except Exception as e:
    print("unreadable:", e)
PY
done
