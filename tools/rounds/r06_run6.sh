#!/bin/bash
# round 6, GPU call 6: same-box A/B of the two add forms (boxes differ by ~3 %), cndmask microbench
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06f
mkdir -p $O
B="timeout 600 python bench.py --workload c4_hdr64 --steps 10 --warmup 1 --no-cpu"
for rep in 1 2; do
  $B > $O/c4_addv3_$rep.json 2> $O/c4_addv3_$rep.err
  FSMI355_LIB=$PWD/build/ab/libfsmi355_h64addv2.so $B > $O/c4_addv2_$rep.json 2> $O/c4_addv2_$rep.err
done
./tools/microbench/valu_rates_f64 > $O/valu_rates_f64.jsonl 2>&1
for f in $O/c4_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print({k: d.get(k) for k in ("ms_per_step", "kernel_ms_warm", "kernel_parts_ms_warm", "frame_crc32_equals_oracle_frame")}, "cold kernel", d["roofline"].get("kernel_ms"))
except Exception as e:
    print("unreadable:", e)
PY
done
cat $O/valu_rates_f64.jsonl | grep -i "cndmask\|addc"
