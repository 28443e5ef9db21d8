#!/bin/bash
# same-box A/B of libfsmi355 builds on one bench workload:  bash tools/rounds/ab_wl.sh <out tag> <workload> <steps> <variant> [...]
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/$1; WL=$2; ST=$3; shift 3
mkdir -p $O
B="timeout 900 python bench.py --workload $WL --steps $ST --warmup 1 --no-cpu --no-secondary"
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = product ]; then $B > $O/${WL}_${v}_$rep.json 2> $O/${WL}_${v}_$rep.err
    else FSMI355_LIB=$PWD/build/ab/libfsmi355_$v.so $B > $O/${WL}_${v}_$rep.json 2> $O/${WL}_${v}_$rep.err; fi
  done
done
for f in $O/${WL}_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "kernel_ms", d["roofline"].get("kernel_ms"), "warm", d.get("kernel_ms_warm"), "ms/frame", d["ms_per_step"], "crc ok" if d.get("frame_crc32_equals_oracle_frame") else "CRC %s" % d.get("frame_crc32_equals_oracle_frame"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
