#!/bin/bash
# same-box A/B of libfsmi355 builds on the C4 HDRFloat<double> line:  bash tools/rounds/ab_c4.sh <out tag> <variant> [<variant> ...]
# (variant = a name under build/ab/libfsmi355_<name>.so, or "product"); two interleaved repeats each
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p $O
B="timeout 600 python bench.py --workload c4_hdr64 --steps 10 --warmup 1 --no-cpu"
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = product ]; then $B > $O/c4_${v}_$rep.json 2> $O/c4_${v}_$rep.err
    else FSMI355_LIB=$PWD/build/ab/libfsmi355_$v.so $B > $O/c4_${v}_$rep.json 2> $O/c4_${v}_$rep.err; fi
  done
done
for f in $O/c4_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], d.get("kernel_parts_ms_warm"), "cold kernel", d["roofline"].get("kernel_ms"), "crc ok" if d.get("frame_crc32_equals_oracle_frame") else "CRC MISMATCH")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
