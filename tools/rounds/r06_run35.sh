#!/bin/bash
# round 6: a view's first frame with its tiles in the order of a sampled PerformAT count (kernels_tile_sample.hip) against the tile mapping's order
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06ag; mkdir -p $O
for rep in 1 2; do
  for wl in c4_hdr64 c4_2x32; do
    for sw in 1 0; do
      FSMI355_COLD_TILE_ORDER=$sw timeout 900 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu --no-secondary > $O/${wl}_order${sw}_$rep.json 2> $O/${wl}_order${sw}_$rep.err
    done
  done
done
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "cold kernel", d["roofline"].get("kernel_ms"), "ms/frame", d["ms_per_step"], "warm", d.get("kernel_ms_warm"), "crc ok" if d.get("frame_crc32_equals_oracle_frame") else "CRC MISMATCH")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
