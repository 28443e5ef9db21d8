#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
bash tools/rounds/ab_c4.sh r06ai cap2k cap1k cap512
bash tools/rounds/ab_wl.sh r06ai c4_2x32 4 cap2k cap1k cap512
for v in cap2k cap1k cap512; do
  if [ $v = product ]; then unset FSMI355_LIB; else export FSMI355_LIB=$PWD/build/ab/libfsmi355_$v.so; fi
  timeout 600 python tools/emulate_ranks.py --workload c4_hdr64 --worlds 8 --tile-order cold --repeats 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$v', 'world', d['world'], 'cold slowest rank', d['slowest_rank_median_ms'])"
done
