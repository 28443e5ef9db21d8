#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hdr64_fast.py tests/test_gpu_goldens.py -x -q 2>&1 | tail -3
bash tools/rounds/ab_c4.sh r06o product h64noz0
FSMI355_STATS_KEEP_ORDER=1 timeout 600 python tools/c4_arm_probe.py 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['frame'], {k: d[k] for k in d if k.startswith('pt_') or k.startswith('la_')})"
