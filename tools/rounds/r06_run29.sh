#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
FSMI355_LIB=$PWD/build/ab/libfsmi355_h64dbg.so FSMI355_STATS_KEEP_ORDER=1 python tools/c4_arm_probe.py | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['frame'], d['ordered'], {k:d[k] for k in d if 'statement' in k or k in ('pt_wave_steps','la_wave_steps')})"
timeout 900 python -m pytest tests/test_gpu_hdr64_fast.py tests/test_gpu_goldens.py tests/test_gpu_pixel_order.py -x -q 2>&1 | tail -5
bash tools/rounds/ab_c4.sh r06aa product h64la
