#!/bin/bash
# round 6: the hand-written statements' rare exits.  (1) which statuses the parity cases reach (probe build); (2) the parity cases and the C4 frame with
# a build in which EVERY step leaves its statement with status 2 (FS_H64_ASM_TINY=1e300): the frames must not change; (3) the product
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
FSMI355_LIB=$PWD/build/ab/libfsmi355_h64dbg.so timeout 800 python tools/hdr64_statement_coverage.py | tail -n 1
echo "== status-2 build:"
FSMI355_LIB=$PWD/build/ab/libfsmi355_h64st2.so timeout 900 python -m pytest tests/test_gpu_hdr64_fast.py tests/test_gpu_goldens.py -x -q 2>&1 | tail -3
FSMI355_LIB=$PWD/build/ab/libfsmi355_h64st2.so timeout 600 python bench.py --workload c4_hdr64 --steps 3 --warmup 1 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('status-2 build C4: cold kernel', d['roofline']['kernel_ms'], 'warm', d.get('kernel_parts_ms_warm'), 'crc ok' if d.get('frame_crc32_equals_oracle_frame') else 'CRC MISMATCH')"
echo "== product:"
timeout 900 python -m pytest tests/test_gpu_hdr64_fast.py tests/test_gpu_goldens.py tests/test_gpu_pixel_order.py -x -q 2>&1 | tail -3
bash tools/rounds/ab_c4.sh r06ae product h64prev
