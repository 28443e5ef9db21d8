#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hdr64_fast.py tests/test_gpu_goldens.py tests/test_gpu_pixel_order.py -x -q 2>&1 | tail -5
bash tools/rounds/ab_c4.sh r06x product ${AB_OTHER:-h64sc1}
