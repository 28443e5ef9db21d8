#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
timeout 900 python tools/bla_fast_check.py 2>&1 | tail -5
timeout 1200 python -m pytest tests -m gpu -x -q -k "bla or full_size or goldens" 2>&1 | tail -4
bash tools/rounds/ab_wl.sh r06aj c5_bla 4 product blanosplit
