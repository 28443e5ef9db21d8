#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
bash tools/rounds/ab_wl.sh r06ac c4_2x32 5 product h64prev
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
