#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
bash tools/rounds/ab_wl.sh r06ak c3_lav2 10 product nocyc
timeout 1500 python -m pytest tests -m gpu -x -q -k "lav2 or goldens or full_size or hdr32 or parity or tile" 2>&1 | tail -4
