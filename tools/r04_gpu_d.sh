#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04e
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt
tail -n 6 $O/pytest_gpu.txt
