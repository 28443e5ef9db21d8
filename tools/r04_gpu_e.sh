#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04h
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wide_positions.py tests/test_known_answers.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -n 30 $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_wide_counters.py tests/test_gpu_goldens.py -m gpu -x -q > $O/pytest2.txt 2>&1; echo "rc=$?" >> $O/pytest2.txt
tail -n 10 $O/pytest2.txt
