#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wide_positions.py tests/test_gpu_wide_counters.py tests/test_gpu_df32x2.py -m gpu -x -q --timeout 300 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -n 30 $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "plain or compress or 2x32 or uint64" --timeout 300 > $O/pytest2.txt 2>&1; echo "rc=$?" >> $O/pytest2.txt
tail -n 5 $O/pytest2.txt
