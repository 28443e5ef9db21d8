#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04d
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_goldens.py tests/test_gpu_tile_order.py tests/test_gpu_sweep.py -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 900 python tools/emulate_ranks.py --repeats 4 --tile-order warm > $O/emulate_ranks.jsonl 2> $O/emulate_ranks.err
tail -n 5 $O/pytest.txt; cat $O/bench_n1.json $O/emulate_ranks.jsonl
