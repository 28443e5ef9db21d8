#!/bin/bash
# round 4, first GPU call: tile-order tests, the LAv2 parity tests, bench (N = 1, FS_FORCE_DIST), rank emulation
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile_order.py -x -q > $O/pytest_tile_order.txt 2>&1; echo "rc=$?" >> $O/pytest_tile_order.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_goldens.py tests/test_gpu_group.py -x -q > $O/pytest_parity.txt 2>&1; echo "rc=$?" >> $O/pytest_parity.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --steps 20 --warmup 5 --natural-tile-order --no-cpu --no-secondary > $O/bench_n1_natural.json 2> $O/bench_n1_natural.err
FS_FORCE_DIST=1 MASTER_PORT=29533 timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary > $O/bench_forcedist.json 2> $O/bench_forcedist.err
timeout 900 python tools/emulate_ranks.py --repeats 4 > $O/emulate_ranks.jsonl 2> $O/emulate_ranks.err
tail -3 $O/pytest_tile_order.txt $O/pytest_parity.txt; cat $O/bench_n1.json $O/bench_n1_natural.json $O/bench_forcedist.json $O/emulate_ranks.jsonl
