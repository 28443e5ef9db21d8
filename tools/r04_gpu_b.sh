#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_group.py tests/test_gpu_bench.py tests/test_gpu_tile_order.py -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu > $O/bench_n1.json 2> $O/bench_n1.err
FS_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary > $O/bench_forcedist.json 2> $O/bench_forcedist.err
timeout 600 python tools/group_pipeline_probe.py --members 8 > $O/group_pipeline_8.json 2> $O/group_pipeline_8.err
timeout 600 python tools/group_pipeline_probe.py --members 1 > $O/group_pipeline_1.json 2> $O/group_pipeline_1.err
tail -n 5 $O/pytest.txt; cat $O/bench_n1.json $O/bench_forcedist.json $O/group_pipeline_8.json $O/group_pipeline_1.json; tail -n 5 $O/group*.err
