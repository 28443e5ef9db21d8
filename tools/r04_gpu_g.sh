#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04l
mkdir -p $O
python3 -c 'from fractalshark_amd import _build; _build.build_all()' > $O/build.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-1300 $O/bench_n1.json
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_goldens.py tests/test_gpu_variants.py tests/test_gpu_tile_order.py tests/test_gpu_sweep.py -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -n 3 $O/pytest.txt
timeout 600 python tools/emulate_ranks.py --repeats 4 --tile-order warm > $O/emulate_ranks.jsonl 2>&1; cat $O/emulate_ranks.jsonl
