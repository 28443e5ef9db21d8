#!/bin/bash
# round 6, GPU call 1: the whole GPU suite, then the lines this round's first changes are judged by
# (C4 cold with the AT pass in every frame; C1; the read-back priced in the rank emulation).
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1
O=gpurun_out/r06a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --workload c4_hdr64 --steps 10 --warmup 1 > $O/bench_c4_hdr64.json 2> $O/bench_c4_hdr64.err
FSMI355_AT_SPLIT_COLD=0 timeout 600 python bench.py --workload c4_hdr64 --steps 10 --warmup 1 --no-cpu > $O/bench_c4_hdr64_r5cold.json 2> $O/bench_c4_hdr64_r5cold.err
timeout 300 python bench.py --workload c1_direct --steps 20 --warmup 2 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 600 python bench.py --steps 20 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 900 python tools/emulate_ranks.py --workload c4_hdr64 --worlds 8 --tile-order all --repeats 4 > $O/emulate_c4_hdr64.jsonl 2> $O/emulate_c4_hdr64.err
timeout 600 python tools/emulate_ranks.py --workload c3_lav2 --worlds 2,4,8 --tile-order warm --repeats 4 > $O/emulate_c3.jsonl 2> $O/emulate_c3.err
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print({k: d.get(k) for k in ("value", "ms_per_step", "value_cold", "value_warm", "kernel_ms_warm", "kernel_parts_ms_warm", "frame_crc32_equals_oracle_frame", "cpu_sample_rows_bit_exact")})
    print(d["roofline"].get("kernel_ms"), d["roofline"].get("kernel_parts_ms"), d["roofline"].get("frac"), d.get("frame_timing", {}))
except Exception as e:
    print("unreadable:", e)
PY
done
tail -3 $O/emulate_c4_hdr64.jsonl | cut -c1-1500
