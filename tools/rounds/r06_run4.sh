#!/bin/bash
# round 6, GPU call 4: order key A/B for the HDRFloat<double> frames (counts against the frame kernel's own steps), cndmask rates
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06d
mkdir -p $O
B="timeout 600 python bench.py --workload c4_hdr64 --steps 10 --warmup 1 --no-cpu"
$B > $O/c4_key_count.json 2> $O/c4_key_count.err
FSMI355_C4_ORDER_KEY=cost $B > $O/c4_key_cost.json 2> $O/c4_key_cost.err
./tools/microbench/valu_rates_f64 > $O/valu_rates_f64.jsonl 2>&1
timeout 300 python -m pytest tests/test_gpu_pixel_order.py -x -q 2>&1 | tail -3
for f in $O/c4_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print({k: d.get(k) for k in ("value", "ms_per_step", "value_warm", "kernel_ms_warm", "kernel_parts_ms_warm", "frame_crc32_equals_oracle_frame")})
    print("cold kernel", d["roofline"].get("kernel_ms"), d["roofline"].get("kernel_parts_ms"), "lat", {k: d["frame_timing"][k] for k in ("latency_ms_warm", "latency_kernel_ms_warm", "latency_ms_cold", "latency_kernel_ms_cold")})
except Exception as e:
    print("unreadable:", e)
PY
done
cat $O/valu_rates_f64.jsonl
