#!/bin/bash
# round 6, GPU call 5: k_lav2_hdr64 iteration 3 (arm-agreement votes, no select in the common arms), 8 waves with spills against 7 without
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hdr64_fast.py tests/test_gpu_pixel_order.py tests/test_gpu_goldens.py -x -q > $O/pytest_hdr64.log 2>&1; echo "pytest rc=$?" >> $O/pytest_hdr64.log
tail -4 $O/pytest_hdr64.log
B="timeout 600 python bench.py --workload c4_hdr64 --steps 10 --warmup 1"
$B > $O/c4_default.json 2> $O/c4_default.err
FSMI355_LIB=$PWD/build/ab/libfsmi355_h64w7.so $B --no-cpu > $O/c4_w7.json 2> $O/c4_w7.err
for f in $O/c4_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print({k: d.get(k) for k in ("value", "ms_per_step", "value_warm", "kernel_ms_warm", "kernel_parts_ms_warm", "frame_crc32_equals_oracle_frame", "cpu_sample_rows_bit_exact")})
    print("cold kernel", d["roofline"].get("kernel_ms"), d["roofline"].get("kernel_parts_ms"), "lat", {k: d["frame_timing"][k] for k in ("latency_ms_warm", "latency_kernel_ms_warm", "latency_ms_cold", "latency_kernel_ms_cold")})
except Exception as e:
    print("unreadable:", e)
PY
done
