#!/bin/bash
# round 6, GPU call 2: the new HDRFloat<double> kernel (k_lav2_hdr64) -- parity first, then A/B of the kernel and of the frame's own
# order, one rocprofv3 kernel trace of the default C4 line.
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06b
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_hdr64_fast.py tests/test_gpu_pixel_order.py tests/test_gpu_goldens.py -x -q > $O/pytest_hdr64.log 2>&1; echo "pytest rc=$?" >> $O/pytest_hdr64.log
tail -4 $O/pytest_hdr64.log
B="timeout 600 python bench.py --workload c4_hdr64 --steps 10 --warmup 1"
$B > $O/c4_default.json 2> $O/c4_default.err
FSMI355_HDR64_LITERAL=1 $B --no-cpu > $O/c4_literal_kernel.json 2> $O/c4_literal_kernel.err
FSMI355_C4_INFRAME_ORDER=0 $B --no-cpu > $O/c4_no_inframe.json 2> $O/c4_no_inframe.err
FSMI355_LIB=$PWD/build/ab/libfsmi355_h64add0.so $B --no-cpu > $O/c4_add0.json 2> $O/c4_add0.err
FSMI355_LIB=$PWD/build/ab/libfsmi355_h64add2.so $B --no-cpu > $O/c4_add2.json 2> $O/c4_add2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 bench.py --no-build --workload c4_hdr64 --steps 6 --warmup 1 --no-cpu > $O/prof_c4.json 2> $O/prof_c4.err
find $O -name "*_kernel_trace.csv" -size +1M -delete; find $O -name "*.db" -delete
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_hdr64_fast.py --deselect tests/test_gpu_pixel_order.py --deselect tests/test_gpu_goldens.py > $O/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rest.log
tail -4 $O/pytest_rest.log
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
cat $O/prof_c4/*/*kernel_stats.csv 2>/dev/null | cut -c1-200 | head -12
