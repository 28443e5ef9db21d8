#!/bin/bash
# Kernel-trace stats (rocprofv3) of the default bench command and of the workloads whose kernels changed late in a round.
# Usage: tools/profile_round_small.sh <tag>      results: gpurun_out/prof_<tag>_*/
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c 'from fractalshark_amd import _build; _build.build_all()' || exit 1
python3 -c 'import sys; sys.path.insert(0, "tests"); import _oracle; _oracle.build()' || exit 1
export FS_NO_BUILD=1
run_stats() { # name, bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_${name} -- python3 bench.py "$@" --no-build > gpurun_out/prof_${tag}_${name}.json 2> gpurun_out/prof_${tag}_${name}.err
}
run_stats c3_default
run_stats c2_po --workload c2_po --steps 2 --warmup 1 --no-cpu
run_stats c4_scaled --workload c4_scaled --steps 2 --warmup 1 --no-cpu
run_stats c4_hdr64 --workload c4_hdr64 --steps 2 --warmup 1 --no-cpu
find gpurun_out -name "*_kernel_trace.csv" -size +2M -delete
find gpurun_out -name "*.db" -delete
du -sh gpurun_out | tail -n 1
