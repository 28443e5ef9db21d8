#!/bin/bash
# One round of rocprofv3 evidence for profiles/ (run on the GPU box through gpurun):
#   kernel-trace stats of the default bench command and of the secondary workloads, then the PMC passes
#   (tools/pmc_passes.sh: one counter group per run, no trace domains) for C3 and C5.
# Usage: tools/profile_round.sh <tag>      results: gpurun_out/prof_<tag>_*/
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
run_stats c5_bla --workload c5_bla --steps 3 --warmup 1 --no-cpu
run_stats c2_po --workload c2_po --steps 2 --warmup 1 --no-cpu
run_stats c4_scaled --workload c4_scaled --steps 2 --warmup 1 --no-cpu
run_stats c3_lds_orbit --variant lds_orbit --steps 5 --warmup 1 --no-cpu --no-secondary
run_stats c5_refill --workload c5_bla --variant refill --steps 3 --warmup 1 --no-cpu
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_render_current -- python3 tools/bench_render_current.py > gpurun_out/prof_${tag}_render_current.json 2> gpurun_out/prof_${tag}_render_current.err
bash tools/pmc_passes.sh ${tag}_c3 --no-secondary
bash tools/pmc_passes.sh ${tag}_c5 --workload c5_bla
bash tools/pmc_passes.sh ${tag}_c5_refill --workload c5_bla --variant refill
# keep only the small summaries
find gpurun_out -name "*_kernel_trace.csv" -size +2M -delete
find gpurun_out -name "*.db" -delete
du -sh gpurun_out | tail -1
