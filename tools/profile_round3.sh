#!/bin/bash
# Round-3 evidence for profiles/ (run on the GPU box through gpurun): one bench line per BASELINE config with its CPU leg,
# rocprofv3 kernel-trace stats of the default command and of C5, PMC passes (one counter group per run) for C3 and C5.
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c 'from fractalshark_amd import _build; _build.build_all()' || exit 1
python3 -c 'import sys; sys.path.insert(0, "tests"); import _oracle; _oracle.build()' || exit 1
export FS_NO_BUILD=1
bash tools/bench_all.sh ${tag}
run_stats() { # name, bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_${name} -- python3 bench.py "$@" --no-build > gpurun_out/prof_${tag}_${name}.json 2> gpurun_out/prof_${tag}_${name}.err
}
run_stats c3_default
run_stats c5_bla --workload c5_bla --steps 3 --warmup 1 --no-cpu
bash tools/pmc_passes.sh ${tag}_c3 --no-secondary > /dev/null 2>&1
bash tools/pmc_passes.sh ${tag}_c5 --workload c5_bla > /dev/null 2>&1
find gpurun_out -name "*_kernel_trace.csv" -size +1M -delete
find gpurun_out -name "*.db" -delete
du -sh gpurun_out | tail -n 1
