#!/bin/bash
# Round-4 evidence for profiles/ (run on the GPU box through gpurun): rocprofv3 kernel-trace stats of the DEFAULT bench command,
# and PMC passes (one counter group per run, no trace domains) for the headline workload.
#   bash tools/profile_round4.sh <tag> [pmc bench args...]
tag=${1:-r04}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c 'from fractalshark_amd import _build; _build.build_all()' || exit 1
python3 -c 'import sys; sys.path.insert(0, "tests"); import _oracle; _oracle.build()' || exit 1
export FS_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_default -- python3 bench.py --no-build > gpurun_out/prof_${tag}_default.json 2> gpurun_out/prof_${tag}_default.err
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-cold --no-secondary --no-build > gpurun_out/pmc_${tag}_${name}.log 2>&1
done
find gpurun_out -name "*_kernel_trace.csv" -size +1M -delete
find gpurun_out -name "*.db" -delete
du -sh gpurun_out | tail -n 1
