#!/bin/bash
# rocprofv3 PMC passes for one bench.py workload (separate runs per counter group, no trace domains; see
# MI355X_MICROARCH.md "HBM / rocprofv3").  Usage: tools/pmc_passes.sh <tag> <bench.py args...>
# Results land in gpurun_out/pmc_<tag>_<group>/ as CSV.
#
# The native libraries are built BEFORE the loop, outside the profiler, and bench.py runs with FS_NO_BUILD=1: a
# profiled process must never spawn a compiler (the profiler's preload has initialised the GPU; exec from such a
# process tree takes the box down).
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c 'from fractalshark_amd import _build; _build.build_all()' || exit 1
python3 -c 'import sys; sys.path.insert(0, "tests"); import _oracle; _oracle.build()' || exit 1
export FS_NO_BUILD=1
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py "$@" --steps 2 --warmup 0 --no-cpu --no-build > gpurun_out/pmc_${tag}_${name}.log 2>&1
done
