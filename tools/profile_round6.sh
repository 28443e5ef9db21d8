#!/bin/bash
# Round-6 evidence for profiles/ (run on the GPU box through gpurun): for EVERY bench workload (every BASELINE configuration, C1 included)
#   * rocprofv3 --kernel-trace --stats of `bench.py --workload W` in the driver's form (kernel statistics + the bench line it printed),
#   * the four PMC passes of MI355X_MICROARCH.md "HBM / rocprofv3" -- one counter group per run, no trace domains.
# Everything lands under gpurun_out/; tools/profile_round6_collect.py (build container) turns it into profiles/r05_*.
#   bash tools/profile_round6.sh <tag> [workload ...]
tag=${1:-r06}; shift
wls=${@:-c3_lav2 c1_direct c2_po c5_bla c4_hdr64 c4_2x32 c4_scaled}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c 'from fractalshark_amd import _build; _build.build_all()' || exit 1
python3 -c 'import sys; sys.path.insert(0, "tests"); import _oracle; _oracle.build()' || exit 1
export FS_NO_BUILD=1
for wl in $wls; do
  extra="--steps 3 --warmup 1 --no-secondary"
  [ "$wl" = c3_lav2 ] && extra=""   # the headline: exactly the driver's command
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_${wl} -- python3 bench.py --no-build --workload $wl $extra > gpurun_out/prof_${tag}_${wl}.json 2> gpurun_out/prof_${tag}_${wl}.err
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_${wl}_${name} -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --no-cold --no-secondary --no-build > gpurun_out/pmc_${tag}_${wl}_${name}.log 2>&1
  done
done
find gpurun_out -name "*_kernel_trace.csv" -size +1M -delete
find gpurun_out -name "*.db" -delete
du -sh gpurun_out | tail -n 1
