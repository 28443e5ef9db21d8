#!/bin/bash
# first counter group only (instruction mix) for the workloads whose kernels changed late in the round
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp FS_NO_BUILD=1
mkdir -p gpurun_out
for wl in c4_hdr64 c4_scaled c2_po c4_2x32; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/pmc_r02_x_${wl} -- python3 bench.py --workload $wl --no-secondary --steps 1 --warmup 0 --no-cpu --no-build > gpurun_out/pmc_r02_x_${wl}.log 2>&1
done
find gpurun_out -name "*.db" -delete
ls gpurun_out/pmc_r02_x_*/*/ | head -30
