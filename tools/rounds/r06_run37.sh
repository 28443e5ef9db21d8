#!/bin/bash
# round 6, final evidence again after the last kernel changes (the two C4 LAv2 workloads profiled again; every workload's bench line; emulated ranks)
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
bash tools/profile_round6.sh r06 c4_hdr64 c4_2x32 > gpurun_out/profile_r06b.log 2>&1
tail -n 1 gpurun_out/profile_r06b.log
bash tools/bench_all.sh r06 2>&1 | tail -n 8
bash tools/rounds/r06_emulate.sh 2>&1 | tail -n 14
