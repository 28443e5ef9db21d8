#!/bin/bash
# round 6, final evidence: every workload profiled (kernel trace + the four counter passes), one bench line per configuration with its
# CPU leg, the emulated ranks with the read-back measured
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
bash tools/profile_round6.sh r06 > gpurun_out/profile_r06.log 2>&1
tail -n 2 gpurun_out/profile_r06.log
bash tools/bench_all.sh r06 2>&1 | tail -n 9
bash tools/rounds/r06_emulate.sh 2>&1 | tail -n 12
