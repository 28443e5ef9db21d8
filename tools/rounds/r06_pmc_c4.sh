#!/bin/bash
# round 6: what the HDRFloat<double> frame kernel waits for -- instruction and wait counters of the timed launches
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06k
mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_WAIT_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc$i -- python3 bench.py --workload c4_hdr64 --steps 3 --warmup 1 --no-cpu --no-cold --no-secondary --no-build > $O/pmc$i.log 2>&1
  python3 - $O/pmc$i <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        if "k_lav2_hdr64" in k or "k_at_pass64" in k:
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k in acc:
        print(k, "launches", len(n[k]), {c: "%.4g" % (v / len(n[k])) for c, v in acc[k].items()})
PY
done
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -size +1M -delete
