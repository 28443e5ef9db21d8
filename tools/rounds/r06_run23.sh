#!/bin/bash
# round 6: where the first frame of a view (natural tile mapping) spends more than the ordered one -- counters per DISPATCH of the
# AT pass and the frame's kernel, with the first frame split into the same two launches (FSMI355_AT_SPLIT_COLD=1)
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp FSMI355_AT_SPLIT_COLD=${SPLIT:-1}
O=gpurun_out/r06s
mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc$i -- python3 bench.py --workload c4_hdr64 --steps 3 --warmup 1 --no-cpu --no-cold --no-secondary --no-build > $O/pmc$i.log 2>&1
  python3 - $O/pmc$i <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_lav2_hdr64" in k or "k_at_pass64" in k:
            d = per.setdefault(int(row["Dispatch_Id"]), {"k": k[26:60] if "anonymous" in k else k[:14], "ms": (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6})
            d[row["Counter_Name"]] = "%.4g" % float(row["Counter_Value"])
    for i, d in per.items():
        print(i, d)
PY
done
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -size +1M -delete
