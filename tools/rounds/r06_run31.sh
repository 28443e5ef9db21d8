#!/bin/bash
# round 6: does the vector memory path (the CU's one texture-address unit: TCP_TOTAL_CACHE_ACCESSES per CU cycle) set the pace of any other
# workload's kernel, as it did for the HDRFloat<double> LA loop before its records went through the scalar cache?
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06ad
mkdir -p $O
for wl in c3_lav2 c2_po c5_bla c4_scaled c4_2x32; do
  rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/$wl -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --no-cold --no-secondary --no-build > $O/$wl.log 2>&1
  python3 - $O/$wl $wl <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        d = per.setdefault(int(row["Dispatch_Id"]), {"k": k[:50], "ms": (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6})
        d[row["Counter_Name"]] = float(row["Counter_Value"])
    big = sorted(per.values(), key=lambda d: -d["ms"])[:3]
    for d in big:
        cu_cycles = d["ms"] * 1e-3 * 2.36e9 * 256
        print(sys.argv[2], d["k"], "ms %.2f" % d["ms"], "TCP accesses/CU-cycle %.3f" % (d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / cu_cycles),
              "VALU x4/1024 ms %.2f" % (d.get("SQ_INSTS_VALU", 0) * 4 / 1024 / 2.36e6), "vmem_rd %.3g" % d.get("SQ_INSTS_VMEM_RD", 0))
PY
done
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -size +1M -delete
