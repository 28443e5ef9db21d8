#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp FSMI355_AT_SPLIT_COLD=1
O=gpurun_out/r06z
mkdir -p $O
for v in product h64sc1; do
  if [ $v != product ]; then export FSMI355_LIB=$PWD/build/ab/libfsmi355_$v.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $O/$v -- python3 tools/c4_phase_probe.py > $O/$v.log 2>&1
  echo == $v; grep "^mode" $O/$v.log
  python3 - $O/$v <<'PY'
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
