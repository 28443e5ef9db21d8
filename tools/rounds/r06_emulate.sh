#!/bin/bash
# round 6: per-rank kernel times of every workload at N = 1 / 8 (each rank's bands alone on the one GPU), read-back measured, the
# pipeline's steady state printed for both host paths
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06_emulate
mkdir -p $O
timeout 900 python tools/emulate_ranks.py --workload c3_lav2 --worlds 2,4,8 --tile-order warm --repeats 4 > $O/c3.jsonl 2> $O/c3.err
timeout 900 python tools/emulate_ranks.py --workload c4_hdr64 --worlds 8 --tile-order all --repeats 4 > $O/c4_hdr64.jsonl 2> $O/c4_hdr64.err
timeout 1200 python tools/emulate_ranks.py --workload c4_2x32 --worlds 8 --tile-order warm --repeats 3 > $O/c4_2x32.jsonl 2> $O/c4_2x32.err
timeout 900 python tools/emulate_ranks.py --workload c5_bla --worlds 8 --tile-order warm --repeats 3 > $O/c5.jsonl 2> $O/c5.err
for f in $O/*.jsonl; do python - "$f" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    print(d["workload"][:40], d["world"], d["tile_order"], "kernel", d["slowest_rank_median_ms"], "d2h whole", d["d2h_whole_frame_ms_measured"], "bands", max(d["d2h_one_ranks_bands_ms_measured"]), d["pipeline_steady_state_ms"])
PY
done
