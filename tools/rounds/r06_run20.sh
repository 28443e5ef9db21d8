#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export FS_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r06p; mkdir -p $O
FSMI355_STALE_ORDER=0 timeout 600 python tools/c4_zoom_probe.py > $O/zoom_off.json 2> $O/zoom_off.err
FSMI355_STALE_REFRESH=0 timeout 600 python tools/c4_zoom_probe.py > $O/zoom_norefresh.json 2> $O/zoom_norefresh.err
timeout 600 python tools/c4_zoom_probe.py > $O/zoom_refresh8.json 2> $O/zoom_refresh8.err
FSMI355_STALE_REFRESH=4 timeout 600 python tools/c4_zoom_probe.py > $O/zoom_refresh4.json 2> $O/zoom_refresh4.err
FSMI355_STALE_REFRESH=0 timeout 600 python tools/c4_zoom_probe.py --zoom 0.9 --frames 8 > $O/zoom09_norefresh.json 2> $O/zoom09.err
for f in $O/zoom*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], d["env"], "mean kernel", d["zoom_mean_kernel_ms"], "mean wall", d["zoom_mean_wall_ms_incl_sorts"], "cold", d["last_view_cold"]["kernel_ms"], "warm", d["last_view_warm"]["kernel_ms"])
    print("   ", [f["kernel_ms"] for f in d["zoom_frames"]])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
