#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
bash tools/rounds/ab_wl.sh r06ah c5_bla 4 product stride2
bash tools/rounds/ab_wl.sh r06ah c4_scaled 3 product stride2
