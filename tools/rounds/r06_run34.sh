#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
bash tools/rounds/ab_wl.sh r06af c4_2x32 4 product stride
bash tools/rounds/ab_c4.sh r06af product stride
