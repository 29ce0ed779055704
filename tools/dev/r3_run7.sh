#!/bin/bash
set -o pipefail
o=gpurun_out/r3g; mkdir -p $o
run() { tag=$1; shift; env "$@" timeout -k 10 200 python3 tools/stamp_ws.py > $o/stamp_$tag.log 2>&1; echo "== $tag rc=$?"; grep -E "kernel:|team" $o/stamp_$tag.log; }
run persist_gn_wr3 GN=1
run persist_plain_wr3 GN=0
run persist_plain_wr6 GN=0 WR=6
run onetile_plain_wr6 GN=0 WR=6 ONE_TILE=1
run onetile_plain_wr3 GN=0 ONE_TILE=1
