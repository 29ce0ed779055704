#!/bin/bash
set -o pipefail
o=gpurun_out/r3h; mkdir -p $o
run() { tag=$1; shift; env "$@" timeout -k 10 200 python3 tools/stamp_ws.py > $o/stamp_$tag.log 2>&1; echo "== $tag rc=$?"; grep -E "kernel:|team" $o/stamp_$tag.log; }
run persist_plain GN=0
run persist_gn GN=1
