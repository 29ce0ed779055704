#!/bin/bash
set -o pipefail
o=gpurun_out/r3e; mkdir -p $o
GN=1 timeout -k 10 300 python3 tools/stamp_ws.py > $o/stamp_ws_gn.log 2>&1; echo "stamp gn rc=$?"; tail -5 $o/stamp_ws_gn.log
GN=0 timeout -k 10 300 python3 tools/stamp_ws.py > $o/stamp_ws_plain.log 2>&1; echo "stamp plain rc=$?"; tail -5 $o/stamp_ws_plain.log
timeout -k 10 300 python3 tools/stamp_halo.py > $o/stamp_halo.log 2>&1; echo "stamp halo rc=$?"; tail -10 $o/stamp_halo.log
timeout -k 10 600 python3 tools/clock_probe.py --seconds 2.2 > $o/inkernel_clock.json 2> $o/inkernel_clock.log; echo "clock rc=$?"; cat $o/inkernel_clock.log | grep "^#"
