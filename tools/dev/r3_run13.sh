#!/bin/bash
set -o pipefail
o=gpurun_out/r3q; mkdir -p $o
GN=1 ABLS=0,16,24,25,40,48,56 timeout -k 10 400 python3 tools/stamp_ws.py > $o/abl2_gn.log 2>&1; echo "gn rc=$?"; grep -v amdgpu.ids $o/abl2_gn.log
GN=0 WR=6 ABLS=0,1,15 timeout -k 10 300 python3 tools/stamp_ws.py > $o/abl2_plain_wr6.log 2>&1; echo "plain rc=$?"; grep -v amdgpu.ids $o/abl2_plain_wr6.log
