#!/bin/bash
set -o pipefail
o=gpurun_out/r3q; mkdir -p $o
GN=1 ABLS=0,1,2,4,6,8,14,15 timeout -k 10 400 python3 tools/stamp_ws.py > $o/abl_gn.log 2>&1; echo "gn rc=$?"; grep -v amdgpu.ids $o/abl_gn.log
GN=0 ABLS=0,1,2,6,15 timeout -k 10 300 python3 tools/stamp_ws.py > $o/abl_plain.log 2>&1; echo "plain rc=$?"; grep -v amdgpu.ids $o/abl_plain.log
GN=1 RES=0 ABLS=0 timeout -k 10 300 python3 tools/stamp_ws.py > $o/abl_gn_nores.log 2>&1; echo "nores rc=$?"; grep -v amdgpu.ids $o/abl_gn_nores.log
