#!/bin/bash
set -o pipefail
o=gpurun_out/r3q; mkdir -p $o
GN=1 WRK=1 ABLS=0,8 timeout -k 10 400 python3 tools/stamp_ws.py > $o/abl_wr_gn.log 2>&1; echo "gn rc=$?"; grep -v amdgpu.ids $o/abl_wr_gn.log
GN=1 WRK=1 DEFS=DC_WR_CONTIG ABLS=0,8 timeout -k 10 400 python3 tools/stamp_ws.py > $o/abl_wrc_gn.log 2>&1; echo "gn contig rc=$?"; grep -v amdgpu.ids $o/abl_wrc_gn.log
GN=0 WRK=1 DEFS=DC_WR_CONTIG ABLS=0 timeout -k 10 300 python3 tools/stamp_ws.py > $o/abl_wrc_plain.log 2>&1; echo "plain contig rc=$?"; grep -v amdgpu.ids $o/abl_wrc_plain.log
