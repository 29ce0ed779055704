#!/bin/bash
# stamps of conv3_ws: pipelined MFMA team (default) against the lock-step one, with timing-only ablations
set -o pipefail
o=gpurun_out/r3af; mkdir -p $o
ABLS=0,8,16,24,1,33 timeout -k 10 500 python3 tools/stamp_ws.py > $o/stamp_pipe_gn.log 2>&1; echo "pipe rc=$?"
DCAMD_WS_NO_PIPE=1 ABLS=0,8,16,24,1,33 timeout -k 10 300 python3 tools/stamp_ws.py > $o/stamp_nopipe_gn.log 2>&1; echo "nopipe rc=$?"
grep -E "ablation|team" $o/stamp_pipe_gn.log
echo ======
grep -E "ablation|team" $o/stamp_nopipe_gn.log
