#!/bin/bash
# timing-only: conv3_ws with one barrier per tap row (racy) against the shipped one barrier per tap — is the barrier count worth a restructure?
set -o pipefail
o=gpurun_out/r3ak; mkdir -p $o
ABLS=0,8,24,33 timeout -k 10 400 python3 tools/stamp_ws.py > $o/stamp_shipped.log 2>&1; echo "shipped rc=$?"
DEFS=DC_WS_ROWBAR_ABL ABLS=0,8,24,33 timeout -k 10 400 python3 tools/stamp_ws.py > $o/stamp_rowbar.log 2>&1; echo "rowbar rc=$?"
grep -E "ablation|team" $o/stamp_shipped.log; echo =====; grep -E "ablation|team" $o/stamp_rowbar.log
