#!/bin/bash
set -o pipefail
o=gpurun_out/r3al; mkdir -p $o
timeout -k 10 500 python3 tools/dev/ws_turnover.py > $o/turnover_gn.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $o/turnover_gn.log | tail -8
