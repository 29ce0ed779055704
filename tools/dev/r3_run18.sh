#!/bin/bash
set -o pipefail
o=gpurun_out/r3v; mkdir -p $o
HW=8 CI=512 CO=256 N=2040 ABLS=0,1,2,4,5,6,7 timeout -k 10 400 python3 tools/stamp_halo.py > $o/halo8_stag.log 2>&1; echo "stag rc=$?"; grep -v amdgpu.ids $o/halo8_stag.log
DCAMD_HALO_NO_STAG=1 HW=8 CI=512 CO=256 N=2040 ABLS=0 timeout -k 10 400 python3 tools/stamp_halo.py > $o/halo8_old.log 2>&1; echo "old rc=$?"; grep -v amdgpu.ids $o/halo8_old.log
