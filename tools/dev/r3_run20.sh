#!/bin/bash
set -o pipefail
o=gpurun_out/r3x; mkdir -p $o
HW=8 CI=512 CO=256 N=2040 ABLS=0,4,8 timeout -k 10 400 python3 tools/stamp_halo.py > $o/halo8_m0.log 2>&1; echo "m0 rc=$?"; grep -v amdgpu.ids $o/halo8_m0.log
NW8=1 HW=32 CI=128 CO=128 N=1020 ABLS=0,4,8 timeout -k 10 400 python3 tools/stamp_halo.py > $o/halo8_m1.log 2>&1; echo "m1 rc=$?"; grep -v amdgpu.ids $o/halo8_m1.log
NW8=1 DCAMD_HALO_NO_STAG=1 HW=32 CI=128 CO=128 N=1020 ABLS=0 timeout -k 10 400 python3 tools/stamp_halo.py > $o/halo8_m1_old.log 2>&1; echo "m1 old rc=$?"; grep -v amdgpu.ids $o/halo8_m1_old.log
HW=32 CI=128 CO=128 N=1020 ABLS=0 timeout -k 10 400 python3 tools/stamp_halo.py > $o/halo4_m1.log 2>&1; echo "4w rc=$?"; grep -v amdgpu.ids $o/halo4_m1.log
