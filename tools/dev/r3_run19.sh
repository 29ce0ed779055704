#!/bin/bash
set -o pipefail
o=gpurun_out/r3y; mkdir -p $o
timeout -k 10 400 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv3x3 or upsample or quad or halo or mosaic" > $o/pytest_conv.log 2>&1; rc=$?; echo "pytest conv rc=$rc"; tail -5 $o/pytest_conv.log
[ $rc -ne 0 ] && exit 1
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in stag old stag2 old2; do
  unset DCAMD_HALO_NO_STAG
  case $arm in old*) export DCAMD_HALO_NO_STAG=1;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_HALO_NO_STAG
python3 - <<'PY'
import json
for f in ("cfg2_stag","cfg2_old","cfg2_stag2","cfg2_old2"):
    d=json.load(open(f"gpurun_out/r3y/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3" in n})
PY
HW=8 CI=512 CO=256 N=2040 ABLS=0,4,8 timeout -k 10 400 python3 tools/stamp_halo.py > $o/halo8_stag.log 2>&1; echo "stag rc=$?"; grep -v amdgpu.ids $o/halo8_stag.log
