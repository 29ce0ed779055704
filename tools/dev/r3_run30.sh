#!/bin/bash
# conv3_ws: next chunk's halo pieces two per tap at taps 0-2 (in-tree) against all six at tap 0 (build_ab/libdcamd_prev.so)
set -o pipefail
o=gpurun_out/r3ah; mkdir -p $o
root=$(pwd)
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "ws or conv or halo or groupnorm" > $o/pytest.log 2>&1; rc=$?; tail -3 $o/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in prev new prev2 new2; do
  unset DCAMD_LIB
  case $arm in prev*) export DCAMD_LIB=$root/build_ab/libdcamd_prev.so;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
for arm in prev new; do
  unset DCAMD_LIB
  case $arm in prev*) export DCAMD_LIB=$root/build_ab/libdcamd_prev.so;; esac
  timeout -k 10 300 python3 bench.py --workload chexpert256-dwt-unet-2x100 $common --steps 3 --warmup 1 --breakdown $o/cfg3_$arm.bd.json > $o/cfg3_$arm.json 2> $o/cfg3_$arm.log; echo "cfg3 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("cfg2_prev","cfg2_new","cfg2_prev2","cfg2_new2","cfg3_prev","cfg3_new"):
    d=json.load(open(f"gpurun_out/r3ah/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3_ws" in n or n=="groupnorm" or "8w" in n})
PY
