#!/bin/bash
# conv3_ws: which of the two prologue / issue changes costs time (prev = neither, NO_SPREAD = helper only, NO_HELP = spread only, in-tree = both)
set -o pipefail
o=gpurun_out/r3ai; mkdir -p $o
root=$(pwd)
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in prev NO_SPREAD NO_HELP both prev2 NO_SPREAD2 NO_HELP2 both2; do
  unset DCAMD_LIB
  case $arm in prev*) export DCAMD_LIB=$root/build_ab/libdcamd_prev.so;; NO_SPREAD*) export DCAMD_LIB=$root/build_ab/libdcamd_NO_SPREAD.so;; NO_HELP*) export DCAMD_LIB=$root/build_ab/libdcamd_NO_HELP.so;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("prev","NO_SPREAD","NO_HELP","both","prev2","NO_SPREAD2","NO_HELP2","both2"):
    d=json.load(open(f"gpurun_out/r3ai/cfg2_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3_ws" in n})
PY
