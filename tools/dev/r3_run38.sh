#!/bin/bash
# timing-only: how much of igemm_xreg's GEGLU launches is the erf of the epilogue (build_ab/libdcamd_noerf.so: gelu = identity)
set -o pipefail
o=gpurun_out/r3ap; mkdir -p $o
root=$(pwd)
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in base noerf base2 noerf2; do
  unset DCAMD_LIB
  case $arm in noerf*) export DCAMD_LIB=$root/build_ab/libdcamd_noerf.so;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("base","noerf","base2","noerf2"):
    d=json.load(open(f"gpurun_out/r3ap/cfg2_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "xreg" in n})
PY
