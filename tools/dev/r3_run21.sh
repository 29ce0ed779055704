#!/bin/bash
set -o pipefail
o=gpurun_out/r3z; mkdir -p $o
bash tools/dev/build_alt.sh xinr -DDC_STG_X_IN_R > $o/build.log 2>&1; echo "build rc=$?"
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in base xinr base2 xinr2; do
  unset DCAMD_LIB
  case $arm in xinr*) export DCAMD_LIB=$(pwd)/gpurun_out/libdcamd_xinr.so;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("cfg2_base","cfg2_xinr","cfg2_base2","cfg2_xinr2"):
    d=json.load(open(f"gpurun_out/r3z/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3_halo" in n})
PY
