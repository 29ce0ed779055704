#!/bin/bash
set -o pipefail
o=gpurun_out/r3aa; mkdir -p $o
bash tools/dev/build_alt.sh winr -DDC_STG_W_IN_R > $o/build_winr.log 2>&1; echo "build winr rc=$?"
bash tools/dev/build_alt.sh wf -DDC_STG_WAIT_FIRST > $o/build_wf.log 2>&1; echo "build wf rc=$?"
bash tools/dev/build_alt.sh both -DDC_STG_WAIT_FIRST -DDC_STG_W_IN_R > $o/build_both.log 2>&1; echo "build both rc=$?"
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in base winr wf both base2 winr2 wf2 both2; do
  unset DCAMD_LIB
  case $arm in winr*) export DCAMD_LIB=$(pwd)/gpurun_out/libdcamd_winr.so;; wf*) export DCAMD_LIB=$(pwd)/gpurun_out/libdcamd_wf.so;; both*) export DCAMD_LIB=$(pwd)/gpurun_out/libdcamd_both.so;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("base","winr","wf","both","base2","winr2","wf2","both2"):
    d=json.load(open(f"gpurun_out/r3aa/cfg2_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3_halo<bf16,8w" in n})
PY
