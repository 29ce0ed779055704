#!/bin/bash
set -o pipefail
o=gpurun_out/r3ab; mkdir -p $o
bash tools/dev/build_alt.sh prio -DDC_STG_PRIO > $o/build_prio.log 2>&1; echo "build prio rc=$?"
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in base prio nw8 nw8nogn base2 prio2 nw8_2 nw8nogn2 nogn; do
  unset DCAMD_LIB DCAMD_HALO_NW DCAMD_NO_GN_WS
  case $arm in prio*) export DCAMD_LIB=$(pwd)/gpurun_out/libdcamd_prio.so;; nw8nogn*) export DCAMD_HALO_NW=8 DCAMD_NO_GN_WS=1;; nw8*) export DCAMD_HALO_NW=8;; nogn) export DCAMD_NO_GN_WS=1;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_LIB DCAMD_HALO_NW DCAMD_NO_GN_WS
python3 - <<'PY'
import json
for f in ("base","prio","nw8","nw8nogn","base2","prio2","nw8_2","nw8nogn2","nogn"):
    d=json.load(open(f"gpurun_out/r3ab/cfg2_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3_" in n or "groupnorm"==n})
PY
