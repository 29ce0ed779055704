#!/bin/bash
# cfg5: the fp32 side-path GEMMs (M = units, 72 tiles of 256x128) on the 128x128 tile instead (DCAMD_PIPE_LIGHT_NK=1000)
set -o pipefail
o=gpurun_out/r3am; mkdir -p $o
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in base light base2 light2; do
  unset DCAMD_PIPE_LIGHT_NK
  case $arm in light*) export DCAMD_PIPE_LIGHT_NK=1000;; esac
  timeout -k 10 400 python3 bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 --steps 3 --warmup 1 $common --breakdown $o/cfg5_$arm.bd.json > $o/cfg5_$arm.json 2> $o/cfg5_$arm.log; echo "cfg5 $arm rc=$?"
done
unset DCAMD_PIPE_LIGHT_NK
python3 - <<'PY'
import json
for f in ("base","light","base2","light2"):
    d=json.load(open(f"gpurun_out/r3am/cfg5_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "igemm_pipe" in n})
PY
