#!/bin/bash
set -o pipefail
o=gpurun_out/r3j; mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv or wave_specialised or upsample or thin" > $o/pytest_conv.log 2>&1; rc=$?; echo "pytest conv rc=$rc"; tail -4 $o/pytest_conv.log
[ $rc -ne 0 ] && exit 1
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in fused unfused persist; do
  unset DCAMD_NO_GN_WS DCAMD_WS_PERSIST
  [ $arm = unfused ] && export DCAMD_NO_GN_WS=1
  [ $arm = persist ] && export DCAMD_WS_PERSIST=1
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_NO_GN_WS DCAMD_WS_PERSIST
python3 - <<'PY'
import json
for f in ("cfg2_fused","cfg2_unfused","cfg2_persist"):
    d=json.load(open(f"gpurun_out/r3j/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3" in n or "groupnorm" in n})
PY
