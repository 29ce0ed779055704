#!/bin/bash
set -o pipefail
o=gpurun_out/r3f; mkdir -p $o
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wave_specialised" > $o/pytest_ws.log 2>&1; rc=$?; echo "pytest ws rc=$rc"; tail -15 $o/pytest_ws.log
[ $rc -ne 0 ] && exit 1
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in persist onetile unfused; do
  unset DCAMD_NO_GN_WS DCAMD_WS_ONE_TILE
  [ $arm = unfused ] && export DCAMD_NO_GN_WS=1
  [ $arm = onetile ] && export DCAMD_WS_ONE_TILE=1
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_NO_GN_WS DCAMD_WS_ONE_TILE
python3 - <<'PY'
import json
for f in ("cfg2_persist","cfg2_onetile","cfg2_unfused"):
    d=json.load(open(f"gpurun_out/r3f/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3" in n or "groupnorm" in n})
PY
timeout -k 10 600 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -q -x > $o/pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -5 $o/pytest_model.log
