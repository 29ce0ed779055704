#!/bin/bash
# conv3_ws with the pipelined MFMA team + W tiles by the MFMA team (default) against the lock-step MFMA team (DCAMD_WS_NO_PIPE)
set -o pipefail
o=gpurun_out/r3ae; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -x -q > $o/pytest.log 2>&1; rc=$?; tail -3 $o/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in nopipe pipe nopipe2 pipe2; do
  unset DCAMD_WS_NO_PIPE
  case $arm in nopipe*) export DCAMD_WS_NO_PIPE=1;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_WS_NO_PIPE
python3 - <<'PY'
import json
for f in ("nopipe","pipe","nopipe2","pipe2"):
    d=json.load(open(f"gpurun_out/r3ae/cfg2_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3_" in n or n=="groupnorm"})
PY
