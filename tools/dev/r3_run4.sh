#!/bin/bash
set -o pipefail
o=gpurun_out/r3d; mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wave_specialised" > $o/pytest_ws.log 2>&1; rc=$?; echo "pytest ws rc=$rc"; tail -15 $o/pytest_ws.log
[ $rc -ne 0 ] && exit 1
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in fused unfused wsplain; do
  unset DCAMD_NO_GN_WS DCAMD_WS_PLAIN
  [ $arm = unfused ] && export DCAMD_NO_GN_WS=1
  [ $arm = wsplain ] && export DCAMD_NO_GN_WS=1 DCAMD_WS_PLAIN=1
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_NO_GN_WS DCAMD_WS_PLAIN
python3 - <<'PY'
import json
for f in ("cfg2_fused","cfg2_unfused","cfg2_wsplain"):
    d=json.load(open(f"gpurun_out/r3d/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3" in n or "groupnorm" in n})
PY
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee $o/pytest.rc
tail -6 $o/pytest.log
grep -E "^(FAILED|ERROR)" $o/pytest.log | head -20
