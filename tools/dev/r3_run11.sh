#!/bin/bash
set -o pipefail
o=gpurun_out/r3k; mkdir -p $o
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wave_specialised or conv3x3" > $o/pytest_ws.log 2>&1; rc=$?; echo "pytest ws rc=$rc"; tail -6 $o/pytest_ws.log
[ $rc -ne 0 ] && exit 1
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in wr onetile unfused wrplain; do
  unset DCAMD_NO_GN_WS DCAMD_WS_ONE_TILE DCAMD_WS_PLAIN
  [ $arm = unfused ] && export DCAMD_NO_GN_WS=1
  [ $arm = onetile ] && export DCAMD_WS_ONE_TILE=1
  [ $arm = wrplain ] && export DCAMD_WS_PLAIN=1
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_NO_GN_WS DCAMD_WS_ONE_TILE DCAMD_WS_PLAIN
python3 - <<'PY'
import json
for f in ("cfg2_wr","cfg2_onetile","cfg2_unfused","cfg2_wrplain"):
    d=json.load(open(f"gpurun_out/r3k/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3" in n or "groupnorm" in n})
PY
GN=1 timeout -k 10 200 python3 tools/stamp_ws.py > $o/stamp_wr_gn.log 2>&1; grep -E "kernel:|team" $o/stamp_wr_gn.log
GN=0 timeout -k 10 200 python3 tools/stamp_ws.py > $o/stamp_wr_plain.log 2>&1; grep -E "kernel:|team" $o/stamp_wr_plain.log
