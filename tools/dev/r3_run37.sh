#!/bin/bash
# LayerNorm: adaLN modulation vectors staged in LDS (in-tree) against per-row L1 fetches (build_ab/libdcamd_prev.so)
set -o pipefail
o=gpurun_out/r3ao; mkdir -p $o
root=$(pwd)
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "layernorm or dit or DiT" > $o/pytest.log 2>&1; rc=$?; tail -3 $o/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in prev new prev2 new2; do
  unset DCAMD_LIB
  case $arm in prev*) export DCAMD_LIB=$root/build_ab/libdcamd_prev.so;; esac
  timeout -k 10 400 python3 bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 --steps 3 --warmup 1 $common > $o/cfg5_$arm.json 2> $o/cfg5_$arm.log; echo "cfg5 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("prev","new","prev2","new2"):
    d=json.load(open(f"gpurun_out/r3ao/cfg5_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["gbps"]) for n,v in k.items() if n=="layernorm"})
PY
