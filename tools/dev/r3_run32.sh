#!/bin/bash
# attention: permlane-swap column reductions (in-tree) against ds_bpermute shuffles (build_ab/libdcamd_prev.so); tests; cfg5 and cfg2 steps
set -o pipefail
o=gpurun_out/r3aj; mkdir -p $o
root=$(pwd)
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" > $o/pytest.log 2>&1; rc=$?; tail -3 $o/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
for arm in prev new; do
  unset DCAMD_LIB
  case $arm in prev) export DCAMD_LIB=$root/build_ab/libdcamd_prev.so;; esac
  echo "== $arm"; timeout -k 10 200 python3 tools/bench_attention.py 200 2>&1 | grep -v amdgpu.ids
done
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in prev new; do
  unset DCAMD_LIB
  case $arm in prev) export DCAMD_LIB=$root/build_ab/libdcamd_prev.so;; esac
  timeout -k 10 400 python3 bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 --steps 3 --warmup 1 $common --breakdown $o/cfg5_$arm.bd.json > $o/cfg5_$arm.json 2> $o/cfg5_$arm.log; echo "cfg5 $arm rc=$?"
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("cfg5_prev","cfg5_new","cfg2_prev","cfg2_new"):
    d=json.load(open(f"gpurun_out/r3aj/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if n=="attention"})
PY
