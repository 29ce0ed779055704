#!/bin/bash
# flash attention: softmax VALU reduction (+ occupancy 3) against the previous library; then the GPU tests and the cfg5 workload
set -o pipefail
o=gpurun_out/r3ag; mkdir -p $o
root=$(pwd)
for arm in old occ2 new; do
  unset DCAMD_LIB
  case $arm in old) export DCAMD_LIB=$root/build_ab/libdcamd_old.so;; occ2) export DCAMD_LIB=$root/build_ab/libdcamd_occ2.so;; esac
  echo "== $arm"; timeout -k 10 200 python3 tools/bench_attention.py 200 2>&1 | grep -v amdgpu.ids
done
unset DCAMD_LIB
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; rc=$?; tail -3 $o/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
for arm in old new; do
  unset DCAMD_LIB
  case $arm in old) export DCAMD_LIB=$root/build_ab/libdcamd_old.so;; esac
  timeout -k 10 400 python3 bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 --steps 3 --warmup 1 --no-parity --no-cpu-baseline --no-other-workloads --no-haar --breakdown $o/cfg5_$arm.bd.json > $o/cfg5_$arm.json 2> $o/cfg5_$arm.log; echo "cfg5 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("old","new"):
    d=json.load(open(f"gpurun_out/r3ag/cfg5_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if v["ms"]>1})
PY
