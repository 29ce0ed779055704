#!/bin/bash
# chunk-swizzled halo image: GPU tests, then A/B of the whole step against the previous library (build_ab/libdcamd_old.so)
set -o pipefail
o=gpurun_out/r3ac; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; rc=$?; tail -3 $o/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in old new old2 new2; do
  unset DCAMD_LIB
  case $arm in old*) export DCAMD_LIB=$(pwd)/build_ab/libdcamd_old.so;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("old","new","old2","new2"):
    d=json.load(open(f"gpurun_out/r3ac/cfg2_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3_" in n})
PY
