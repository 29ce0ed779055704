#!/bin/bash
# W tiles fetched by the MFMA team of conv3_ws (in-tree build) against "by the loader team" (build_ab/libdcamd_wl.so); LDS bank-conflict
# counters of the swizzled halo image (wl) against the previous library (old)
set -o pipefail
o=gpurun_out/r3ad; mkdir -p $o
root=$(pwd)
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q > $o/pytest.log 2>&1; rc=$?; tail -3 $o/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in wl wm wl2 wm2; do
  unset DCAMD_LIB
  case $arm in wl*) export DCAMD_LIB=$root/build_ab/libdcamd_wl.so;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_LIB
python3 - <<'PY'
import json
for f in ("wl","wm","wl2","wm2"):
    d=json.load(open(f"gpurun_out/r3ad/cfg2_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in k.items() if "conv3_" in n or n=="groupnorm"})
PY
export TMPDIR=/tmp
cd /tmp
for arm in old wl; do
  export DCAMD_LIB=$root/build_ab/libdcamd_$arm.so
  timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $root/$o/pmc_lds_$arm -- python3 $root/bench.py --steps 1 --warmup 1 $common > $root/$o/pmc_lds_$arm.log 2>&1; echo "pmc $arm rc=$?"
done
unset DCAMD_LIB
cd $root
for arm in old wl; do echo "== $arm"; python3 tools/dev/lds_pmc_summary.py $o/pmc_lds_$arm | grep -E "conv3|igemm"; done
find $o -name "*.csv" -size +5M -delete
