#!/bin/bash
# persistent igemm_wide8 (DCAMD_WIDE_PERSIST): bit-identity and rate against the one-tile form, then the cfg2 / cfg5 steps
set -o pipefail
o=gpurun_out/r3ar; mkdir -p $o
timeout -k 10 300 python3 tools/bench_wide_ab.py --persist --dtype bf16 --rounds 3 --reps 5 --shapes t8_qkv,t8_out,t8_ffo,t4_qkv,t4_ffo,dit_proj 2>&1 | grep -v amdgpu.ids | tee $o/ab_bf16.log
rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && exit $rc
grep -q "bit-identical=False" $o/ab_bf16.log && { echo "MISMATCH"; exit 9; }
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in base persist base2 persist2; do
  unset DCAMD_WIDE_PERSIST
  case $arm in persist*) export DCAMD_WIDE_PERSIST=1;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_WIDE_PERSIST
python3 - <<'PY'
import json
for f in ("base","persist","base2","persist2"):
    d=json.load(open(f"gpurun_out/r3ar/cfg2_{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"],v["gbps"]) for n,v in k.items() if "wide8" in n})
PY
