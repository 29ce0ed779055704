#!/bin/bash
set -o pipefail
o=gpurun_out/r3c; mkdir -p $o
timeout -k 10 300 python3 tools/bench_wide_ab.py --order --dtype f16 --shapes dit_qkv,dit_proj,dit_fc1,dit_fc2 > $o/order_ab_f16.log 2>&1; echo "ab rc=$?"
cat $o/order_ab_f16.log
timeout -k 10 300 python3 tools/bench_wide_ab.py --order --dtype bf16 --shapes t8_qkv,t8_ffo,t4_qkv,t4_ffo > $o/order_ab_bf16.log 2>&1; echo "ab rc=$?"
cat $o/order_ab_bf16.log
common="--steps 3 --warmup 1 --no-parity --no-cpu-baseline --no-other-workloads --no-haar"
timeout -k 10 300 python3 bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 $common --breakdown $o/cfg5.bd.json > $o/cfg5.json 2> $o/cfg5.log; echo "cfg5 rc=$?"
timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2.bd.json > $o/cfg2.json 2> $o/cfg2.log; echo "cfg2 rc=$?"
DCAMD_NFAST_GEMM_BYTES=0 timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 > $o/cfg2_mfast.json 2> $o/cfg2_mfast.log; echo "cfg2 mfast rc=$?"
python3 - <<'PY'
import json
for f in ("cfg5","cfg2","cfg2_mfast"):
    d=json.load(open(f"gpurun_out/r3c/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["tflops"]) for n,v in k.items() if "256x256" in n or "attention" in n or "xreg" in n})
PY
