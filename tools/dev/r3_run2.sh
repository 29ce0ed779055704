#!/bin/bash
# round-3 GPU session 2: full GPU suite (with the 8-phase GEMM as the default wide tile), A/B of the two wide loops, cfg5 + cfg2 benches
set -o pipefail
o=gpurun_out/r3b; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee $o/pytest.rc
tail -8 $o/pytest.log
grep -q "failed" $o/pytest.log && grep -E "^(FAILED|ERROR)" $o/pytest.log | head -20
timeout -k 10 300 python3 tools/bench_wide_ab.py --dtype f16 --shapes dit_qkv,dit_proj,dit_fc1,dit_fc2,sq4k,sq8k > $o/wide_ab_f16.log 2>&1; echo "ab rc=$?"
cat $o/wide_ab_f16.log
timeout -k 10 300 python3 tools/bench_wide_ab.py --dtype bf16 --shapes t8_qkv,t8_out,t8_ffo,t4_qkv,t4_ffo,sq4k > $o/wide_ab_bf16.log 2>&1; echo "ab rc=$?"
cat $o/wide_ab_bf16.log
common="--steps 3 --warmup 1 --no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in new old; do
  if [ $arm = old ]; then export DCAMD_WIDE_OLD=1; else unset DCAMD_WIDE_OLD; fi
  timeout -k 10 300 python3 bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 $common --breakdown $o/cfg5_$arm.bd.json > $o/cfg5_$arm.json 2> $o/cfg5_$arm.log; echo "cfg5 $arm rc=$?"
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_WIDE_OLD
python3 - <<'PY'
import json
for f in ("cfg5_new","cfg5_old","cfg2_new","cfg2_old"):
    d=json.load(open(f"gpurun_out/r3b/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["tflops"]) for n,v in k.items() if "256x256" in n or "attention" in n})
PY
