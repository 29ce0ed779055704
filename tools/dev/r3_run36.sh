#!/bin/bash
set -o pipefail
o=gpurun_out/r3an; mkdir -p $o
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; rc=$?; tail -3 $o/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
timeout -k 10 400 python3 bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 --steps 3 --warmup 1 $common > $o/cfg5.json 2> $o/cfg5.log; echo "cfg5 rc=$?"
timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 > $o/cfg2.json 2> $o/cfg2.log; echo "cfg2 rc=$?"
timeout -k 10 300 python3 bench.py --workload chexpert256-dwt-unet-2x100 $common --steps 3 --warmup 1 > $o/cfg3.json 2> $o/cfg3.log; echo "cfg3 rc=$?"
python3 - <<'PY'
import json
for f in ("cfg5","cfg2","cfg3"):
    d=json.load(open(f"gpurun_out/r3an/{f}.json"))
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"]) for n,v in k.items() if "igemm_pipe" in n})
PY
