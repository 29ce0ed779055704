#!/bin/bash
# developer tool: run bench.py through its option matrix with short runs and report value / errors (catches paths the default run never takes)
out=gpurun_out/${1:-matrix}; mkdir -p $out
run() { name=$1; shift; python bench.py "$@" > $out/$name.json 2> $out/$name.err; rc=$?; v=$(python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], (d.get("parity") or {}).get("f32_max_rel_eps_mse"))
except Exception as e: print("NO JSON", e)
PY
); echo "$name rc=$rc $v"; [ $rc -ne 0 ] && tail -3 $out/$name.err; }
run cfg1 --workload small-unet-2x8
run cfg2_f32 --dtype f32 --steps 1 --warmup 1 --no-cpu-baseline
run cfg2_f16 --dtype f16 --steps 2 --warmup 1 --no-cpu-baseline
run cfg2_noshare --no-share-trunk --steps 1 --warmup 1 --no-parity --no-cpu-baseline
run cfg2_upl1000 --units-per-launch 1000 --steps 2 --warmup 1 --no-parity --no-cpu-baseline
run cfg2_ipg3 --images-per-gpu 3 --steps 2 --warmup 1 --no-parity --no-cpu-baseline
run cfg2_gb5 --global-batch 5 --steps 2 --warmup 1 --no-parity --no-cpu-baseline
run cfg2_3stage --stages 5:6,20:3,50:1 --steps 2 --warmup 1 --no-parity --no-cpu-baseline
run cfg3_f32 --workload chexpert256-dwt-unet-2x100 --dtype f32 --images-per-gpu 1 --steps 1 --warmup 0 --no-parity --no-cpu-baseline
