#!/bin/bash
# 1 -> 8 GPU strong-scaling PROJECTION of the CheXpert-DWT UNet workload on ONE GPU: for a global batch B the
# time of the whole step on one GPU, and the time of rank 0's / rank 7's share of the 8-rank deal (bench.py --simulate-rank r/8:
# no process group, no all-gather).  Projected efficiency = t(1/1) / (8 * max_r t(r/8)).  Usage: project_scaling.sh OUTDIR
set -o pipefail
out=${1:-gpurun_out/scaling}
mkdir -p "$out"
wl=${WL:-chexpert256-dwt-unet-2x100}
common="--workload $wl --steps 3 --warmup 1 --no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for B in ${BATCHES:-2 8 16}; do
  python3 bench.py $common --global-batch $B > "$out/full_B$B.json" 2> "$out/full_B$B.log" || exit 1
  for r in 0 7; do
    python3 bench.py $common --global-batch $B --simulate-rank $r/8 > "$out/rank${r}of8_B$B.json" 2> "$out/rank${r}of8_B$B.log" || exit 1
  done
done
python3 - "$out" <<'PY'
import json, sys, os
out = sys.argv[1]
rows = []
for B in sorted({int(f.split("_B")[1].split(".")[0]) for f in os.listdir(out) if f.startswith("full_B") and f.endswith(".json")}):
    full = json.load(open(f"{out}/full_B{B}.json"))["ms_per_step"]
    rk = {r: json.load(open(f"{out}/rank{r}of8_B{B}.json"))["ms_per_step"] for r in (0, 7)}
    worst = max(rk.values())
    rows.append(dict(global_batch=B, ms_1gpu=full, ms_rank0_of8=rk[0], ms_rank7_of8=rk[7], projected_speedup_8gpu=round(full / worst, 2),
                     projected_efficiency=round(full / worst / 8, 3)))
json.dump(rows, open(f"{out}/projection.json", "w"), indent=1)
for r in rows:
    print(r)
PY
