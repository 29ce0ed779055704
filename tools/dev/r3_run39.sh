#!/bin/bash
# cfg2: the short-K 1-tap GEMMs on the 128x128 tile (two workgroups per CU, DCAMD_PIPE_NO_WIDE=1) against the 256x256 8-phase tile, per op
set -o pipefail
o=gpurun_out/r3aq; mkdir -p $o
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar"
for arm in wide nowide wide2 nowide2; do
  unset DCAMD_PIPE_NO_WIDE
  case $arm in nowide*) export DCAMD_PIPE_NO_WIDE=1;; esac
  timeout -k 10 300 python3 bench.py $common --steps 5 --warmup 2 --breakdown $o/cfg2_$arm.bd.json > $o/cfg2_$arm.json 2> $o/cfg2_$arm.log; echo "cfg2 $arm rc=$?"
done
unset DCAMD_PIPE_NO_WIDE
python3 - <<'PY'
import json
A=json.load(open("gpurun_out/r3aq/cfg2_wide.bd.json"))["ops"]; B=json.load(open("gpurun_out/r3aq/cfg2_nowide.bd.json"))["ops"]
A2=json.load(open("gpurun_out/r3aq/cfg2_wide2.bd.json"))["ops"]; B2=json.load(open("gpurun_out/r3aq/cfg2_nowide2.bd.json"))["ops"]
import collections
agg=collections.defaultdict(lambda:[0,0,0,0,0])
for a,b,a2,b2 in zip(A,B,A2,B2):
    if "wide8" in a["family"]:
        key=a["name"].split(".")[-1]+("@"+a["name"].split(".")[0]+"."+a["name"].split(".")[1] if True else "")
        g=agg[key]; g[0]+=a["ms"]; g[1]+=b["ms"]; g[2]+=a2["ms"]; g[3]+=b2["ms"]; g[4]+=1
for k,g in sorted(agg.items()): print(f"{k:40s} n={g[4]} wide {g[0]:.3f} {g[2]:.3f}  128x128 {g[1]:.3f} {g[3]:.3f}")
for f in ("wide","nowide","wide2","nowide2"):
    d=json.load(open(f"gpurun_out/r3aq/cfg2_{f}.json")); print(f,d["value"],d["ms_per_step"])
PY
