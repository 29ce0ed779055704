#!/bin/bash
set -o pipefail
o=gpurun_out/r3i; mkdir -p $o
common="--no-parity --no-cpu-baseline --no-other-workloads --no-haar --steps 4 --warmup 2"
DCAMD_NO_GN_WS=1 DCAMD_HALO_NW=8 timeout -k 10 300 python3 bench.py $common --breakdown $o/nw8.bd.json > $o/nw8.json 2> $o/nw8.log; echo "nw8 rc=$?"
DCAMD_NO_GN_WS=1 timeout -k 10 300 python3 bench.py $common --breakdown $o/nw4.bd.json > $o/nw4.json 2> $o/nw4.log; echo "nw4 rc=$?"
python3 - <<'PY'
import json
A=json.load(open('gpurun_out/r3i/nw8.bd.json'))['ops']; B={o['name']:o for o in json.load(open('gpurun_out/r3i/nw4.bd.json'))['ops']}
for f in ("nw8","nw4"):
    d=json.load(open(f"gpurun_out/r3i/{f}.json")); print(f, d["value"], d["ms_per_step"], {n:(v["ms"],v["launches"],v["tflops"]) for n,v in d["kernels"].items() if "conv3" in n})
for o in A:
    b=B.get(o['name'])
    if b and 'conv3' in o['family'] and o['family']!=b['family']:
        print(f"{o['name']:36s} K={o['K']:5d} {b['family']:22s} {b['ms']:.3f} -> {o['family']:22s} {o['ms']:.3f}")
PY
