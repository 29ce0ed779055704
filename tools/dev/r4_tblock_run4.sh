set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
timeout -k 10 200 python tools/bench_tblock.py 8000 8 > gpurun_out/r4t/b5.log 2>&1; tail -1 gpurun_out/r4t/b5.log
B="--steps 4 --warmup 1 --no-other-workloads --no-cpu-baseline --no-haar"
timeout -k 10 300 python bench.py $B --breakdown gpurun_out/r4t/bd5.json > gpurun_out/r4t/bench5_all.log 2>&1; tail -c 400 gpurun_out/r4t/bench5_all.log | head -c 200; echo
DCAMD_NO_PO_FOLD=1 timeout -k 10 300 python bench.py $B > gpurun_out/r4t/bench5_nopo.log 2>&1
DCAMD_NO_PO_FOLD=1 DCAMD_NO_TBLOCK=1 timeout -k 10 300 python bench.py $B > gpurun_out/r4t/bench5_none.log 2>&1
timeout -k 10 300 python bench.py $B > gpurun_out/r4t/bench5_all2.log 2>&1
python - <<'PY'
import json
for f in ('bench5_all','bench5_nopo','bench5_none','bench5_all2'):
    l=[x for x in open(f'gpurun_out/r4t/{f}.log') if x.startswith('{')][-1]
    d=json.loads(l); print(f, d['value'], d['ms_per_step'], d['parity'].get('bf16_pred_rel_l2_vs_fp32_oracle'), d['parity'].get('bf16_max_rel_vs_fp32_oracle'))
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4t/gputests5.log 2>&1 || { tail -40 gpurun_out/r4t/gputests5.log; exit 1; }
tail -2 gpurun_out/r4t/gputests5.log
