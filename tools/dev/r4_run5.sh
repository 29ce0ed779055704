set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
B="--steps 4 --warmup 1 --no-other-workloads --no-cpu-baseline --no-haar"
timeout -k 10 300 python bench.py $B --breakdown gpurun_out/r4t/bd6.json > gpurun_out/r4t/bench6.log 2>&1
timeout -k 10 300 python bench.py $B > gpurun_out/r4t/bench6b.log 2>&1
python - <<'PY'
import json
for f in ('bench6','bench6b'):
    l=[x for x in open(f'gpurun_out/r4t/{f}.log') if x.startswith('{')][-1]
    d=json.loads(l); print(f, d['value'], d['ms_per_step'], d['parity'].get('bf16_pred_rel_l2_vs_fp32_oracle'), d['parity'].get('f32_max_rel'), d['parity'].get('f32_labels_equal'))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -x -q > gpurun_out/r4t/gputests6.log 2>&1 || { tail -40 gpurun_out/r4t/gputests6.log; exit 1; }
tail -2 gpurun_out/r4t/gputests6.log
