set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
for spw in 1 2 1; do
DCAMD_TB_SPW=$spw timeout -k 10 200 python tools/bench_tblock.py 8000 8 > gpurun_out/r4t/b4_$spw.log 2>&1; echo spw $spw; tail -1 gpurun_out/r4t/b4_$spw.log
done
DCAMD_TB_SPW=1 timeout -k 10 120 python tools/stamp_tblock.py 8000 8 > gpurun_out/r4t/stamp4.log 2>&1; cat gpurun_out/r4t/stamp4.log
DCAMD_TB_SPW=1 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k tblock > gpurun_out/r4t/t4.log 2>&1 || { tail -30 gpurun_out/r4t/t4.log; exit 1; }
tail -1 gpurun_out/r4t/t4.log
