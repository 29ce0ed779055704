set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
timeout -k 10 200 python tools/bench_tblock.py 8000 8 > gpurun_out/r4t/b1.log 2>&1; tail -1 gpurun_out/r4t/b1.log
timeout -k 10 200 python tools/bench_tblock.py 8000 4 >> gpurun_out/r4t/b1.log 2>&1; tail -1 gpurun_out/r4t/b1.log
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-other-workloads --no-cpu-baseline --no-haar > gpurun_out/r4t/bench_fused.log 2>&1 --breakdown gpurun_out/r4t/bd_fused.json; tail -c 600 gpurun_out/r4t/bench_fused.log
DCAMD_NO_TBLOCK=1 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-other-workloads --no-cpu-baseline --no-haar > gpurun_out/r4t/bench_nofuse.log 2>&1; tail -c 300 gpurun_out/r4t/bench_nofuse.log
