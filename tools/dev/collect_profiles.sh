#!/bin/bash
# developer tool: collect the records profiles/ keeps for one build (run on the GPU box; outputs under gpurun_out/$1)
# usage: bash tools/dev/collect_profiles.sh r2x
set -o pipefail
out=gpurun_out/${1:-prof}
mkdir -p $out
root=$(pwd)
export TMPDIR=/tmp
python3 bench.py --breakdown $out/breakdown.json > $out/bench.json 2> $out/bench.err || exit 1
echo "bench done"; tail -c 400 $out/bench.json | head -c 200; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-other-workloads --no-haar > $root/$out/prof.log 2>&1 || exit 2
echo "kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/$out/pmc_fetch -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-other-workloads --no-haar > $root/$out/pmc_fetch.log 2>&1 || exit 3
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/$out/pmc_write -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-other-workloads --no-haar > $root/$out/pmc_write.log 2>&1 || exit 4
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $root/$out/pmc_sq -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-other-workloads --no-haar > $root/$out/pmc_sq.log 2>&1 || exit 5
echo "sq done"
cd $root
find $out -name "*.csv" | head -20
# keep only the small summaries (the traces are large)
find $out -name "*kernel_trace.csv" -size +20M -delete
