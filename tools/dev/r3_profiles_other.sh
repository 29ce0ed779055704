#!/bin/bash
# developer tool: rocprofv3 kernel stats + SQ counters of the DiT-B/4 (cfg5) and CheXpert-DWT UNet (cfg3) workloads (outputs under gpurun_out/$1)
set -o pipefail
out=gpurun_out/${1:-r3o}
mkdir -p $out
root=$(pwd)
export TMPDIR=/tmp
common="--no-cpu-baseline --no-parity --no-other-workloads --no-haar"
python3 bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 --steps 3 --warmup 1 $common --breakdown $out/cfg5_breakdown.json > $out/cfg5_bench.json 2> $out/cfg5_bench.err || exit 1
python3 bench.py --workload chexpert256-dwt-unet-2x100 --steps 3 --warmup 1 $common --breakdown $out/cfg3_breakdown.json > $out/cfg3_bench.json 2> $out/cfg3_bench.err || exit 1
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/cfg5_prof -- python3 $root/bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 --steps 2 --warmup 1 $common > $root/$out/cfg5_prof.log 2>&1 || exit 2
echo "cfg5 kernel-trace done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $root/$out/cfg5_pmc_sq -- python3 $root/bench.py --workload chexpert256-dwt-dit-b4-2x250 --dtype f16 --steps 1 --warmup 1 $common > $root/$out/cfg5_pmc_sq.log 2>&1 || exit 3
echo "cfg5 sq done"
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/cfg3_prof -- python3 $root/bench.py --workload chexpert256-dwt-unet-2x100 --steps 2 --warmup 1 $common > $root/$out/cfg3_prof.log 2>&1 || exit 4
echo "cfg3 kernel-trace done"
cd $root
find $out -name "*kernel_trace.csv" -size +20M -delete
find $out -name "*.csv" | head
