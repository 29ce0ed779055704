#!/bin/bash
# Round-4 record collection on the GPU box (outputs under gpurun_out/$1; the summaries are folded into profiles/ afterwards, see
# profiles/README.md § r04).  Parts, selected by $2 (default "all"): suite | bench | prof | pmc | other | ab | stamps
set -o pipefail
tag=${1:-r4p}; what=${2:-all}
o=gpurun_out/$tag; mkdir -p $o
root=$(pwd)
export TMPDIR=/tmp
quick="--no-cpu-baseline --no-parity --no-other-workloads --no-haar"
has() { [[ "$what" == "all" || "$what" == *"$1"* ]]; }
if has suite; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee $o/pytest.rc; tail -3 $o/pytest.log
fi
if has bench; then
  timeout -k 10 600 python3 bench.py --breakdown $o/breakdown.json > $o/bench.json 2> $o/bench.err || exit 1
  echo "bench done"; tail -c 300 $o/bench.json; echo
fi
if has ab; then       # same session, same box: the producer-side GroupNorm on / off
  for v in X=1 DCAMD_NO_PN=1 X=2 DCAMD_NO_PN=2; do
    env $v timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 $quick > $o/ab_$v.json 2>/dev/null
    tail -1 $o/ab_$v.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" | tee -a $o/ab.log
  done
fi
if has prof; then
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/prof -- python3 $root/bench.py --steps 3 --warmup 1 $quick > $root/$o/prof.log 2>&1 || exit 2
  cd $root; echo "kernel-trace done"
fi
if has pmc; then
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/$o/pmc_fetch -- python3 $root/bench.py --steps 1 --warmup 1 $quick > $root/$o/pmc_fetch.log 2>&1 || exit 3
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/$o/pmc_write -- python3 $root/bench.py --steps 1 --warmup 1 $quick > $root/$o/pmc_write.log 2>&1 || exit 4
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $root/$o/pmc_sq -- python3 $root/bench.py --steps 1 --warmup 1 $quick > $root/$o/pmc_sq.log 2>&1 || exit 5
  cd $root; echo "pmc cfg2 done"
fi
if has other; then
  for wl in chexpert256-dwt-unet-2x100:bf16:cfg3 ipmsa5-unet-5x200:bf16:cfg4 chexpert256-dwt-dit-b4-2x250:f16:cfg5; do
    IFS=: read w d n <<< "$wl"
    python3 bench.py --workload $w --dtype $d --steps 3 --warmup 1 $quick --breakdown $o/${n}_breakdown.json > $o/${n}_bench.json 2> $o/${n}_bench.err || exit 6
    cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/${n}_prof -- python3 $root/bench.py --workload $w --dtype $d --steps 2 --warmup 1 $quick > $root/$o/${n}_prof.log 2>&1 || exit 7
    if [[ "$n" != "cfg5" ]]; then
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/$o/${n}_pmc_fetch -- python3 $root/bench.py --workload $w --dtype $d --steps 1 --warmup 1 $quick > $root/$o/${n}_pmc_fetch.log 2>&1 || exit 8
      rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/$o/${n}_pmc_write -- python3 $root/bench.py --workload $w --dtype $d --steps 1 --warmup 1 $quick > $root/$o/${n}_pmc_write.log 2>&1 || exit 9
    fi
    cd $root; echo "$n done"
  done
fi
if has stamps; then
  timeout -k 10 300 python3 tools/stamp_pn.py > $o/stamp_pn_32x32.log 2>&1
  RAW=0 RES=0 ABLS=0,16 timeout -k 10 300 python3 tools/stamp_pn.py > $o/stamp_pn_32x32_single.log 2>&1
  HW=16 N=8000 ABLS=0 timeout -k 10 300 python3 tools/stamp_pn.py > $o/stamp_pn_16x16.log 2>&1
  HW=64 N=500 ABLS=0,16 timeout -k 10 300 python3 tools/stamp_pn.py > $o/stamp_pn_64x64.log 2>&1
  grep -h "^---\|^kernel" $o/stamp_pn_*.log | head -40
fi
find $o -name "*kernel_trace.csv" -size +8M -delete
find $o -name "*.csv" -size +8M -delete
find $o -name "*.csv" | head -30
