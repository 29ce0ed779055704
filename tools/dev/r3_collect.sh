#!/bin/bash
# round-3 profile collection: full GPU suite, default bench line, rocprofv3 kernel stats + PMC passes (outputs under gpurun_out/$1)
set -o pipefail
tag=${1:-r3p}
o=gpurun_out/$tag; mkdir -p $o
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee $o/pytest.rc; tail -3 $o/pytest.log
bash tools/dev/collect_profiles.sh $tag > $o/collect.log 2>&1; echo "collect rc=$?"; tail -5 $o/collect.log
