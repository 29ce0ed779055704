#!/bin/bash
# round-3 GPU session 1: full GPU suite, the default bench line (with other_workloads), the scaling projection
set -o pipefail
o=gpurun_out/r3a; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee $o/pytest.rc
tail -5 $o/pytest.log
timeout -k 10 600 python3 bench.py --breakdown $o/breakdown.json > $o/bench.json 2> $o/bench.log; echo "bench rc=$?" | tee $o/bench.rc
tail -c 1500 $o/bench.json
timeout -k 10 600 tools/dev/project_scaling.sh $o/scaling > $o/scaling.log 2>&1; echo "scaling rc=$?"
tail -5 $o/scaling.log
