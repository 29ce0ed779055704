#!/bin/bash
# round-3 final records: full GPU suite, default bench line + rocprofv3 stats + PMC passes (cfg2), stats / SQ counters of cfg5 and cfg3
set -o pipefail
o=gpurun_out/r3f; mkdir -p $o
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee $o/pytest.rc; tail -3 $o/pytest.log
bash tools/dev/collect_profiles.sh r3f > $o/collect.log 2>&1; echo "collect rc=$?"; tail -5 $o/collect.log
bash tools/dev/r3_profiles_other.sh r3fo > gpurun_out/r3fo_collect.log 2>&1; echo "other rc=$?"; tail -3 gpurun_out/r3fo_collect.log
find gpurun_out/r3f gpurun_out/r3fo -name "*.csv" -size +8M -delete
