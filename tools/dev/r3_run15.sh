#!/bin/bash
set -o pipefail
timeout -k 10 700 bash tools/dev/project_scaling.sh gpurun_out/r3s > gpurun_out/r3s.log 2>&1; echo "scaling rc=$?"; tail -5 gpurun_out/r3s.log
timeout -k 10 450 bash tools/dev/r3_profiles_other.sh r3o > gpurun_out/r3o.log 2>&1; echo "other rc=$?"; tail -8 gpurun_out/r3o.log
