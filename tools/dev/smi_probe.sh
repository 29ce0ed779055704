#!/bin/bash
# developer diagnostic: sample socket power / clocks while the bench runs (every 0.5 s, from start to exit)
mkdir -p gpurun_out/r2u
python bench.py --no-parity --no-cpu-baseline --steps 150 --warmup 2 > gpurun_out/r2u/b.json 2> gpurun_out/r2u/b.err &
BP=$!
: > gpurun_out/r2u/smi.log
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed 's/[[:space:]]\+/ /g;s/GPU\[0\] : //' | tr '\n' ';' >> gpurun_out/r2u/smi.log
  echo >> gpurun_out/r2u/smi.log
  sleep 0.4
done
wait $BP
tail -c 200 gpurun_out/r2u/b.json
