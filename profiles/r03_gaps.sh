#!/bin/bash
# idle time between consecutive kernels of the Ant step, with the bench's event brackets on every step / every 5th / none
set -e
mkdir -p gpurun_out
out=gpurun_out/step_gaps_Ant.txt
: > $out
for ev in 1 5 1000; do
  echo "=== --event-every $ev ===" >> $out
  bash profiles/step_gaps.sh Ant 100 --event-every $ev >> $out
done
