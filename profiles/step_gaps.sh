#!/bin/bash
# kernel durations and inter-kernel gaps of one bench task: bash profiles/step_gaps.sh <Task> <last-N-launches> [extra bench args]
T=$1; N=$2; shift 2
export TMPDIR=/tmp
rm -rf /tmp/sg_$T
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/sg_$T -o k -- python3 bench.py --task $T --steps 100 --warmup 5 --no-cpu-baseline --no-graph-leg "$@" > /tmp/sg_$T.log 2>&1 || { echo "failed"; tail -5 /tmp/sg_$T.log; exit 1; }
f=$(find /tmp/sg_$T -name "*kernel_trace.csv" | head -1)
python3 profiles/tools/step_gaps.py "$f" $N
