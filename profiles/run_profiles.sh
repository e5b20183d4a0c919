#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats for the task benches, then HBM byte counters for the dominant kernels.
# Usage: bash profiles/run_profiles.sh <tag>     outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
TASKS=${TASKS:-"Ant Humanoid ShadowHand AnymalTerrain Anymal AllegroHand Quadcopter BallBalance Ingenuity Crazyflie FrankaCabinet"}
PMC=${PMC:-1}
OUT=/tmp/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for T in $TASKS; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$T -o $T -- python3 bench.py --task $T --steps 50 --warmup 5 --no-cpu-baseline > $OUT/$T.bench.log 2>&1 || { echo "stats $T failed"; exit 1; }
  echo "stats $T done"
done
[ "$PMC" = "1" ] && for T in $TASKS; do
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${T}_fetch -o $T -- python3 bench.py --task $T --steps 10 --warmup 2 --no-cpu-baseline > $OUT/$T.fetch.log 2>&1 || { echo "fetch $T failed"; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${T}_write -o $T -- python3 bench.py --task $T --steps 10 --warmup 2 --no-cpu-baseline > $OUT/$T.write.log 2>&1 || { echo "write $T failed"; exit 1; }
  echo "pmc $T done"
done
python3 profiles/summarize.py $OUT $TAG && ls -la gpurun_out/summary_$TAG
