#!/bin/bash
# kernel-trace stats of one bench task: bash profiles/kernel_times.sh <Task> [extra bench args]
T=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/kt_$T
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$T -o k -- python3 bench.py --task $T --steps 50 --warmup 5 --no-cpu-baseline --no-graph-leg "$@" > /tmp/kt_$T.log 2>&1 || { echo "failed"; tail -5 /tmp/kt_$T.log; exit 1; }
f=$(find /tmp/kt_$T -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    if "oige::" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
grep '^{"metric' /tmp/kt_$T.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step_ms', round(d['ms_per_step'],4), 'post_ms', round(d['roofline']['avg_kernel_ms'],4), 'pre_ms', round(d['roofline']['pre_physics_ms'],4), 'frac', round(d['roofline']['frac'],4))"
