#!/bin/bash
# SQ counters of one kernel: bash profiles/pmc_kernel.sh <Task> <kernel substring> <counter> [<counter> ...]   (<= 8 SQ_ per pass)
T=$1; K=$2; shift 2
export TMPDIR=/tmp
rm -rf /tmp/pmc_$T
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$T -o p -- python3 bench.py --task $T --steps 4 --warmup 2 --no-cpu-baseline --no-graph-leg > /tmp/pmc_$T.log 2>&1 || { echo failed; tail -5 /tmp/pmc_$T.log; exit 1; }
f=$(find /tmp/pmc_$T -name "*counter_collection.csv" | head -1)
python3 - "$f" "$K" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} launches {len(v):4d} mean {sum(v) / len(v):16.1f}")
PY
