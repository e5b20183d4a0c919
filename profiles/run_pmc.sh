#!/bin/bash
# one rocprofv3 --pmc pass over a bench run: TASK=Ant bash profiles/run_pmc.sh <tag> COUNTER [COUNTER...]
TASK=${TASK:-Ant}
TAG=$1; shift
export TMPDIR=/tmp
OUT=/tmp/pmc_$TAG
rm -rf $OUT
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT -o p -- python3 bench.py --task $TASK --steps 10 --warmup 2 --no-cpu-baseline > /tmp/pmc_$TAG.log 2>&1 || { echo "pmc pass failed"; tail -5 /tmp/pmc_$TAG.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "oige::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in cs.items():
        print(f"    {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
