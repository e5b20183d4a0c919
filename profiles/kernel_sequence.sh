#!/bin/bash
# the kernels of the last few steps of a bench run, in launch order: bash profiles/kernel_sequence.sh <Task> [extra bench args]
T=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/ks_$T
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_$T -o k -- python3 bench.py --task $T --steps 6 --warmup 2 --no-cpu-baseline --no-graph-leg "$@" > /tmp/ks_$T.log 2>&1 || { echo "failed"; tail -5 /tmp/ks_$T.log; exit 1; }
f=$(find /tmp/ks_$T -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"][:90], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
# print the tail: the last 40 launches before the calibration copies / statistics at the end
idx = [i for i, (n, _) in enumerate(names) if "post_kernel" in n or "anymal_scan_kernel" in n]
lo = idx[-3] if len(idx) >= 3 else 0
for n, us in names[lo: idx[-1] + 4]:
    print(f"{us:9.1f} us  {n}")
PY
