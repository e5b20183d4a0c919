#!/usr/bin/env python3
"""Per-step kernel durations and the idle gaps between consecutive kernels, from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python3 bench.py --task Ant --steps 50 --no-cpu-baseline --no-graph-leg
    python profiles/tools/step_gaps.py /tmp/kt/**/k_kernel_trace.csv [last-N-launches] [name-filter]

Prints, over the window that ends at the LAST launch whose name contains the filter (default "oige::": the library's own
kernels, so the bench's bookkeeping behind the timed loop stays out) and holds N launches (default 200: timed steps, past
warm-up and calibration), each kernel's mean duration and the mean gap between a kernel's end and the next kernel's start,
keyed by the pair of kernels."""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
flt = sys.argv[3] if len(sys.argv) > 3 else "oige::"
last = max((i for i, r in enumerate(rows) if flt in r["Kernel_Name"]), default=len(rows) - 1)
rows = rows[max(0, last + 1 - n):last + 1]


def short(name):
    name = name.replace("oige::", "").replace("void ", "")
    return name.split("(")[0][:48]


dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gap[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows:
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"kernel {k:50s} n {len(v):4d} mean {sum(v) / len(v) / 1e3:8.2f} us  min {min(v) / 1e3:8.2f}")
for (a, b), v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    print(f"gap    {a:34s} -> {b:34s} n {len(v):4d} mean {sum(v) / len(v) / 1e3:7.2f} us  min {min(v) / 1e3:7.2f}")
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"span of the {len(rows)} launches: {span / 1e3:.1f} us")
