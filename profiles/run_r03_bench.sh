#!/bin/bash
# r03 bench lines into gpurun_out/r03/: per task the default workload, --reset-frac 0.001 (the reference's steady regime) and the
# default workload with the r01 / r02 harness (--event-every 1: three event records per step), so that the harness-side share of
# the r02 -> r03 change can be read off next to the kernel-side share.
mkdir -p gpurun_out/r03
for T in ${TASKS:-Ant Humanoid ShadowHand AnymalTerrain}; do
  timeout -k 10 300 python bench.py --task $T --steps 200 --warmup 20 ${EXTRA:-} > gpurun_out/r03/bench_$T.json 2> gpurun_out/r03/bench_$T.err || { echo "$T failed"; tail -5 gpurun_out/r03/bench_$T.err; exit 1; }
  timeout -k 10 300 python bench.py --task $T --steps 200 --warmup 20 --reset-frac 0.001 --no-cpu-baseline ${EXTRA:-} > gpurun_out/r03/bench_${T}_steady.json 2> gpurun_out/r03/bench_${T}_steady.err || { echo "$T steady failed"; tail -5 gpurun_out/r03/bench_${T}_steady.err; exit 1; }
  timeout -k 10 300 python bench.py --task $T --steps 200 --warmup 20 --event-every 1 --no-cpu-baseline --no-graph-leg ${EXTRA:-} > gpurun_out/r03/bench_${T}_events_every_step.json 2> gpurun_out/r03/bench_${T}_events_every_step.err || { echo "$T every-step failed"; exit 1; }
  rm -f gpurun_out/r03/*.err
  python3 - $T <<'PY'
import json, sys
t = sys.argv[1]
for tag in ("", "_steady", "_events_every_step"):
    d = json.loads(open(f"gpurun_out/r03/bench_{t}{tag}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    net = r.get("net_of_event_overhead", {})
    print(f"{t+tag:44s} G/s {d['value']/1e9:6.3f} step {d['ms_per_step']*1e3:7.1f} us graph {(d.get('graph_ms_per_step') or 0)*1e3:7.1f} gpu {r['gpu_ms_per_step']*1e3:7.1f} post {r['avg_kernel_ms']*1e3:6.1f} ({r['frac']:.3f}; net {net.get('frac', 0):.3f}) pre {r['pre_physics_ms']*1e3:6.1f} ({r.get('pre',{}).get('frac',0):.3f}; net {net.get('pre_frac', 0):.3f}) stepfrac {r.get('step',{}).get('frac',0):.3f} resets {d['config']['resets_per_step']}")
PY
done
