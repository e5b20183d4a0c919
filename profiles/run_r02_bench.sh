#!/bin/bash
# r02 bench lines: the four north-star tasks, default workload and --reset-frac 0.001, into gpurun_out/r02/
mkdir -p gpurun_out/r02
for T in ${TASKS:-Ant Humanoid ShadowHand AnymalTerrain}; do
  timeout -k 10 300 python bench.py --task $T --steps 200 --warmup 20 ${EXTRA:-} > gpurun_out/r02/bench_$T.json 2> gpurun_out/r02/bench_$T.err || { echo "$T failed"; tail -5 gpurun_out/r02/bench_$T.err; exit 1; }
  timeout -k 10 300 python bench.py --task $T --steps 200 --warmup 20 --reset-frac 0.001 --no-cpu-baseline ${EXTRA:-} > gpurun_out/r02/bench_${T}_steady.json 2> gpurun_out/r02/bench_${T}_steady.err || { echo "$T steady failed"; tail -5 gpurun_out/r02/bench_${T}_steady.err; exit 1; }
  python3 - $T <<'PY'
import json, sys
t = sys.argv[1]
for tag in ("", "_steady"):
    d = json.loads(open(f"gpurun_out/r02/bench_{t}{tag}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{t+tag:22s} G/s {d['value']/1e9:6.3f} step {d['ms_per_step']*1e3:7.1f} us graph {(d.get('graph_ms_per_step') or 0)*1e3:7.1f} gpu {r['gpu_ms_per_step']*1e3:7.1f} post {r['avg_kernel_ms']*1e3:6.1f} ({r['frac']:.3f}) pre {r['pre_physics_ms']*1e3:6.1f} ({r.get('pre',{}).get('frac',0):.3f}) stepfrac {r.get('step',{}).get('frac',0):.3f} resets {d['config']['resets_per_step']}")
PY
done
