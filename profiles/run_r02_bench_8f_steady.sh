#!/bin/bash
# r02 steady-state bench lines (--reset-frac 0.001) of the section-8f tasks, into gpurun_out/r02/
mkdir -p gpurun_out/r02
for T in ${TASKS:-Anymal AllegroHand Quadcopter BallBalance Ingenuity Crazyflie FrankaCabinet TomoDJCabinet TomoSPCCabinet TomoDJnewhandCabinet TomoDJnewhandTendonCabinet TomoSPCnewhandCabinet}; do
  timeout -k 10 200 python bench.py --task $T --steps 200 --warmup 20 --no-cpu-baseline --reset-frac 0.001 > gpurun_out/r02/bench_${T}_steady.json 2> gpurun_out/r02/bench_${T}_steady.err || { echo "$T failed"; tail -5 gpurun_out/r02/bench_${T}_steady.err; exit 1; }
  python3 - $T <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open(f"gpurun_out/r02/bench_{t}_steady.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{t+'_steady':28s} G/s {d['value']/1e9:6.3f} step {d['ms_per_step']*1e3:7.1f} us gpu {r['gpu_ms_per_step']*1e3:7.1f} post {r['avg_kernel_ms']*1e3:6.1f} ({r['frac']:.3f}) pre {r['pre_physics_ms']*1e3:6.1f} resets {d['config']['resets_per_step']}")
PY
done
