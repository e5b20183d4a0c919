#!/bin/bash
# r02 bench lines of the section-8f tasks (default workload, CPU baseline skipped), into gpurun_out/r02/
mkdir -p gpurun_out/r02
for T in ${TASKS:-Anymal AllegroHand Quadcopter BallBalance Ingenuity Crazyflie FrankaCabinet TomoDJCabinet TomoSPCCabinet TomoDJnewhandCabinet TomoDJnewhandTendonCabinet TomoSPCnewhandCabinet}; do
  timeout -k 10 200 python bench.py --task $T --steps 200 --warmup 20 --no-cpu-baseline ${EXTRA:-} > gpurun_out/r02/bench_$T.json 2> gpurun_out/r02/bench_$T.err || { echo "$T failed"; tail -5 gpurun_out/r02/bench_$T.err; exit 1; }
  python3 - $T <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open(f"gpurun_out/r02/bench_{t}.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{t:28s} G/s {d['value']/1e9:6.3f} step {d['ms_per_step']*1e3:7.1f} us graph {(d.get('graph_ms_per_step') or 0)*1e3:7.1f} gpu {r['gpu_ms_per_step']*1e3:7.1f} post {r['avg_kernel_ms']*1e3:6.1f} ({r['frac']:.3f}) pre {r['pre_physics_ms']*1e3:6.1f}")
PY
done
