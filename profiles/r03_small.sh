#!/bin/bash
# eager and graph-replay step at the reference's own size (4096 envs), per task: the host-bound regime
for T in ${TASKS:-Cartpole Ant Humanoid Anymal Quadcopter BallBalance Ingenuity Crazyflie FrankaCabinet TomoDJnewhandCabinet AllegroHand AnymalTerrain}; do
  timeout -k 10 200 python bench.py --task $T --envs 4096 --steps 300 --warmup 30 --no-cpu-baseline > /tmp/s_$T.json 2>/tmp/s_$T.err || { echo "$T failed"; tail -3 /tmp/s_$T.err; continue; }
  python3 - $T <<'PY'
import json, sys
d = json.loads([l for l in open(f"/tmp/s_{sys.argv[1]}.json") if l.startswith("{")][-1])
print(f"{sys.argv[1]:28s} eager {d['ms_per_step']*1e3:7.1f} us/step  graph {(d.get('graph_ms_per_step') or 0)*1e3:7.1f} us/step  ({d['value']/1e6:7.1f} M env-steps/s eager)")
PY
done
