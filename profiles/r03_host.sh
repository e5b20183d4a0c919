#!/bin/bash
# host cost per step after the argument-block cache: GPU tests of the touched mirrors, cProfile at 4096 envs, the first timed steps
set -e
timeout -k 10 600 python -u -m pytest tests/test_locomotion_gpu.py tests/test_graph_step_gpu.py tests/test_wrapper_gpu.py tests/test_wrapper_all_tasks_gpu.py tests/test_counts_valid_gpu.py tests/test_hand_gpu.py tests/test_allegro_hand_gpu.py tests/test_anymal_gpu.py -x -q -m gpu --timeout 120 --timeout-method thread 2>&1 | tail -4
for T in "$@"; do
  timeout -k 10 120 python profiles/tools/host_overhead.py $T 2000 2>&1 | tail -18
  timeout -k 10 100 python bench.py --task $T --steps 200 --warmup 20 --no-cpu-baseline --envs 4096 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T 4096 envs: ms_per_step', round(d['ms_per_step'],4), 'graph', d['graph_ms_per_step'])"
  OIGE_BENCH_DUMP_STEPS=1 timeout -k 10 100 python bench.py --task $T --steps 20 --warmup 5 --no-cpu-baseline --no-graph-leg 2>&1 | grep -E "^host us|^steps" | cut -c1-330
done
