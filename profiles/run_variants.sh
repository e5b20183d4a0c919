#!/bin/bash
# bench each kernel variant (omniisaacgymenvs_amd/variants/liboige_<name>.so) on the GPU box: TASK=<task> bash profiles/run_variants.sh name1 name2 ...
TASK=${TASK:-Ant}
for V in base "$@"; do
  if [ "$V" = "base" ]; then unset OIGE_LIB; else export OIGE_LIB=$PWD/omniisaacgymenvs_amd/variants/liboige_$V.so; fi
  timeout -k 10 120 python bench.py --task $TASK --no-cpu-baseline --steps 300 > /tmp/v.json 2>/tmp/v.err || { echo "$V failed"; tail -5 /tmp/v.err; exit 1; }
  python3 -c "import json; d=json.loads(open('/tmp/v.json').read().strip().splitlines()[-1]); print('$V', 'step_ms', round(d['ms_per_step'],4), 'post_ms', round(d['roofline']['avg_kernel_ms'],4), 'pre_ms', round(d['roofline']['pre_physics_ms'],4), 'frac', round(d['roofline']['frac'],4))"
done
