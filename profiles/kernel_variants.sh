#!/bin/bash
# kernel durations and gaps per variant: TASK=<task> bash profiles/kernel_variants.sh name1 name2 ... (base first)
TASK=${TASK:-Ant}
for V in base "$@"; do
  if [ "$V" = "base" ]; then unset OIGE_LIB; else export OIGE_LIB=$PWD/omniisaacgymenvs_amd/variants/liboige_$V.so; fi
  echo "== $V"
  bash profiles/step_gaps.sh $TASK 150 $EXTRA | grep -v "at::\|rocclr\|Cijk\|vectorized\|elementwise" 
done
