#!/bin/bash
# pre-kernel durations, this tree against omniisaacgymenvs_amd/variants/liboige_prev.so: bash profiles/pre_ab.sh Task1 Task2 ...
for T in "$@"; do
  echo "#### $T"
  TASK=$T bash profiles/kernel_variants.sh prev | grep -E "^==|^kernel"
done
