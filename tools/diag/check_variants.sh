#!/bin/bash
# Diagnostic (GPU box): tools/diag/check_general.py over every library under tools/diag/variants, twice each
out=${1:-gpurun_out/variants.txt}
: > $out
for lib in tools/diag/variants/*.so; do
  for round in 1 2; do
    echo "== $lib (round $round)" >> $out
    DPLL_BISECT_ABI=1 DPLL_LIB=$lib DPLL_MODELS="${DPLL_MODELS:-gripper grasp}" DPLL_F64_ONLY=${DPLL_F64_ONLY-1} python tools/diag/check_general.py 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
