#!/bin/bash
# Diagnostic (GPU box): the step-backward tests of the 3-joint models against every tools/diag/variants/poison_*.so
for lib in $(ls tools/diag/variants/poison_*.so | sort -t_ -k2 -n); do
  r=$(DPLL_HIP_LIBRARY=$lib python -m pytest tests/test_general_models.py -m gpu -q -k "step_backward and (gripper or grasp)" 2>&1 | tail -n 1)
  echo "$lib: $r"
done
