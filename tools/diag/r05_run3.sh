cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r05c/gpu_suite.txt; cat gpurun_out/r05c/gpu_suite.txt
echo "every automatic variable of all five translation units pre-filled with clang's poison pattern (-ftrivial-auto-var-init=pattern), general units with -amdgpu-promote-alloca-to-vector-limit=190: the whole GPU suite" > gpurun_out/r05c/poison_everything.txt
DPLL_HIP_LIBRARY=tools/diag/variants/poison_everything.so timeout 2400 python3 -m pytest tests -q -m gpu 2>&1 | tail -6 >> gpurun_out/r05c/poison_everything.txt; cat gpurun_out/r05c/poison_everything.txt
timeout 900 python3 tools/diag/time_forest.py > gpurun_out/r05c/forest_times.txt 2>&1; cat gpurun_out/r05c/forest_times.txt
