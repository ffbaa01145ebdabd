cd $GRAFT_REPO_ROOT
K='step_backward and (gripper or grasp)'
for lib in tools/diag/variants/poison_all.so tools/diag/variants/poison190.so tools/diag/variants/lim190.so; do
  r=$(DPLL_HIP_LIBRARY=$lib timeout 600 python3 -m pytest tests/test_general_models.py -m gpu -q -k "$K" 2>&1 | tail -n 1); echo "$lib: $r"
done
r=$(DPLL_HIP_LIBRARY=tools/diag/variants/poison190.so timeout 1200 python3 -m pytest tests/test_general_models.py tests/test_hip_mesh.py -m gpu -q 2>&1 | tail -n 1); echo "poison190, all general + mesh tests: $r"
echo "== shipped"; timeout 600 python3 tools/diag/time_general.py 2>&1 | tail -14
echo "== limit 190"; DPLL_HIP_LIBRARY=tools/diag/variants/lim190.so timeout 600 python3 tools/diag/time_general.py 2>&1 | tail -14
echo "== shipped"; timeout 600 python3 tools/diag/time_general.py 2>&1 | tail -14
echo "== limit 190"; DPLL_HIP_LIBRARY=tools/diag/variants/lim190.so timeout 600 python3 tools/diag/time_general.py 2>&1 | tail -14
