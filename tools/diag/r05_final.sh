# Round-5 closing evidence on the MI355X box (one gpurun call): the whole GPU suite, the poisoned-locals build against every
# general / learned-shape / welded / actuation test, the profile refresh, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -q -m gpu 2>&1 | tail -n 6 > gpurun_out/r05_final_gpu_suite.txt
r=$(DPLL_HIP_LIBRARY=$PWD/tools/diag/variants/libdpll_hip_poison.so timeout 1200 python3 -m pytest tests/test_general_models.py tests/test_hip_mesh.py tests/test_welded_links.py tests/test_actuation.py -m gpu -q --deselect tests/test_general_models.py::test_poisoned_locals_build 2>&1 | tail -n 1)
echo "final library of the round (ABI 25), general units with every local poisoned (make poison-check): general + learned-shape + welded-link + actuation GPU tests: $r" > gpurun_out/r05_final_poison.txt
bash tools/diag/refresh_profiles.sh r05 > gpurun_out/r05_refresh.log 2>&1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
cat gpurun_out/r05_final_gpu_suite.txt gpurun_out/r05_final_poison.txt; tail -c 300 gpurun_out/r05_bench_line.json
