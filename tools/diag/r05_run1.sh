cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05a
timeout 900 python3 -m pytest tests/test_hip_mesh.py -x -q -m gpu > gpurun_out/r05a/test_mesh.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a/test_mesh.log
tail -5 gpurun_out/r05a/test_mesh.log
timeout 300 python3 tools/diag/time_mesh_ab.py 0 1 2 > gpurun_out/r05a/ab.log 2>&1; tail -20 gpurun_out/r05a/ab.log
BATCH=65536 timeout 300 python3 tools/diag/time_mesh_ab.py 0 1 > gpurun_out/r05a/ab64k.log 2>&1; tail -8 gpurun_out/r05a/ab64k.log
cd /tmp && timeout 200 rocprofv3 --att --kernel-include-regex icnn_bwd1 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05a/att -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload mesh --steps 2 --warmup 0 --no-graph --no-cpu-baseline --no-configs > $GRAFT_REPO_ROOT/gpurun_out/r05a/att.log 2>&1; echo "att rc $?"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/r05a/att.log
