#!/bin/bash
# Regenerates the raw material of profiles/ on the MI355X box (run through gpurun from the repo root):
#   tools/diag/refresh_profiles.sh [tag, default r02]
# Every rocprofv3 pass is its own process; counter passes carry no trace domain.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-r05}
O=gpurun_out/${TAG}p
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-configs"
python3 bench.py --steps 2000 --warmup 100 > $O/bench_f32.json 2> $O/bench_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 300 --warmup 20 $B > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f64 -o run -- python3 bench.py --dtype f64 --steps 300 --warmup 20 $B > $O/stats_f64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_elbow -o run -- python3 bench.py --workload elbow --steps 300 --warmup 20 $B > $O/stats_elbow.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_elbow_f64 -o run -- python3 bench.py --workload elbow --dtype f64 --steps 300 --warmup 20 $B > $O/stats_elbow_f64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b65536 -o run -- python3 bench.py --batch 65536 --steps 100 --warmup 20 $B > $O/stats_b65536.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b65536_f64 -o run -- python3 bench.py --batch 65536 --dtype f64 --steps 100 --warmup 20 $B > $O/stats_b65536_f64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mesh -o run -- python3 bench.py --workload mesh --mesh-gemm 0 --steps 50 --warmup 50 $B > $O/stats_mesh.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mesh_bf16 -o run -- python3 bench.py --workload mesh --mesh-gemm 2 --steps 50 --warmup 50 $B > $O/stats_mesh_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mesh_f16 -o run -- python3 bench.py --workload mesh --mesh-gemm 4 --steps 50 --warmup 50 $B > $O/stats_mesh_f16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sim -o run -- python3 tools/diag/sim_bench.py > $O/stats_sim.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_general -o run -- python3 tools/diag/time_general.py > $O/stats_general.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_clasp_mesh -o run -- python3 tools/diag/time_clasp_mesh.py > $O/stats_clasp_mesh.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 bench.py --steps 20 --warmup 0 --no-graph $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 bench.py --steps 20 --warmup 0 --no-graph $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq1 -o run -- python3 bench.py --steps 20 --warmup 0 --no-graph $B > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_sq1_p1 -o run -- python3 bench.py --steps 20 --warmup 0 --no-graph --portfolio 1 $B > $O/pmc_sq1_p1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $O/pmc_sq2 -o run -- python3 bench.py --steps 20 --warmup 0 --no-graph $B > $O/pmc_sq2.log 2>&1
[ -f tools/diag/libdpll_hip_stamps.so ] && python3 tools/diag/stamps.py > $O/stamps.txt 2>&1  # (the -DDPLL_STAMPS build of the CURRENT ABI: see tools/diag/stamps.py)
bash tools/diag/pmc_mesh.sh > $O/pmc_mesh.txt 2>&1
cp -r gpurun_out/pmc_mesh $O/ 2>/dev/null
python3 tools/diag/time_general.py > $O/general_times.txt 2>&1
python3 tools/diag/time_forest.py > $O/forest_times.txt 2>&1
[ -f tools/diag/libdpll_hip_fstamps.so ] && DPLL_HIP_LIBRARY=tools/diag/libdpll_hip_fstamps.so python3 tools/diag/forest_stamps.py cube two_cubes gripper chain6 > $O/forest_stamps.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_forest -o run -- python3 tools/diag/time_forest.py two_cubes chain6 > $O/stats_forest.log 2>&1
# keep what profiles/ is made of: the stats summaries, the counter tables, the tail of the headline trace
head -1 $O/stats/run_kernel_trace.csv > $O/stats/trace_tail.csv; tail -40 $O/stats/run_kernel_trace.csv >> $O/stats/trace_tail.csv
find $O -name 'run_kernel_trace.csv' -delete
find $O -name '*_agent_info.csv' -delete
# the summaries profiles/ keeps are made here (the raw counter tables are tens of MB: over gpurun's 64 MiB return limit)
for f in general_times.txt pmc_mesh.txt forest_times.txt forest_stamps.txt; do cp $O/$f gpurun_out/r03_profiles_$f 2>/dev/null; done
cp gpurun_out/mesh_pmc.csv gpurun_out/r03_profiles_mesh_pmc.csv 2>/dev/null
DPLL_PROFILE_DST=gpurun_out/${TAG}_profiles python3 tools/diag/summarize_profiles.py ${TAG}p ${TAG} > gpurun_out/${TAG}_summarize.log 2>&1
mv gpurun_out/r03_profiles_*.txt gpurun_out/r03_profiles_*.csv gpurun_out/${TAG}_profiles/ 2>/dev/null
rm -rf $O
ls gpurun_out/${TAG}_profiles
