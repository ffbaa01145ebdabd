#!/bin/bash
# SQ counters of the mesh pipeline's kernels (diagnostic)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/pmc_mesh; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/p1 -o run -- python3 bench.py --workload mesh --mesh-gemm 0 --steps 10 --warmup 0 --no-graph --no-cpu-baseline > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o run -- python3 bench.py --workload mesh --mesh-gemm 0 --steps 10 --warmup 0 --no-graph --no-cpu-baseline > $O/p2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/p3 -o run -- python3 bench.py --workload mesh --mesh-gemm 2 --steps 10 --warmup 0 --no-graph --no-cpu-baseline > $O/p3.log 2>&1
python3 - <<'PY'
import csv, collections, re
rows = []
for sub, tag in (('p1', ''), ('p2', ''), ('p3', ' [mesh_gemm=2]')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    try:
        for r in csv.DictReader(open(f'gpurun_out/pmc_mesh/{sub}/run_counter_collection.csv')):
            m = re.search(r'(icnn_pipe_kernel<[^>]*>|icnn_\w+(?:<\d>)?|loss_kernel)', r['Kernel_Name'])
            if not m:
                continue
            k = (m.group(1) + tag, r['Counter_Name'])
            acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    except Exception as e:
        print(sub, 'failed', e); continue
    rows += [(k[0], k[1], acc[k][0] / acc[k][1], acc[k][1]) for k in sorted(acc)]
with open('gpurun_out/mesh_pmc.csv', 'w') as f:
    f.write('kernel,counter,mean_per_dispatch,dispatches\n')
    for r in rows:
        f.write(f'{r[0]},{r[1]},{r[2]:g},{r[3]}\n')
        print(r[0].ljust(34), r[1].ljust(30), '%.4g' % r[2], r[3])
PY
