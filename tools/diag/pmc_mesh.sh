#!/bin/bash
# SQ counters of the mesh pipeline's kernels (diagnostic)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/pmc_mesh; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/p1 -o run -- python3 bench.py --workload mesh --steps 10 --warmup 0 --no-graph --no-cpu-baseline > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o run -- python3 bench.py --workload mesh --steps 10 --warmup 0 --no-graph --no-cpu-baseline > $O/p2.log 2>&1
python3 - <<'PY'
import csv, collections
for sub in ('p1','p2'):
    acc=collections.defaultdict(lambda: [0.0,0])
    try:
        for r in csv.DictReader(open(f'gpurun_out/pmc_mesh/{sub}/run_counter_collection.csv')):
            k=(r['Kernel_Name'].split('(')[0][-28:], r['Counter_Name'])
            acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
    except Exception as e:
        print(sub, 'failed', e); continue
    for k in sorted(acc): 
        if 'icnn' in k[0] or 'loss' in k[0]: print(k[0].ljust(30), k[1].ljust(30), '%.4g' % (acc[k][0]/acc[k][1]), acc[k][1])
PY
