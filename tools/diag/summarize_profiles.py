"""Turns the raw rocprofv3 output of tools/diag/refresh_profiles.sh (gpurun_out/r01p) into the files kept under profiles/."""
import csv, json, os, re, shutil, sys
from collections import defaultdict
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(REPO, 'gpurun_out', sys.argv[1] if len(sys.argv) > 1 else 'r05p')
DST = os.environ.get('DPLL_PROFILE_DST', os.path.join(REPO, 'profiles'))  # (on the GPU box: a directory under gpurun_out/)
os.makedirs(DST, exist_ok=True)
TAG = sys.argv[2] if len(sys.argv) > 2 else 'r05'
KERNEL = 'loss_kernel<float, 0, false'  # <T, NJ, MESH, DENSE>: the box builds


def counter_means(path):
    sums, counts = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(path)):
        if KERNEL in row['Kernel_Name']:
            sums[row['Counter_Name']] += float(row['Counter_Value'])
            counts[row['Counter_Name']] += 1
    return {k: (sums[k] / counts[k], counts[k]) for k in sums}


shutil.copy(os.path.join(SRC, 'stats', 'run_kernel_stats.csv'), os.path.join(DST, f'{TAG}_bench_f32_kernel_stats.csv'))
for sub, name in (('stats_mesh', 'mesh_f32'), ('stats_f64', 'f64'), ('stats_elbow', 'elbow_f32'), ('stats_elbow_f64', 'elbow_f64'),
                  ('stats_b65536', 'f32_b65536'), ('stats_b65536_f64', 'f64_b65536'), ('stats_sim', 'simulate'),
                  ('stats_mesh_bf16', 'mesh_bf16'), ('stats_mesh_f16', 'mesh_f16'), ('stats_general', 'general_build'), ('stats_clasp_mesh', 'clasp_mesh'),
                  ('stats_forest', 'forest_build')):
    src = os.path.join(SRC, sub, 'run_kernel_stats.csv')
    if os.path.exists(src):
        shutil.copy(src, os.path.join(DST, f'{TAG}_bench_{name}_kernel_stats.csv'))
shutil.copy(os.path.join(SRC, 'stats', 'trace_tail.csv'), os.path.join(DST, f'{TAG}_bench_f32_kernel_trace_tail.csv'))
means = {}
for sub in ('pmc_sq1', 'pmc_sq2'):
    means.update(counter_means(os.path.join(SRC, sub, 'run_counter_collection.csv')))
p1 = os.path.join(SRC, 'pmc_sq1_p1', 'run_counter_collection.csv')  # the same launch without racing copies (bench.py --portfolio 1)
if os.path.exists(p1):
    for name, value in counter_means(p1).items():
        means[name + '_portfolio1'] = value
with open(os.path.join(DST, f'{TAG}_loss_kernel_pmc.csv'), 'w') as f:
    f.write('counter,mean_per_dispatch,dispatches\n')
    for name, (mean, n) in means.items():
        f.write(f'{name},{mean:g},{n}\n')
traffic = {}
for sub, name in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    path = os.path.join(SRC, sub, 'run_counter_collection.csv')
    traffic[name] = counter_means(path)[name]
    rows = [r for r in csv.reader(open(path))]
    keep = [rows[0]] + [r for r in rows[1:] if KERNEL in r[8]]
    with open(os.path.join(DST, f'{TAG}_pmc_{name.lower()}.csv'), 'w', newline='') as f:
        csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(keep)
fetch_kb, n_f = traffic['FETCH_SIZE']
write_kb, n_w = traffic['WRITE_SIZE']
template = os.path.join(DST, f'{TAG}_hbm_traffic.json')
old = json.load(open(template if os.path.exists(template) else os.path.join(REPO, 'profiles', 'r03_hbm_traffic.json')))
old['kernel'] = 'loss_kernel<float,0,false,false>'
import hashlib
old['lib_sha256'] = hashlib.sha256(open(os.path.join(REPO, 'dair_pll_amd', 'csrc', 'libdpll_hip.so'), 'rb').read()).hexdigest()  # the library the passes ran (built here, shipped to the box)
old.update({'FETCH_SIZE_KB_per_launch': round(fetch_kb, 2), 'WRITE_SIZE_KB_per_launch': round(write_kb, 2),
            'traffic_bytes_per_launch': int(round((fetch_kb + write_kb) * 1024)),
            'method': f'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no trace domains), means over {n_f} / {n_w} '
                      'dispatches of python3 bench.py --steps 20 --no-graph; unit KB as reported.'})
json.dump(old, open(os.path.join(DST, f'{TAG}_hbm_traffic.json'), 'w'), indent=1)
text = []
if os.path.exists(os.path.join(SRC, 'stamps.txt')):
  with open(os.path.join(SRC, 'stamps.txt')) as f:
    text = [l for l in f.read().splitlines() if 'amdgpu.ids' not in l]
with open(os.path.join(DST, f'{TAG}_loss_kernel_stamps.txt'), 'w') as f:
    f.write('tools/diag/stamps.py, f32, B=4096, the shipping launch: racing build, 256 four-wave workgroups of 16 items (a wave: 4 items x '
            '4 copies x 4 contact lanes); the stamps are those of wave 0 of every workgroup, its reduce+store segment includes the wait '
            'for the other three waves at the barrier before the shared partial row; units: s_memtime ticks; the stamped build runs '
            '~10 % slower than the shipped one\n' + re.sub(r'np\.(?:int64|float64)\(([^)]*)\)', r'\1', '\n'.join(text)) + '\n')
print(json.dumps(old, indent=1)); print(open(os.path.join(DST, f'{TAG}_loss_kernel_pmc.csv')).read())
