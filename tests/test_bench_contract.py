"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys, `roofline` and (N = 1)
`cpu_baseline` objects, exactly --steps timed steps.  The GPU test runs the real script as a child process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

REQUIRED = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline'}


def test_bench_helpers_without_a_gpu():
    """algorithmic bytes per step (read x, x+, write the loss), the fixture loader's resampling rule and the
    graph-size rule (largest divisor of --steps that is <= 50, whatever --warmup is)"""
    import bench
    assert bench.bytes_per_step('cube', 'f32') == 108 and bench.bytes_per_step('elbow', 'f64') == 248
    x, xp, dt = bench.load_pairs(4096, 0)
    assert x.shape == (4096, 13) and xp.shape == (4096, 13) and dt == pytest.approx(0.0068, rel=0.2)
    x1, _, _ = bench.load_pairs(4096, 1)  # another rank: the same pairs in another order
    assert not np.array_equal(x, x1) and np.array_equal(np.sort(x[:, 4]), np.sort(x1[:, 4]))
    x2, _, _ = bench.load_pairs(65536, 0)
    assert x2.shape == (65536, 13)
    baseline = json.load(open(os.path.join(REPO, 'BASELINE.json')))
    assert 'trajectory' in baseline['metric'].lower() or 'step' in baseline['metric'].lower()


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5', '--no-configs'],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [line for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert REQUIRED <= set(line), REQUIRED - set(line)
    assert line['n_gpus'] == 1 and line['steps'] == 20 and line['warmup'] == 5 and line['higher_is_better'] is True
    assert line['scaling'] == 'weak' and line['vs_baseline'] is None and 'workload' in line['config']
    assert line['value'] == pytest.approx(4096 * 20 / (line['ms_per_step'] * 20 / 1e3), rel=1e-4)  # (6 significant digits are printed)
    roof = line['roofline']
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(roof) and roof['bound'] == 'hbm' and roof['peak'] == 8000.0
    assert roof['frac'] == pytest.approx(roof['achieved'] / roof['peak'], rel=1e-4) and 0 < roof['frac'] < 1
    assert len(lines[0]) < 6000  # the driver keeps an 8 KB tail of stdout: the line must fit with the configs list (not run here)
    assert line['config']['repeats'] >= 5 and 'measured in total' in line['config']['timing']
    cpu = line['cpu_baseline']
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(cpu) and cpu['kind'] in ('port', 'reference') and cpu['value'] > 0
    assert line['value'] > 1000 * cpu['value']  # (orders of magnitude, not a target: the roofline fraction is the measure)


def test_bare_multi_gpu_command_line_launches_its_own_ranks():
    """`python bench.py --gpus 2` as the driver types it (no torch.distributed.run in front, no WORLD_SIZE): the process
    becomes the launcher, two ranks rendezvous on 127.0.0.1 and count themselves with one all-reduce of the backend
    (gloo here: no GPU), rank 0 prints one JSON line, the exit code is the ranks' """
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--launch-check', '--backend', 'gloo'],
                         capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1 and lines[0] == {'launch_check': True, 'n_gpus': 2, 'rccl_ranks': 2, 'backend': 'gloo'}
    # a rank that fails makes the launcher fail
    bad = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--launch-check', '--backend', 'no-such-backend'],
                         capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert bad.returncode != 0


@pytest.mark.gpu
def test_bare_two_rank_bench_on_one_device():
    """the N > 1 path end to end through the bare command line, two ranks sharing cuda:0 (gloo collectives, the peer-memory
    exchange as the alternative route): one contract line, whole-job value, the ranks counted by a real all-reduce"""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--single-device',
                          '--steps', '20', '--warmup', '5', '--min-timed-s', '0.05'],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1
    line = lines[0]
    assert REQUIRED <= set(line) and line['n_gpus'] == 2 and line['steps'] == 20
    assert line['config']['rccl_ranks'] == 2 and line['config']['global_batch'] == 8192
    assert 'all-reduce' in line['config']['collective'] and 'gloo' in line['config']['collective']  # the backend's collective is the reported route
    assert line['value'] == pytest.approx(8192 * 20 / (line['ms_per_step'] * 20 / 1e3), rel=1e-4)
    assert 'cpu_baseline' not in line and 'configs' not in line


def test_eight_rank_launch_check_on_cpu():
    """the command the driver's scaling tier ends with -- `python bench.py --gpus 8` -- as far as it goes without GPUs: the
    launcher starts eight ranks, they rendezvous on 127.0.0.1 and count themselves with one all-reduce (gloo), rank 0
    prints the one line (the 8-rank launch must not see its first execution on the 8-GPU node)"""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '8', '--single-device', '--backend', 'gloo', '--launch-check'],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith('{')]
    assert lines == [{'launch_check': True, 'n_gpus': 8, 'rccl_ranks': 8, 'backend': 'gloo'}]


@pytest.mark.gpu
def test_rccl_allreduce_inside_graph_capture_on_one_gpu():
    """Dress rehearsal of the multi-GPU step on ONE GPU: a world-size-1 `nccl` (= RCCL) process group
    on cuda:0, bench.py's own timed(reducer) with the collective transport -- communicator set-up, the all-reduce of
    [loss | gradients] captured in the step's hipGraph and replayed, barrier + max-over-ranks fences.  Capture must have
    worked, or the line must say that the launches were eager and why (never a silent fall-back)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--rehearse', '--steps', '20', '--warmup', '5',
                          '--min-timed-s', '0.05', '--no-configs', '--no-cpu-baseline', '--ref-value', '2.0e8'],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1
    line = lines[0]
    assert REQUIRED <= set(line) and line['n_gpus'] == 1 and line['value'] > 0
    config = line['config']
    assert config['rccl_ranks'] == 1 and 'nccl all-reduce' in config['collective'] and 'rehearsal' in config
    launch = config['launch']
    print('rehearsal launch:', launch, '| ms_per_step', line['ms_per_step'])
    assert launch.startswith('hipGraph replay') or 'hipGraph capture failed' in launch, launch
    # the same loss as without the process group (a SUM over one rank), and the efficiency figure against the given N = 1 value
    assert config['mean_loss'] == pytest.approx(3.555887e-04, rel=1e-4)
    eff = config['weak_scaling_vs_ref']
    assert eff['ref_source'] == '--ref-value' and eff['efficiency'] == pytest.approx(line['value'] / 2.0e8, rel=1e-4)
