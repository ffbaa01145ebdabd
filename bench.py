"""Headline benchmark: trajectory-steps/s of the ContactNets loss, forward + backward, on the
4096-pair cube-toss batch (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|f64] [--batch B] [--no-graph]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

One "step" = one pass of the hot path over one batch: every (x, x+) pair goes through MultibodyTerms,
the cone solve, the loss and the analytic backward to parameter gradients (dpll_contactnets_loss =
loss kernel + finalize kernel), plus -- for N > 1 -- the single RCCL all-reduce of [loss, gradients].
Inputs are resident in HBM before the timed region.  Batches shard over ranks (weak scaling: 4096 pairs
per GPU); `value` is pairs processed by all ranks per second.

The printed JSON line also carries `roofline` (algorithmic bytes of the loss kernel over its HIP-event
duration, against the 8 TB/s HBM peak) and, on rank 0 at N = 1, `cpu_baseline` (the oracle -- a PyTorch
CPU float64 restatement of the reference path -- timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32 matrix rate: 256 CUs x 256 flop/clk x 2.4 GHz (v_mfma_f32_32x32x2_f32: 64 cyc/SIMD)
VALU_F64_PEAK_TFLOPS = 78.6   # f64 vector FMA rate
BYTES_PER_STEP = {('cube', 'f32'): 2 * 13 * 4 + 4, ('cube', 'f64'): 2 * 13 * 8 + 8,  # read x, x+; write loss
                  ('elbow', 'f32'): 2 * 15 * 4 + 4, ('elbow', 'f64'): 2 * 15 * 8 + 8,
                  ('mesh', 'f32'): 2 * 13 * 4 + 4, ('mesh', 'f64'): 2 * 13 * 8 + 8}


def load_pairs(batch: int, seed: int, workload: str = 'cube'):
    """cube: the 4096 real cube pairs of the reference's data set committed as a fixture (inputs only
    are used here); ranks > 0 take them in a permuted order (weak scaling: every GPU gets the same mix of easy
    and hard pairs), other batch sizes resample them with replacement.
    elbow: the 144 synthetic elbow-toss pairs of the elbow fixture, resampled to the batch size."""
    name = 'elbow_box_literal.npz' if workload == 'elbow' else 'cube_box_4096.npz'
    g = np.load(os.path.join(REPO, 'tests', 'golden', name))
    x, xp = g['x'], g['x_plus']
    if batch == x.shape[0] and seed != 0:
        pick = np.random.default_rng(seed).permutation(batch)  # other ranks: the same pairs in another order
        x, xp = x[pick], xp[pick]
    elif batch != x.shape[0]:
        pick = np.random.default_rng(seed).integers(0, x.shape[0], size=batch)
        x, xp = x[pick], xp[pick]
    return x, xp, float(g['dt'])


def cpu_baseline(x, xp, dt, workload: str = 'cube', budget_s: float = 20.0):
    """Oracle timing (checker code; measured, never shipped): PyTorch CPU fwd+bwd of the restated reference path on
    the box's host threads in float64 (the reference's dtype, dair_pll/inertia.py:96; the oracle's solver is not
    tuned for float32, where it runs 10x slower), 3 warm-up passes then >= 10 timed passes."""
    from oracle import dpll_oracle as O
    threads = torch.get_num_threads()
    sample = min(1024, x.shape[0])
    system = O.OracleSystem(os.path.join(REPO, 'assets', workload + '.urdf'), dt).requires_grad_()
    xs, xps = torch.tensor(x[:sample]), torch.tensor(xp[:sample])

    def one():
        system.zero_grad()
        system.contactnets_loss(xs, xps).mean().backward()

    # all host threads and 16: the restated path is thousands of small batched ops, which many threads slow down
    results = []
    for n_threads in dict.fromkeys((threads, min(16, threads))):
        torch.set_num_threads(n_threads)
        for _ in range(3):
            one()
        reps, t0 = 0, time.perf_counter()
        while reps < 10 or (time.perf_counter() - t0 < budget_s / 2 and reps < 50):
            one()
            reps += 1
        results.append((sample * reps / (time.perf_counter() - t0), n_threads, reps))
    torch.set_num_threads(threads)
    best = max(results)
    return {'value': best[0], 'unit': 'trajectory-steps/s', 'cores': best[1], 'kind': 'port',
            'sample': f'{best[2]} fwd+bwd passes over the first {sample} pairs of the workload after 3 warm-up passes, float64, '
                      f'oracle/dpll_oracle.py (PyTorch CPU); ' +
                      ', '.join(f'{rate:.0f} steps/s with {n} threads' for rate, n, _ in results)}


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=2000)
    parser.add_argument('--warmup', type=int, default=200)
    parser.add_argument('--dtype', choices=['f32', 'f64'], default='f32')
    parser.add_argument('--batch', type=int, default=4096, help='pairs per GPU')
    parser.add_argument('--workload', choices=['cube', 'elbow', 'mesh'], default='cube',
                        help='cube = BASELINE configs[1] (the headline metric); elbow = configs[2]')
    parser.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    parser.add_argument('--steps-per-graph', type=int, default=50,
                        help='steps captured per hipGraph (amortises the ~10 us replay floor); the timed region '
                             'still runs exactly --steps steps, serialised on one stream')
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL)')
    parser.add_argument('--allreduce', choices=['auto', 'peer', 'collective'], default='auto',
                        help='gradient exchange: peer = one-shot kernel over xGMI peer memory, collective = the '
                             'backend all_reduce (RCCL); auto = peer if its start-up self-test passes')
    parser.add_argument('--no-fuse', action='store_true',
                        help='peer transport: run the exchange as its own kernel after the loss launch instead of inside its finalize kernel')
    parser.add_argument('--single-device', action='store_true',
                        help='testing aid: every rank uses cuda:0 (with --backend gloo on a 1-GPU box)')
    args = parser.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    distributed = world > 1
    host_staged = False
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)
            host_staged = True  # gloo collectives cannot be captured (the peer-memory kernel can)

    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.distributed import GradientAllReduce

    dtype = torch.float32 if args.dtype == 'f32' else torch.float64
    x_np, xp_np, dt = load_pairs(args.batch, seed=rank, workload=args.workload)
    torch.manual_seed(0)  # mesh workload: ICNN weights from the reference's init distributions
    urdf_name = {'cube': 'cube.urdf', 'elbow': 'elbow.urdf', 'mesh': 'cube_mesh.urdf'}[args.workload]
    system = MultibodyLearnableSystem({args.workload: os.path.join(REPO, 'assets', urdf_name)}, dt, dtype=dtype,
                                      device=str(device))
    x = torch.tensor(x_np, dtype=dtype, device=device)
    xp = torch.tensor(xp_np, dtype=dtype, device=device)
    reducer = GradientAllReduce(system, transport=args.allreduce, fuse=not args.no_fuse) if distributed else None

    def step():
        system.contactnets_loss_and_grad(x, xp)
        if reducer is not None:
            reducer.all_reduce_mean()

    # eager warm-up (allocates workspace / gradient buffers, builds RCCL communicators)
    for _ in range(3):
        step()
    torch.cuda.synchronize()

    use_graph = not args.no_graph and not (host_staged and reducer is not None and reducer.transport != 'peer')
    graph = None
    per_graph = 1
    if use_graph:
        per_graph = max(d for d in range(1, max(1, args.steps_per_graph) + 1) if args.steps % d == 0 and args.warmup % d == 0) \
            if args.warmup > 0 else max(d for d in range(1, max(1, args.steps_per_graph) + 1) if args.steps % d == 0)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(per_graph):
                    step()
        except Exception as error:  # noqa: BLE001 -- fall back to eager launches, say so in the output
            print(f'[bench] hipGraph capture failed ({error!r}); running eagerly', file=sys.stderr)
            graph = None
            use_graph = False
            per_graph = 1
    run = graph.replay if graph is not None else step

    for _ in range(args.warmup // per_graph):
        run()

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps // per_graph):
        run()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    mesh_ms = None
    if args.workload == 'mesh':
        mesh_ms = system.profile_mesh_kernels(x, xp, reps=50)
        ms_loss, ms_fin = mesh_ms['loss_kernel'], mesh_ms['icnn_reduce']
    else:
        ms_loss, ms_fin = system.profile_loss_kernels(x, xp, reps=200)
    total = system.contactnets_loss_and_grad(x, xp)
    if reducer is not None:
        total = reducer.all_reduce_mean()[:1]
    total_loss = total.item()

    traffic = None
    try:  # HBM bytes per launch from the committed PMC passes (profiles/), only for the configuration they measured
        with open(os.path.join(REPO, 'profiles', 'r01_hbm_traffic.json')) as handle:
            pmc = json.load(handle)
        if (pmc['workload'], pmc['dtype'], pmc['batch']) == (args.workload, args.dtype, args.batch):
            traffic = pmc['traffic_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        pass

    valu_frac = None
    try:  # VALU issue-slot occupancy from the committed SQ counter pass (same configuration only)
        if traffic is not None:
            with open(os.path.join(REPO, 'profiles', 'r01_loss_kernel_pmc.csv')) as handle:
                counters = {row.split(',')[0]: float(row.split(',')[1]) for row in handle.read().splitlines()[1:]}
            valu_frac = counters['SQ_INSTS_VALU'] * 4.0 / (ms_loss * 1e-3 * 2.4e9 * 1024)
    except (OSError, KeyError, ValueError, IndexError):
        pass

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        alg_bytes = BYTES_PER_STEP[(args.workload, args.dtype)] * args.batch
        achieved = alg_bytes / (ms_loss * 1e-3) / 1e9
        line = {
            'metric': 'trajectory-steps/sec (fwd+bwd), batched cube-toss contact sim',
            'value': args.batch * world * args.steps / elapsed,
            'unit': 'trajectory-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype + (' (cone residual accumulated in f64)' if args.dtype == 'f32' else ''),
            'data': ('fixture: 4096 of the 57,812 real cube-toss (x, x+) pairs of the reference data set '
                     '(tests/golden/cube_box_4096.npz, seed 0); ranks > 0 take the same pairs in a permuted order, other batch sizes resample with replacement; URDF-initial parameters')
            if args.workload == 'cube' else 'synthetic elbow tosses (tests/golden/elbow_box_literal.npz) resampled with replacement',
            'config': {'workload': (f'contactnets_cube.urdf, 4 friction contacts, batch={args.batch} per GPU, '
                                    f'fwd+bwd contactnets_loss') if args.workload == 'cube' else
                                   (f'contactnets_elbow.urdf, 8 friction contacts, batch={args.batch} per GPU, '
                                    f'fwd+bwd contactnets_loss (synthetic elbow tosses, resampled)') if args.workload == 'elbow' else
                                   (f'contactnets_cube_mesh.urdf, DeepSupportConvex (ICNN 2x256) geometry, batch={args.batch} per GPU, '
                                    f'fwd+bwd contactnets_loss incl. 67,328 network weights'), 'per_gpu_batch': args.batch,
                       'global_batch': args.batch * world, 'launch': f'hipGraph replay, {per_graph} steps per graph' if use_graph else 'eager',
                       'collective': ('none' if not distributed else
                                      ('one-shot peer-memory all-reduce (xGMI stores + in-order sum) of [loss, gradients] per step, '
                                       + ('inside the finalize kernel of the loss launch' if reducer.fused else 'one kernel after the loss launch'))
                                      if reducer.transport == 'peer' else
                                      f'one {args.backend} all-reduce of [loss, gradients] per step'),
                       'mean_loss': total_loss},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_source': 'profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)' if traffic else None,
                         'kernel': 'loss_kernel', 'kernel_ms': ms_loss, 'finalize_kernel_ms': ms_fin,
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'valu_issue_frac': valu_frac,
                         'note': 'latency/instruction bound by construction (SURVEY 8d): 0.44 MB per launch; valu_issue_frac = '
                                 'SQ_INSTS_VALU per launch (profiles/r01_loss_kernel_pmc.csv) x 4 cycles / (kernel time x 2.4 GHz x '
                                 '1024 SIMDs): the launch is 256 one-wave workgroups, one wave on every fourth SIMD'},
        }
        if mesh_ms is not None:
            # the mesh pipeline is bounded by its four N x 256 x 256 f32 GEMMs (MFMA); the dominant kernel is
            # the slowest of them, its algorithmic work 2 * N * 256 * 256 flop with N = 4 * batch support queries
            gemms = {k: mesh_ms[k] for k in ('icnn_fwd1', 'icnn_fwd2', 'icnn_bwd1', 'icnn_bwd2')}
            dominant = max(gemms, key=gemms.get)
            flops = 2.0 * (4 * args.batch) * 256 * 256
            tflops = flops / (gemms[dominant] * 1e-3) / 1e12
            peak = MFMA_F32_PEAK_TFLOPS if args.dtype == 'f32' else VALU_F64_PEAK_TFLOPS
            line['roofline'] = {'bound': 'mfma', 'achieved': tflops, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tflops / peak,
                                'traffic': None, 'kernel': dominant, 'kernel_ms': gemms[dominant],
                                'algorithmic_flops_per_launch': flops, 'all_kernels_ms': mesh_ms,
                                'pipeline_gemm_tflops': 4 * flops / (sum(gemms.values()) * 1e-3) / 1e12,
                                'note': 'v_mfma_f32_32x32x2_f32 (exact f32); peak = dense f32 matrix rate of MI355X_MICROARCH.md'
                                        if args.dtype == 'f32' else 'float64 path: register-tiled VALU GEMMs (no f64 MFMA form is built)'}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(x_np, xp_np, dt, args.workload if args.workload != 'mesh' else 'cube_mesh')
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
