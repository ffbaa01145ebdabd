"""Headline benchmark: trajectory-steps/s of the ContactNets loss, forward + backward, on the
4096-pair cube-toss batch (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|f64] [--batch B] [--workload cube|elbow|mesh|simulate]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

One "step" = one pass of the hot path over one batch: every (x, x+) pair goes through MultibodyTerms,
the cone solve, the loss and the analytic backward to parameter gradients (dpll_contactnets_loss =
loss kernel + finalize kernel), plus -- for N > 1 -- the single all-reduce of [loss, gradients].
Inputs are resident in HBM before the timed region.  Batches shard over ranks (weak scaling: 4096 pairs
per GPU); `value` is pairs processed by all ranks per second.  The timed region (exactly --steps steps
between barrier + synchronize fences, MAX over ranks) is repeated --repeats times and the MEDIAN repeat
is reported (all of them are listed under config.repeat_ms).

The printed JSON line also carries `roofline` (algorithmic bytes of the loss kernel over its HIP-event
duration, against the 8 TB/s HBM peak), on rank 0 at N = 1 `cpu_baseline` (the oracle -- a PyTorch CPU
float64 restatement of the reference path -- timed on this host's cores on a bounded sample) and
`configs`: the other BASELINE.json configurations measured in the same run (elbow 4096, mesh 4096, cube f64,
65,536 pairs in f32 / f64, fused rollouts; plus the elbow with a learned mesh on both links and two general-build models:
one with a body-body pair, one with a prismatic joint and turned frames), each with its own value / kernel time / roofline.
"""
import argparse
import glob
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
N_X = {'cube': 13, 'elbow': 15, 'mesh': 13, 'elbow_mesh': 15, 'clasp': 17, 'slider': 17, 'clasp_mesh': 17, 'two_cubes': 26, 'chain6': 23}
ELEM = {'f32': 4, 'f64': 8}
URDF = {'cube': 'cube.urdf', 'elbow': 'elbow.urdf', 'mesh': 'cube_mesh.urdf', 'elbow_mesh': 'elbow_mesh.urdf', 'clasp': 'clasp.urdf',
        'slider': 'slider.urdf', 'clasp_mesh': 'clasp_mesh.urdf',  # (clasp: a box-box body-body candidate; slider: a prismatic joint, turned frames -- the general build)
        # the forest build (csrc/dpll_forest.hip): two models in one system; a five-joint chain with ten body-body candidates
        'two_cubes': {'cube_a': 'cube.urdf', 'cube_b': 'cube.urdf'}, 'chain6': 'chain6.urdf'}
FIXTURE = {'cube': 'cube_box_4096.npz', 'elbow': 'elbow_box_4096.npz', 'mesh': 'cube_box_4096.npz', 'elbow_mesh': 'elbow_box_4096.npz',
           'clasp': 'clasp_literal.npz', 'slider': 'slider_literal.npz', 'clasp_mesh': 'clasp_mesh_literal.npz',
           'two_cubes': 'two_cubes_literal.npz', 'chain6': 'chain6_literal.npz'}
# (clasp_mesh: two learned shapes whose pair is a collision candidate -- the reference's own body-body case: GJK / EPA + ICNN)
# (what the fixtures are: DESIGN.md section 2; ranks > 0 take the same pairs permuted, other batch sizes resample with replacement)
DATA = {'cube': 'fixture: 4096 of the 57,812 real cube-toss pairs of the reference data set (tests/golden/cube_box_4096.npz), URDF-initial parameters',
        'elbow': 'fixture: 4096 seeded pairs of 40 synthetic elbow tosses (tests/golden/elbow_box_4096.npz), URDF-initial parameters',
        'mesh': 'the cube fixture pairs; ICNN weights from the reference init distributions, seed 0'}


def bytes_per_step(workload: str, dtype: str) -> int:
    """loss fwd+bwd: read x, x+, write loss (the fused path skips the loss write but the figure is the contract's)"""
    return (2 * N_X[workload] + 1) * ELEM[dtype]


def load_pairs(batch: int, seed: int, workload: str = 'cube'):
    g = np.load(os.path.join(REPO, 'tests', 'golden', FIXTURE[workload]))
    x, xp = g['x'], g['x_plus']
    if batch == x.shape[0] and seed != 0:
        pick = np.random.default_rng(seed).permutation(batch)  # other ranks: the same pairs in another order
        x, xp = x[pick], xp[pick]
    elif batch != x.shape[0]:
        pick = np.random.default_rng(seed + 1).integers(0, x.shape[0], size=batch)  # with replacement, seed 1 on rank 0
        x, xp = x[pick], xp[pick]
    return x, xp, float(g['dt'])


def cpu_baseline(x, xp, dt, workload: str = 'cube', budget_s: float = 24.0):
    """Oracle timing (checker code; measured, never shipped): PyTorch CPU fwd+bwd of the restated reference path over the
    WHOLE workload batch on the box's host threads, float64 (the reference's dtype, dair_pll/inertia.py:96), 2 warm-up
    passes then timed passes for about `budget_s` seconds (at least 4).  Beside it the same graph with the cone solve
    replaced by a zero-force stub ("graph only": what BASELINE.md section 2 measured for the reference's own torch graph,
    58 k steps/s on 8 cores), so that the two can be set against each other.  The restated path is thousands of small
    batched ops: more than 16 threads slow it down (round 2: 1.2 k steps/s with 128 threads, 8.3 k with 16)."""
    from oracle import dpll_oracle as O
    available = torch.get_num_threads()
    threads = min(16, available)
    sample = x.shape[0]
    system = O.OracleSystem(os.path.join(REPO, 'assets', URDF[workload]), dt).requires_grad_()
    xs, xps = torch.tensor(x[:sample]), torch.tensor(xp[:sample])

    def one():
        system.zero_grad()
        system.contactnets_loss(xs, xps).mean().backward()

    def rate(seconds, least):
        for _ in range(2):
            one()
        reps, t0 = 0, time.perf_counter()
        while reps < least or time.perf_counter() - t0 < seconds:
            one()
            reps += 1
        return sample * reps / (time.perf_counter() - t0), reps

    torch.set_num_threads(threads)
    full, reps = rate(0.7 * budget_s, 4)
    solver = O.sap_solve
    try:  # graph only: the same autograd graph around a solver that returns no force
        O.sap_solve = lambda J, q, eps, *args, **kwargs: torch.zeros_like(q)
        graph_only, graph_reps = rate(0.15 * budget_s, 3)
    finally:
        O.sap_solve = solver
        torch.set_num_threads(available)
    return {'value': full, 'unit': 'trajectory-steps/s', 'cores': threads, 'kind': 'port', 'graph_only_value': graph_only,
            'sample': f'{reps} fwd+bwd passes over all {sample} pairs of the workload after 2 warm-up passes, float64, '
                      f'oracle/dpll_oracle.py (PyTorch CPU, {threads} of {available} host threads); graph_only_value: '
                      f'{graph_reps} passes with the cone solve stubbed out (cf. BASELINE.md section 2: 58 k steps/s for the '
                      f'reference graph, 8 cores)'}


def compact(value, digits: int = 6):
    """floats rounded to `digits` significant digits, recursively: the whole JSON line has to survive the driver's 8 KB
    tail window (a 14 KB line lost its first configs)"""
    if isinstance(value, float):
        return float(f'{value:.{digits}g}') if np.isfinite(value) else None
    if isinstance(value, dict):
        return {k: compact(v, digits) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [compact(v, digits) for v in value]
    return value


def library_sha256() -> str:
    import hashlib
    from dair_pll_amd import _capi
    with open(_capi.LIB_PATH, 'rb') as handle:
        return hashlib.sha256(handle.read()).hexdigest()


def newest_profile(pattern: str):
    """profiles/rNN_<pattern>, newest round first"""
    found = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r[0-9][0-9]_' + pattern)), reverse=True)
    return found[0] if found else None


class Timer:
    """Exactly `steps` steps between fences, `repeats` times; hipGraph replay of `per_graph` steps at a time."""

    def __init__(self, step, steps: int, warmup: int, use_graph: bool, steps_per_graph: int, fence, max_over_ranks):
        self.step, self.steps, self.fence, self.max_over_ranks = step, steps, fence, max_over_ranks
        self.graph, self.per_graph = None, 1
        self.capture_error = None
        if use_graph:
            # the largest divisor of --steps that is <= --steps-per-graph, whatever --warmup is (the warm-up replays the
            # same graph and may run a few steps more than asked: a single-step graph pays a ~8.5 us replay gap per step)
            self.per_graph = max(d for d in range(1, max(1, min(steps, steps_per_graph)) + 1) if steps % d == 0)
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step()
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(self.per_graph):
                        step()
                self.graph = graph
            except Exception as error:  # noqa: BLE001 -- fall back to eager launches, say so in the output
                self.capture_error = repr(error)
                self.graph, self.per_graph = None, 1
        self.run = self.graph.replay if self.graph is not None else step
        for _ in range(-(-warmup // self.per_graph)):
            self.run()

    def agree_on_graph(self, dist, device) -> None:
        """every rank must run the same launch sequence: if capture failed anywhere, all ranks go eager"""
        flag = torch.tensor([1 if self.graph is not None else 0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() == 0 and self.graph is not None:
            self.graph, self.per_graph, self.run = None, 1, self.step

    def measure(self, repeats: int, min_total_s: float = 0.0, max_repeats: int = 5000):
        """The timed region -- exactly `steps` steps between two fences, MAX over ranks -- at least `repeats` times and until
        `min_total_s` seconds have been measured in total (a 20-step region of the headline workload lasts 0.4 ms: one of
        them alone is below what any outside observer of the GPU can see); the MEDIAN region is the reported time.  The
        repeat count is decided on the max-over-ranks times, which every rank holds alike."""
        times = []
        while len(times) < repeats or (sum(times) < min_total_s and len(times) < max_repeats):
            self.fence()
            t0 = time.perf_counter()
            for _ in range(self.steps // self.per_graph):
                self.run()
            self.fence()
            times.append(self.max_over_ranks(time.perf_counter() - t0))
        return float(np.median(times)), times

    @property
    def launch(self) -> str:
        if self.graph is not None:
            return f'hipGraph replay, {self.per_graph} steps per graph'
        return 'eager' + (f' (hipGraph capture failed: {self.capture_error})' if self.capture_error else '')


# dpll_solver_opts_t.mesh_gemm: the forms of the float32 ICNN GEMM kernels (library default: 4; None below = that default)
MESH_GEMM_FORMS = {0: 'ICNN GEMMs: exact f32 MFMA, pipelined', 1: 'ICNN GEMMs: f32 MFMA, the 8-wave kernels', 2: 'ICNN GEMMs on the bf16 matrix cores, 2 planes',
                   3: 'ICNN GEMMs on the bf16 matrix cores, 3 planes',
                   4: 'ICNN GEMMs on the fp16 matrix cores, 2 planes with the low one scaled by 2^11: f32-grade products'}


def loss_roofline(system, workload, dtype, batch, x, xp, mesh_gemm=None):
    """roofline object of the dominant kernel, measured live with HIP events on the launch stream"""
    alg_bytes = bytes_per_step(workload, dtype) * batch
    if workload in ('elbow_mesh', 'clasp', 'slider', 'clasp_mesh', 'two_cubes', 'chain6'):
        return None  # no per-kernel utility for these pipelines: run_loss_config prices the whole step
    if workload == 'mesh':
        form = 4 if mesh_gemm is None else mesh_gemm  # (the library's default: dpll_solver_opts_t.mesh_gemm)
        mesh_ms = system.profile_mesh_kernels(x, xp, reps=50)
        # the mesh pipeline is bounded by its four N x 256 x 256 f32 GEMMs (MFMA); the dominant kernel is
        # the slowest of them, its algorithmic work 2 * N * 256 * 256 flop with N = 4 * batch support queries
        gemms = {k: mesh_ms[k] for k in ('icnn_fwd1', 'icnn_fwd2', 'icnn_bwd1', 'icnn_bwd2')}
        dominant = max(gemms, key=gemms.get)
        flops = 2.0 * (4 * batch) * 256 * 256
        tflops = flops / (gemms[dominant] * 1e-3) / 1e12
        # (split-bf16 forms: still priced against the f32 matrix rate -- the algorithmic work is the f32 GEMM; the bf16
        # cores do 3 or 6 products per f32 product, which is how the fraction can pass the f32 form's ceiling)
        peak = MFMA_F32_PEAK_TFLOPS if dtype == 'f32' else VALU_F64_PEAK_TFLOPS
        return {'bound': 'mfma', 'achieved': tflops, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tflops / peak,
                'traffic': None, 'kernel': dominant, 'kernel_ms': gemms[dominant],
                'algorithmic_flops_per_launch': flops, 'all_kernels_ms': mesh_ms,
                'pipeline_gemm_tflops': 4 * flops / (sum(gemms.values()) * 1e-3) / 1e12,
                'note': 'v_mfma_f32_32x32x2_f32 (exact f32); peak = dense f32 matrix rate of MI355X_MICROARCH.md.  That rate needs the '
                        'vector ALU to itself: the f32 MFMA runs on its multipliers, so every VALU / LDS instruction of the fused fill and '
                        'epilogue adds to the MFMA time (profiles/r05_mfma_fill.txt, r05_mfma_step.txt), and the chip runs these kernels '
                        'at ~2.15 GHz (matrix floor of one GEMM at 4096 pairs: 15.2 us)'
                        if dtype == 'f32' and form in (0, 1) else
                        (MESH_GEMM_FORMS[form] + ' (csrc/dpll_mesh_bf16.hpp, dpll_icnn_pipe.hip); priced against the f32 matrix rate: the '
                         'algorithmic work is the f32 GEMM' if dtype == 'f32' else
                         'float64 path: register-tiled VALU GEMMs (no f64 MFMA form is built)')}
    passes = [system.profile_loss_kernels(x, xp, reps=200) for _ in range(3)]  # (HIP events on the launch stream, 200 launches each)
    ms_loss, ms_fin = sorted(p[0] for p in passes)[1], sorted(p[1] for p in passes)[1]
    achieved = alg_bytes / (ms_loss * 1e-3) / 1e9
    traffic = valu_frac = useful_frac = source = lib_match = None
    path = newest_profile('hbm_traffic.json')
    try:  # HBM bytes per launch / VALU issue slots from the committed PMC passes, only for the configuration they measured
        with open(path) as handle:
            pmc = json.load(handle)
        if (pmc['workload'], pmc['dtype'], pmc['batch']) == (workload, dtype, batch):
            traffic, source = pmc['traffic_bytes_per_launch'], os.path.relpath(path, REPO)
            # counters of one binary must not be mixed with timings of another: the issue fraction is only
            # formed when the profiled library IS the loaded one; clock and SIMD count come from the device
            lib_match = pmc.get('lib_sha256') == library_sha256()
            if lib_match:
                with open(path.replace('hbm_traffic.json', 'loss_kernel_pmc.csv')) as handle:
                    counters = {row.split(',')[0]: float(row.split(',')[1]) for row in handle.read().splitlines()[1:]}
                props = torch.cuda.get_device_properties(torch.cuda.current_device())
                # (clock_rate in kHz where this torch build reports it; else the 2.4 GHz peak engine clock of the part)
                simds, hz = 4 * props.multi_processor_count, (getattr(props, 'clock_rate', 0) or 2.4e6) * 1e3
                valu_frac = counters['SQ_INSTS_VALU'] * 4.0 / (ms_loss * 1e-3 * hz * simds)
                # the instructions one solve per item executes (the same launch without racing copies: portfolio = 1) over the
                # time of the launch WITH copies: what the copies' discarded work must not be credited with
                if 'SQ_INSTS_VALU_portfolio1' in counters:
                    useful_frac = counters['SQ_INSTS_VALU_portfolio1'] * 4.0 / (ms_loss * 1e-3 * hz * simds)
    except (OSError, KeyError, ValueError, IndexError, TypeError, AttributeError):
        pass
    return {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': traffic,
            'traffic_source': f'{source} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)' if traffic else None,
            'traffic_of_loaded_library': lib_match,
            'kernel': 'loss_kernel', 'kernel_ms': ms_loss, 'finalize_kernel_ms': ms_fin,
            'algorithmic_bytes_per_launch': alg_bytes, 'valu_issue_frac': valu_frac, 'useful_valu_issue_frac': useful_frac,
            'note': 'latency bound by construction (SURVEY 8d, DESIGN 5): independent 6-7-dimensional cone solves, 0.44 MB per launch'}


def build_system(workload, dtype_name, dt, device):
    from dair_pll_amd import MultibodyLearnableSystem
    torch.manual_seed(0)  # mesh workload: ICNN weights from the reference's init distributions
    dtype = torch.float32 if dtype_name == 'f32' else torch.float64
    urdfs = URDF[workload] if isinstance(URDF[workload], dict) else {workload: URDF[workload]}
    return MultibodyLearnableSystem({k: os.path.join(REPO, 'assets', v) for k, v in urdfs.items()}, dt, dtype=dtype, device=str(device))


def run_loss_config(workload, dtype_name, batch, steps, warmup, repeats, device, use_graph=True, steps_per_graph=50, mesh_gemm=None):
    """one single-GPU configuration of the loss path: value, step time, roofline"""
    dtype = torch.float32 if dtype_name == 'f32' else torch.float64
    x_np, xp_np, dt = load_pairs(batch, 0, workload)
    system = build_system(workload, dtype_name, dt, device)
    if mesh_gemm is not None:  # another form of the ICNN GEMM kernels than the library's default
        system.set_solver(mesh_gemm=mesh_gemm)
    x, xp = torch.tensor(x_np, dtype=dtype, device=device), torch.tensor(xp_np, dtype=dtype, device=device)
    step = lambda: system.contactnets_loss_and_grad(x, xp)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    timer = Timer(step, steps, warmup, use_graph, steps_per_graph, torch.cuda.synchronize, lambda t: t)
    elapsed, times = timer.measure(repeats)
    roof = loss_roofline(system, workload, dtype_name, batch, x, xp, mesh_gemm)
    if roof is None:  # whole-step roofline (an upper bound on every kernel's time, so a lower bound on its fraction)
        step_ms = elapsed / steps * 1e3
        if workload == 'elbow_mesh':  # 2 networks x 4 GEMMs of (4 batch) x 256 x 256 (per link)
            tflops = 8 * 2.0 * (4 * batch) * 256 * 256 / (step_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'achieved': tflops, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tflops / MFMA_F32_PEAK_TFLOPS,
                    'kernel': 'whole step: 8 GEMM launches (one ICNN per link) + item kernel + reductions', 'kernel_ms': step_ms}
        elif workload == 'clasp_mesh':  # 2 networks x 4 GEMMs of (5 batch) x 256 x 256: 4 ground queries + the pair's per item
            tflops = 8 * 2.0 * (5 * batch) * 256 * 256 / (step_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'achieved': tflops, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tflops / MFMA_F32_PEAK_TFLOPS,
                    'kernel': 'whole step: hull extraction + GJK/EPA query kernel + 8 GEMM launches + general item kernel + reductions', 'kernel_ms': step_ms}
        else:
            gbs = bytes_per_step(workload, dtype_name) * batch / (step_ms * 1e-3) / 1e9
            kernel = ('forest_loss_kernel + row fold + finalize (forest build: one wave per item, blocks in LDS)' if workload in ('two_cubes', 'chain6')
                      else 'gen_loss_kernel + row fold + finalize (general build: one lane per contact slot)')
            roof = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'kernel': kernel, 'kernel_ms': step_ms}
    if mesh_gemm is not None:
        workload = f'{workload} ({MESH_GEMM_FORMS[mesh_gemm]})'
    return {'workload': workload, 'dtype': dtype_name, 'batch': batch, 'value': batch * steps / elapsed,
            'unit': 'trajectory-steps/s', 'ms_per_step': elapsed / steps * 1e3, 'steps': steps, 'launch': timer.launch,
            'racing_copies': system.racing_copies(batch), 'kernel_ms': roof['kernel_ms'], 'mean_loss': system.contactnets_loss_and_grad(x, xp).item(),
            'roofline': {k: roof[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel') if k in roof}}


def run_simulate_config(workload, dtype_name, batch, horizon, repeats, device, mesh_gemm=None):
    """fused rollouts (dpll_simulate: the time loop inside the kernel): one step = one VelocityIntegrator.step of one
    trajectory; algorithmic bytes per step = read x + write x+ (cube 104 B in f32)"""
    dtype = torch.float32 if dtype_name == 'f32' else torch.float64
    x_np, _, dt = load_pairs(batch, 0, workload)
    system = build_system(workload, dtype_name, dt, device)
    if mesh_gemm is not None:
        system.set_solver(mesh_gemm=mesh_gemm)
    x0 = torch.tensor(x_np, dtype=dtype, device=device).unsqueeze(-2)
    carry = torch.zeros((batch, 1), device=device)
    with torch.no_grad():
        for _ in range(2):
            system.simulate(x0, carry, horizon)
        times = []
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(repeats):
            torch.cuda.synchronize()
            start.record()
            system.simulate(x0, carry, horizon)
            end.record()
            torch.cuda.synchronize()
            times.append(start.elapsed_time(end))  # ms; the kernel is launched on torch's current stream
    ms = float(np.median(times))
    if workload == 'mesh':  # per step two forward GEMMs of (4 batch) x 256 x 256 (the support points follow the state)
        tflops = 2 * 2.0 * (4 * batch) * 256 * 256 * horizon / (ms * 1e-3) / 1e12
        return {'workload': f'simulate ({workload}, {horizon} steps per call)' + (f', {MESH_GEMM_FORMS[mesh_gemm]}' if mesh_gemm is not None else ''),
                'dtype': dtype_name, 'batch': batch,
                'value': batch * horizon / (ms * 1e-3), 'unit': 'trajectory-steps/s (forward only)', 'ms_per_step': ms / horizon,
                'kernel_ms': ms, 'launch': 'dpll_simulate_mesh: weights prepared once, 4 kernels per step enqueued by the library',
                'roofline': {'bound': 'mfma', 'achieved': tflops, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': tflops / MFMA_F32_PEAK_TFLOPS, 'kernel': 'whole step: icnn_fwd1 + icnn_fwd2 + simulate_kernel'}}
    alg = 2 * N_X[workload] * ELEM[dtype_name] * batch * horizon
    achieved = alg / (ms * 1e-3) / 1e9
    return {'workload': f'simulate ({workload}, {horizon} steps per launch)', 'dtype': dtype_name, 'batch': batch,
            'value': batch * horizon / (ms * 1e-3), 'unit': 'trajectory-steps/s (forward only)', 'ms_per_step': ms / horizon,
            'kernel_ms': ms, 'launch': 'one simulate_kernel launch per rollout', 'racing_copies': system.racing_copies(batch, rollout=True),
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'kernel': 'simulate_kernel'}}


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of `python -m torch.distributed.run`
    (the command the driver documents), relay what they print, return their worst exit code.  The parent never touches
    the GPU (nothing is exec'ed either: a process that has initialised the GPU must not be replaced)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL and the peer exchange both need it on this pool
    env.setdefault('OMP_NUM_THREADS', '1')
    command = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(command, env=env, check=False).returncode


def run_train_config(dtype_name, batch, steps, device, fused, workload='cube'):
    """One training step of the loop the kernel lives in (experiment.py:332-363 with the optimizer of :213-228): the mean
    ContactNets loss of a 4096-pair batch, its gradients and the Adam update of every parameter -- 20 steps per hipGraph,
    median of 5 repeats of `steps` steps.  `fused`: the update is done by the finalize kernel of the loss launch
    (dpll_contactnets_train_step: two launches per step); otherwise torch.optim.Adam(capturable=True)'s kernels follow it.
    (The batch is fixed: the shuffled gather of a real epoch is two index_select launches more, ContactNetsTrainer.)"""
    from dair_pll_amd.system import FusedAdamState
    dtype = torch.float32 if dtype_name == 'f32' else torch.float64
    x_np, xp_np, dt = load_pairs(batch, 0, workload)
    system = build_system(workload, dtype_name, dt, device)
    x, xp = torch.tensor(x_np, dtype=dtype, device=device), torch.tensor(xp_np, dtype=dtype, device=device)
    if fused:
        adam = FusedAdamState(lr=1e-3)
        step = lambda: system.contactnets_train_step(x, xp, adam)
    else:
        optimizer = torch.optim.Adam(system.parameters(), lr=1e-3, capturable=True)

        def step():
            total = system.contactnets_loss_and_grad(x, xp)
            optimizer.step()
            return total
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    timer = Timer(step, steps, 20, True, 20, torch.cuda.synchronize, lambda t: t)
    elapsed, _ = timer.measure(5)
    return {'workload': f'train_step ({workload}, ' + ('fused Adam' if fused else 'torch.optim.Adam') + ')',
            'dtype': dtype_name, 'batch': batch, 'value': steps * batch / elapsed, 'optimizer_steps_per_s': steps / elapsed,
            'ms_per_step': elapsed / steps * 1e3, 'launch': timer.launch, 'mean_loss': step().item()}


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=2000)
    parser.add_argument('--warmup', type=int, default=200)
    parser.add_argument('--repeats', type=int, default=5, help='repeats of the timed region; the median is reported')
    parser.add_argument('--dtype', choices=['f32', 'f64'], default='f32')
    parser.add_argument('--batch', type=int, default=4096, help='pairs per GPU')
    parser.add_argument('--workload', choices=['cube', 'elbow', 'mesh'], default='cube',
                        help='cube = BASELINE configs[1] (the headline metric); elbow = configs[2]; mesh = configs[3]')
    parser.add_argument('--mesh-gemm', type=int, choices=[0, 1, 2, 3, 4], default=None,
                        help='mesh workload: form of the ICNN GEMM kernels (default: the library\'s, 4 = two fp16 planes, f32-grade; 0 = f32 MFMA, 1 = its '
                             '8-wave kernels, 2 / 3 = bf16 matrix cores on 2 / 3 bf16 planes)')
    parser.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    parser.add_argument('--portfolio', type=int, default=None,
                        help='diagnostic: racing copies of the cone solve per item (1 = none; default: what the library picks) -- the '
                             'counter pass behind roofline.useful_valu_issue_frac runs with 1')
    parser.add_argument('--steps-per-graph', type=int, default=50,
                        help='steps captured per hipGraph (amortises the ~10 us replay floor); the timed region '
                             'still runs exactly --steps steps, serialised on one stream')
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-configs', action='store_true', help='skip the other BASELINE configurations (N = 1 only)')
    parser.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL)')
    parser.add_argument('--allreduce', choices=['auto', 'peer', 'collective'], default='collective',
                        help='gradient exchange of the REPORTED value: collective = the backend all_reduce (RCCL, what north_star '
                             'names; the default), peer = one-shot kernel over xGMI peer memory, auto = peer if its start-up '
                             'self-test passes.  With collective the peer route is timed too and reported as collective_alt')
    parser.add_argument('--no-fuse', action='store_true',
                        help='peer transport: run the exchange as its own kernel after the loss launch instead of inside its finalize kernel')
    parser.add_argument('--single-device', action='store_true',
                        help='testing aid: every rank uses cuda:0 (with --backend gloo on a 1-GPU box)')
    parser.add_argument('--launch-check', action='store_true',
                        help='testing aid for the N > 1 launcher: the ranks rendezvous, count themselves with one all-reduce of '
                             'the backend and rank 0 prints a JSON line; no GPU is touched')
    parser.add_argument('--rehearse', action='store_true',
                        help='N = 1 only: run the N > 1 code path on one GPU -- a world-size-1 process group of --backend (nccl = RCCL: '
                             'communicator set-up, the all-reduce of [loss, gradients] inside hipGraph capture and replay, the barrier / '
                             'max-over-ranks fences) -- so that the first multi-GPU run is not the first execution of that path')
    parser.add_argument('--ref-value', type=float, default=None,
                        help='N > 1: the N = 1 value (trajectory-steps/s) to set this run against; config.weak_scaling_vs_ref = value / '
                             '(N * ref).  Default: the newest profiles/rNN_bench_line.json of the same workload, dtype and batch')
    parser.add_argument('--min-timed-s', type=float, default=0.5,
                        help='the timed region (exactly --steps steps) is repeated until at least this much time has been '
                             'measured in total (and at least --repeats times); the median region is reported')
    args = parser.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # bare `python bench.py --gpus N`: this process becomes the launcher -- it has made no GPU call and makes none -- and
        # the N ranks are children started through torch.distributed.run (one per GPU, rendezvous on 127.0.0.1)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.launch_check:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo' if args.backend == 'nccl' and not torch.cuda.is_available() else args.backend)
        probe = torch.ones(1, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(probe)
        if rank == 0:
            print(json.dumps({'launch_check': True, 'n_gpus': world, 'rccl_ranks': int(probe.item()), 'backend': dist.get_backend()}),
                  flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    distributed = world > 1 or args.rehearse
    host_staged = False
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1 and 'MASTER_PORT' not in os.environ:  # the rehearsal without a launcher: a rendezvous of one
            import socket
            with socket.socket() as sock:
                sock.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)
            host_staged = True  # gloo collectives cannot be captured (the peer-memory kernel can)

    from dair_pll_amd.distributed import GradientAllReduce

    dtype = torch.float32 if args.dtype == 'f32' else torch.float64
    x_np, xp_np, dt = load_pairs(args.batch, seed=rank, workload=args.workload)
    system = build_system(args.workload, args.dtype, dt, device)
    if args.mesh_gemm is not None and args.workload == 'mesh':
        system.set_solver(mesh_gemm=args.mesh_gemm)
    if args.portfolio is not None:
        system.set_solver(portfolio=args.portfolio)
    x = torch.tensor(x_np, dtype=dtype, device=device)
    xp = torch.tensor(xp_np, dtype=dtype, device=device)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(elapsed):
        if not distributed:
            return elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    def timed(reducer):
        def step():
            system.contactnets_loss_and_grad(x, xp)
            if reducer is not None:
                reducer.all_reduce_mean()
        for _ in range(3):  # eager warm-up (allocates workspace / gradient buffers, builds RCCL communicators)
            step()
        torch.cuda.synchronize()
        use_graph = not args.no_graph and not (host_staged and reducer is not None and reducer.transport != 'peer')
        timer = Timer(step, args.steps, args.warmup, use_graph, args.steps_per_graph, fence, max_over_ranks)
        if distributed:
            timer.agree_on_graph(dist, device)
        elapsed, times = timer.measure(args.repeats, args.min_timed_s)
        if reducer is not None:
            reducer.check_healthy()  # raises if a peer exchange timed out: never report numbers of a broken exchange
        return timer, elapsed, times

    reducer = alt = None
    rccl_ranks = None
    error = None

    def describe(red):
        if red.transport == 'peer':
            return ('one-shot peer-memory all-reduce (xGMI stores + in-order sum) of [loss, gradients] per step, '
                    + ('inside the finalize kernel of the loss launch' if red.fused else 'one kernel after the loss launch'))
        return f'one {args.backend} all-reduce of [loss, gradients] per step'

    try:
        if distributed:
            reducer = GradientAllReduce(system, transport=args.allreduce, fuse=not args.no_fuse)
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe)  # an actual collective of the backend (RCCL for nccl): counts the ranks that took part
            rccl_ranks = int(probe.item())
        timer, elapsed, times = timed(reducer)
        if distributed:
            # the other route in the same run: north_star names a single RCCL all-reduce (the reported default); the
            # hand-written peer exchange is its alternative until a multi-GPU run has shown it sound (and vice versa)
            try:
                other = GradientAllReduce(system, transport='peer' if reducer.transport == 'collective' else 'collective',
                                          fuse=not args.no_fuse)
                if other.transport != reducer.transport:
                    alt_timer, alt_elapsed, _ = timed(other)
                    alt = {'collective': describe(other), 'launch': alt_timer.launch, 'ms_per_step': alt_elapsed / args.steps * 1e3,
                           'value': args.batch * world * args.steps / alt_elapsed}
            except Exception as exc:  # noqa: BLE001 -- the alternative route is informational
                alt = {'collective': 'alternative route failed', 'error': repr(exc)}
            system._fused_ar = reducer.peer._ar if reducer.fused else None  # back to the reported route
    except Exception as exc:  # noqa: BLE001 -- one JSON error line, non-zero exit
        error = repr(exc)
    if error is not None:
        if rank == 0:
            print(json.dumps({'metric': 'trajectory-steps/sec (fwd+bwd), batched cube-toss contact sim', 'value': None,
                              'n_gpus': world, 'error': error}), flush=True)
        if distributed:
            dist.destroy_process_group()
        raise SystemExit(1)

    roof = loss_roofline(system, args.workload, args.dtype, args.batch, x, xp, args.mesh_gemm)
    total = system.contactnets_loss_and_grad(x, xp)
    if reducer is not None:
        total = reducer.all_reduce_mean()[:1]
    total_loss = total.item()

    if rank == 0:
        names = {'cube': 'contactnets_cube.urdf, 4 friction contacts', 'elbow': 'contactnets_elbow.urdf, 8 friction contacts',
                 'mesh': 'contactnets_cube_mesh.urdf, DeepSupportConvex (ICNN 2x256) geometry incl. 67,328 network weights'}
        collective = describe(reducer) if distributed else 'none'
        line = {
            'metric': 'trajectory-steps/sec (fwd+bwd), batched cube-toss contact sim',
            'value': args.batch * world * args.steps / elapsed,
            'unit': 'trajectory-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype,
            'data': DATA[args.workload],
            'config': {'workload': f'{names[args.workload]}, batch={args.batch} per GPU, fwd+bwd contactnets_loss',
                       'per_gpu_batch': args.batch, 'global_batch': args.batch * world, 'launch': timer.launch,
                       'racing_copies': system.racing_copies(args.batch),
                       'timing': f'median of {len(times)} repeats of the {args.steps}-step timed region '
                                 f'({sum(times):.2f} s measured in total)',
                       'repeats': len(times), 'repeat_ms': [round(t * 1e3, 5) for t in times[:5]],
                       'repeat_ms_min_max': [round(min(times) * 1e3, 5), round(max(times) * 1e3, 5)],
                       'collective': collective, 'mean_loss': total_loss},
            'roofline': roof,
        }
        if distributed:
            line['config']['rccl_ranks'] = rccl_ranks
            line['config']['collective_alt'] = alt
            line['config']['collective_alt_ms'] = alt.get('ms_per_step') if alt else None
            if args.rehearse and world == 1:
                line['config']['rehearsal'] = f'world-size-1 {args.backend} process group: the N > 1 code path on one GPU'
            # informational (the driver forms the scaling curve from the per-N values itself): this run against an N = 1 value
            ref, ref_source = args.ref_value, '--ref-value'
            if ref is None:
                try:
                    path = newest_profile('bench_line.json')
                    with open(path) as handle:
                        cached = json.load(handle)
                    same = (cached.get('n_gpus') == 1 and cached.get('dtype') == args.dtype and args.workload == 'cube'
                            and cached.get('config', {}).get('per_gpu_batch') == args.batch and 'cube' in cached.get('config', {}).get('workload', ''))
                    if same:
                        ref, ref_source = float(cached['value']), os.path.relpath(path, REPO)
                except (OSError, KeyError, ValueError, TypeError):
                    ref = None
            if ref:
                line['config']['weak_scaling_vs_ref'] = {
                    'ref_value_n1': ref, 'ref_source': ref_source, 'efficiency': line['value'] / (world * ref),
                    'efficiency_alt': (alt['value'] / (world * ref)) if alt and alt.get('value') else None}
        if world == 1 and not args.no_configs and args.workload == 'cube' and args.batch == 4096:
            # the other BASELINE.json configurations, same process, after the headline (about a minute in total)
            configs = []
            for w, d, b, k in (('elbow', 'f32', 4096, 1000), ('elbow', 'f64', 4096, 500), ('mesh', 'f32', 4096, 200),
                               ('cube', 'f64', 4096, 1000), ('cube', 'f32', 16384, 400), ('cube', 'f32', 65536, 200), ('cube', 'f64', 65536, 100),
                               ('elbow_mesh', 'f32', 4096, 100), ('clasp', 'f32', 4096, 50), ('slider', 'f32', 4096, 50), ('clasp_mesh', 'f32', 4096, 20),
                               ('two_cubes', 'f32', 4096, 20), ('chain6', 'f32', 4096, 10)):
                try:
                    configs.append(run_loss_config(w, d, b, k, max(10, k // 10), 3, device))
                except Exception as exc:  # noqa: BLE001
                    configs.append({'workload': w, 'dtype': d, 'batch': b, 'error': repr(exc)})
            for w, d, b, h in (('cube', 'f32', 4096, 80), ('cube', 'f32', 65536, 80), ('elbow', 'f32', 4096, 120), ('mesh', 'f32', 4096, 80)):
                try:
                    configs.append(run_simulate_config(w, d, b, h, 5, device))
                except Exception as exc:  # noqa: BLE001
                    configs.append({'workload': f'simulate ({w})', 'dtype': d, 'batch': b, 'error': repr(exc)})
            try:
                configs.append(run_loss_config('mesh', 'f32', 4096, 200, 20, 3, device, mesh_gemm=0))
                configs.append(run_simulate_config('mesh', 'f32', 4096, 80, 5, device, mesh_gemm=0))
                configs.append(run_loss_config('mesh', 'f32', 4096, 200, 20, 3, device, mesh_gemm=2))
                configs.append(run_simulate_config('mesh', 'f32', 4096, 80, 5, device, mesh_gemm=2))
            except Exception as exc:  # noqa: BLE001
                configs.append({'workload': 'mesh (16-bit planes)', 'dtype': 'f32', 'batch': 4096, 'error': repr(exc)})
            for w, k, fused in (('cube', 400, False), ('cube', 400, True), ('slider', 60, True), ('mesh', 60, True)):
                try:
                    configs.append(run_train_config('f32', 4096, k, device, fused, w))
                except Exception as exc:  # noqa: BLE001
                    configs.append({'workload': f'train_step ({w})', 'dtype': 'f32', 'batch': 4096, 'error': repr(exc)})
            # compact (the driver keeps an 8 KB tail of stdout): no launch / note strings (DESIGN.md section 5 has them), the
            # roofline of a configuration as [bound, achieved, fraction of peak] -- the peaks are the headline's (8 TB/s HBM,
            # 157.3 TFLOP/s f32 MFMA) and the kernel behind every figure is named in DESIGN.md section 5's table
            def short(c):
                out = {k: v for k, v in c.items() if k not in ('launch', 'unit', 'steps', 'roofline', 'optimizer_steps_per_s')}
                if 'roofline' in c:
                    out['roofline'] = [c['roofline']['bound'], c['roofline']['achieved'], c['roofline']['frac']]
                return out
            line['configs'] = [short(c) for c in configs]
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(x_np, xp_np, dt, args.workload)
        print(json.dumps(compact(line), separators=(',', ':')), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
