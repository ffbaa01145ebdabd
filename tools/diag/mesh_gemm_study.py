"""Diagnostic / study: the three forms of the float ICNN GEMM kernels -- f32 MFMA (mesh_gemm = 0), bf16 matrix
cores on 3 bf16 planes (3: six products per k-step) and on 2 planes (2: "bf16 x 3") -- on the 4096-pair cube batch with the
reference-initialised network of tests/golden/cube_mesh_literal.npz: per-kernel times, step time, and against the float64 kernels
on the same inputs: fraction of support points that differ (a LeakyReLU mask flipped), loss and gradient differences.
Run on the MI355X: python tools/diag/mesh_gemm_study.py [out.json]"""
import json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
big = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_mesh_literal.npz'))
GEOM = 'multibody_terms.contact_terms.geometries.'


def build(dtype, mode=0):
    s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'cube_mesh.urdf')}, float(big['dt']), dtype=dtype, device='cuda:0')
    s.load_state_dict({name: torch.tensor(g['param/' + name]) for name, _ in s.named_parameters()})
    s.multibody_terms.contact_terms.geometries[1].perturbations = torch.tensor(g[f'param/{GEOM}1.perturbations'], dtype=dtype, device='cuda:0')
    if dtype == torch.float32:
        s.set_solver(mesh_gemm=mode)
    return s


x64 = torch.tensor(big['x'], device='cuda:0'); xp64 = torch.tensor(big['x_plus'], device='cuda:0')
x32, xp32 = x64.float(), xp64.float()
ref = build(torch.float64)
p_ref = ref.support_points(xp64).cpu().numpy()
l_ref = ref.contact_forces(x64, xp64)[0].cpu().numpy()
t_ref = ref.contactnets_loss_and_grad(x64, xp64).item()
g_ref = {n: p.grad.cpu().numpy().copy() for n, p in ref.named_parameters()}
out = {}
for mode in (0, 3, 4, 2):
    s = build(torch.float32, mode)
    pts = s.support_points(xp32).cpu().double().numpy()
    differ = np.abs(pts - p_ref).max(-1) > 1e-5          # a support point that is another vertex of the learned shape
    loss = s.contact_forces(x32, xp32)[0].cpu().double().numpy()
    total = s.contactnets_loss_and_grad(x32, xp32).item()
    grads = {n: p.grad.cpu().double().numpy().copy() for n, p in s.named_parameters()}
    ms = None
    for _ in range(3):
        cur = s.profile_mesh_kernels(x32, xp32, reps=30)
        ms = cur if ms is None else {k: min(ms[k], v) for k, v in cur.items()}
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): s.contactnets_loss_and_grad(x32, xp32)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(10): s.contactnets_loss_and_grad(x32, xp32)
    for _ in range(3): graph.replay()
    torch.cuda.synchronize()
    step = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10): graph.replay()
        torch.cuda.synchronize()
        step = min(step, (time.perf_counter() - t0) / 100 * 1e6)
    worst = max(np.abs(grads[n] - g_ref[n]).max() / max(np.abs(g_ref[n]).max(), 1e-30) for n in g_ref)
    out[str(mode)] = {'kernels_us': {k: round(v * 1e3, 2) for k, v in ms.items()}, 'step_us': round(step, 2),
                      'support_points_differing_frac': float(differ.mean()), 'support_point_err_same_vertex_max': float(np.abs(pts - p_ref).max(-1)[~differ].max()),
                      'loss_err_max': float(np.abs(loss - l_ref).max()), 'loss_err_q995': float(np.quantile(np.abs(loss - l_ref), 0.995)),
                      'mean_loss_err': abs(total - t_ref), 'grad_rel_err_worst': float(worst)}
    print(mode, json.dumps(out[str(mode)]), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
