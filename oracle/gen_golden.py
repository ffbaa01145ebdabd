"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON in the authoring container.

TEST INFRASTRUCTURE ONLY -- needs /root/reference, never runs on the GPU box.

Recipe: the eight third-party packages the reference imports
but this image lacks are replaced by MagicMock modules; the Drake-dependent constructors are
bypassed with ``Cls.__new__``; the three pieces of arithmetic that live in those packages
(drake_pytorch closures for M, F and geometry kinematics; sappy's QP solve) are supplied by
``oracle/dpll_oracle.py``.  Everything else executed here -- contactnets_loss,
forward_dynamics, Integrator.simulate, MultibodyTerms / LagrangianTerms / ContactTerms
.forward, GeometryCollider, Box, DeepSupportConvex / HomogeneousICNN,
InertialParameterConverter, state_space, quaternion, TrajectorySliceDataset and
DrakeMultibodyLearnableExperiment.contactnets_loss -- is the reference's unmodified code,
and its outputs are what the fixtures record.

Body-body contact (GeometryCollider.collide_mesh_mesh, geometry.py:585-643) asks `fcl` for one direction per pair; here
`fcl` is `DirectionSearchFcl` below -- the few entry points collide_mesh_mesh calls, answered by the oracle's exact
direction search (oracle.pair_direction_exact) -- so that the reference's own collide_mesh_mesh / ContactTerms.forward
arithmetic (contact frame, witness points, signed distance, Jacobian difference, friction combination, contact order)
runs on the pair fixtures.  The direction itself stays the oracle's: fcl's role is parity-unpinned.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = '/root/reference'
ASSETS = os.path.join(REFERENCE, 'assets')
GOLDEN = os.path.join(REPO, 'tests', 'golden')
DT = 0.0068  # reference examples/contactnets_simple.py:52



class DirectionSearchFcl:
    """What reference geometry.py:603-625 uses of fcl, for shapes given as (vertices (N, 3), margin): `collide`
    reports one contact whose normal is the direction of minimum penetration, `distance` nearest points whose difference
    is the nearest-points direction."""

    class Transform:
        def __init__(self, rotation=None, translation=None):
            self.R = np.eye(3) if rotation is None else np.asarray(rotation, dtype=np.float64)
            self.p = np.zeros(3) if translation is None else np.asarray(translation, dtype=np.float64)

    class CollisionObject:
        def __init__(self, shape, transform):
            self.shape, self.transform = shape, transform

        def setTransform(self, transform):  # noqa: N802 (fcl's name)
            self.transform = transform

        def world(self):
            vertices, margin = (self.shape.vertices, 0.0) if isinstance(self.shape, DirectionSearchFcl.BVHModel) else self.shape
            return vertices @ self.transform.R.T + self.transform.p, margin

    class BVHModel:
        """DeepSupportConvex.get_fcl_geometry (geometry.py:343-358) hands its extracted mesh over through this interface"""

        def beginModel(self, n_vertices, n_faces):  # noqa: N802 (fcl's name)
            self.vertices = np.zeros((0, 3))

        def addSubModel(self, vertices, faces):  # noqa: N802
            self.vertices = np.asarray(vertices, dtype=np.float64)

        def endModel(self):  # noqa: N802
            pass

    class CollisionRequest:
        enable_contact = False

    class DistanceRequest:
        enable_nearest_points = False

    class CollisionResult:
        def __init__(self):
            self.contacts = []

    class DistanceResult:
        def __init__(self):
            self.nearest_points = [np.zeros(3), np.zeros(3)]

    class _Contact:
        def __init__(self, normal):
            self.normal = normal

    @staticmethod
    def _search(a_obj, b_obj):
        (va, ma), (vb, mb) = a_obj.world(), b_obj.world()
        direction = _oracle().pair_direction_exact(va, vb)
        separation = (vb @ direction).min() - (va @ direction).max() - ma - mb
        return direction, separation

    @staticmethod
    def collide(a_obj, b_obj, request, result):
        direction, separation = DirectionSearchFcl._search(a_obj, b_obj)
        if separation < 0:
            result.contacts = [DirectionSearchFcl._Contact(torch.tensor(direction))]
            return 1
        return 0

    @staticmethod
    def distance(a_obj, b_obj, request, result):
        direction, separation = DirectionSearchFcl._search(a_obj, b_obj)
        result.nearest_points = [np.zeros(3), separation * direction]
        return separation


def _oracle():
    from oracle import dpll_oracle
    return dpll_oracle


sys.modules['fcl'] = DirectionSearchFcl
for _name in ['fcl', 'pywavefront', 'sappy', 'drake_pytorch', 'optuna', 'optuna.trial', 'optuna.logging', 'wandb',
              'pydrake', 'pydrake.geometry', 'pydrake.multibody', 'pydrake.multibody.plant',
              'pydrake.multibody.tree', 'pydrake.multibody.parsing', 'pydrake.symbolic', 'pydrake.systems',
              'pydrake.systems.framework', 'pydrake.systems.analysis', 'pydrake.autodiffutils', 'pydrake.math',
              'pydrake.visualization', 'pydrake.all', 'pydrake.common', 'pydrake.systems.primitives',
              'pydrake.geometry.render', 'pydrake.systems.sensors']:
    sys.modules.setdefault(_name, MagicMock())
sys.path.insert(0, REFERENCE)
sys.path.insert(0, REPO)

from torch import nn  # noqa: E402
from torch.nn import ModuleList, Parameter  # noqa: E402

from dair_pll import state_space  # noqa: E402
from dair_pll.data_config import TrajectorySliceConfig  # noqa: E402
from dair_pll.dataset_management import TrajectorySliceDataset  # noqa: E402
from dair_pll.drake_experiment import DrakeMultibodyLearnableExperiment  # noqa: E402
from dair_pll.geometry import Box, DeepSupportConvex, Plane, Polygon, Sphere  # noqa: E402
from dair_pll.inertia import InertialParameterConverter  # noqa: E402
from dair_pll.integrator import VelocityIntegrator  # noqa: E402
from dair_pll.multibody_learnable_system import MultibodyLearnableSystem  # noqa: E402
from dair_pll.multibody_terms import ContactTerms, LagrangianTerms, MultibodyTerms  # noqa: E402
from dair_pll.system import System  # noqa: E402

from oracle import dpll_oracle as O  # noqa: E402

assert torch.get_default_dtype() == torch.float64  # reference inertia.py:96
# deep_support_function.py:12-16 builds its surface directions at import, in whatever default dtype the import order left
# (here float32: geometry was imported before inertia); extract_mesh then multiplies them with float64 weights.  As the
# module would have built them under the float64 default:
import dair_pll.deep_support_function as _dsf  # noqa: E402
_line = torch.linspace(-1, 1, steps=8)
_grid = torch.cartesian_prod(_line, _line, _line)
_dsf._SURFACE = _grid[_grid.abs().max(dim=-1).values >= 1.0]
_dsf._SURFACE = _dsf._SURFACE / _dsf._SURFACE.norm(dim=-1, keepdim=True)


class RecordingSolver:
    """Stands in for sappy.SAPSolver(); records exactly what the reference passes."""

    def __init__(self):
        self.calls = []

    def apply(self, J, q, eps):
        # differentiable wherever the reference keeps the graph (forward_dynamics; the loss detaches the result itself)
        diff = torch.is_grad_enabled() and (J.requires_grad or q.requires_grad)
        f = O.sap_solve_diff(J, q, eps) if diff else O.sap_solve(J, q, eps)
        self.calls.append((J.detach().clone(), q.detach().clone(), eps, f.detach().clone()))
        return f


class VertexSupport(nn.Module):
    """`network` of a pair member: one support point per direction, as DeepSupportConvex.network returns it
    (geometry.py:627-629) -- the vertex furthest along the direction (+ a sphere's radius along it), ties by the
    oracle's rule (oracle.PAIR_TIE)."""

    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, 'owner', owner)  # not a submodule: the owner holds this module

    def forward(self, directions):
        vertices, margin = self.owner.vertex_set()
        dots = (directions @ vertices.t()).detach()
        best = torch.zeros(dots.shape[:-1], dtype=torch.long)
        value = dots[..., 0].clone()
        for u in range(1, dots.shape[-1]):
            better = dots[..., u] > value + O.PAIR_TIE
            best = torch.where(better, torch.full_like(best, u), best)
            value = torch.where(better, dots[..., u], value)
        return vertices[best] + margin * directions


def pair_member(base, geometry):
    """`geometry` (a reference Box / Sphere / Polygon) re-typed so that GeometryCollider.collide dispatches a pair of them
    to collide_mesh_mesh (geometry.py:543-546 tests for DeepSupportConvex): a class of the same name -- the reference's
    type order reads the class name (:66-69) -- that also derives from DeepSupportConvex, with `network` the shape's own
    support function and `get_fcl_geometry` its vertex set for DirectionSearchFcl.  Ground contacts still go through the
    base class's support_points."""
    def vertex_set(self):
        if base is Box:
            return self.get_vertices(torch.zeros(3)), torch.zeros(())
        if base is Sphere:
            return torch.zeros((1, 3)), torch.abs(self.length_param)
        return self.vertices, torch.zeros(())

    def get_fcl_geometry(self):
        vertices, margin = self.vertex_set()
        return vertices.detach().numpy().astype(np.float64), float(margin)
    cls = type(base.__name__, (base, DeepSupportConvex), {'vertex_set': vertex_set, 'get_fcl_geometry': get_fcl_geometry,
                                                          'train': nn.Module.train})
    geometry.__class__ = cls
    geometry.network = VertexSupport(geometry)
    return geometry


def build_reference_system(urdf, inertia_mode: str, mesh_seed: int = 0, mesh_representation: str = 'deep_support'):
    """The reference's MultibodyLearnableSystem with Drake-dependent construction bypassed.  ``urdf``: one path, or
    ``{name: path}`` -- the ``init_urdfs`` of a system of several models (multibody_learnable_system.py:51-54)."""
    spec = O.system_spec(urdf, mesh_representation)
    geoms = O.geometry_table(spec)
    n_joints = spec['n_joints']

    lt = LagrangianTerms.__new__(LagrangianTerms)
    nn.Module.__init__(lt)
    lt.mass_matrix = lambda q, inertia: O.mass_matrix(spec, q, inertia, inertia_mode)
    lt.lagrangian_forces = lambda q, v, u, inertia: O.lagrangian_forces(spec, q, v, inertia, inertia_mode, u)  # (+ B u, :142-146)
    # one row per Drake body (multibody_terms.py:161-207): a link welded on by a `fixed` joint has a row of its own
    pi_cm = torch.tensor([[b['mass']] + [b['mass'] * c for c in b['com']] + b['inertia_cm']
                          for b in O.inertia_rows(spec)])
    lt.inertial_parameters = Parameter(InertialParameterConverter.pi_cm_to_theta(pi_cm), requires_grad=True)

    ct = ContactTerms.__new__(ContactTerms)
    nn.Module.__init__(ct)
    ct.geometry_rotations = lambda q: O.geometry_kinematics(spec, q)[0]
    ct.geometry_translations = lambda q: O.geometry_kinematics(spec, q)[1]
    ct.geometry_spatial_jacobians = lambda q: O.geometry_kinematics(spec, q)[2]
    modules = []
    for geom in geoms:
        if geom['kind'] == 'plane':
            modules.append(Plane())
        elif geom['kind'] == 'box':
            modules.append(Box(torch.tensor(geom['half']), 4))
        elif geom['kind'] == 'sphere':
            # Sphere.__init__ cannot run (geometry.py:431 asserts `radius.numel == 1`, a bound method compared with 1):
            # built like the Drake-dependent classes, by __new__ + the attribute its methods use
            sphere = Sphere.__new__(Sphere)
            nn.Module.__init__(sphere)
            sphere.length_param = Parameter(torch.tensor(geom['radius']), requires_grad=True)
            modules.append(sphere)
        elif geom['kind'] == 'polygon':
            modules.append(Polygon(torch.tensor(geom['vertices'])))  # geometry.py:220-252, n_query = 4
        else:
            # one seed per network: mesh_seed for the first (the cube), mesh_seed + 1 for the elbow's second link, ...
            torch.manual_seed(mesh_seed + sum(isinstance(m, DeepSupportConvex) for m in modules))
            modules.append(DeepSupportConvex(torch.tensor(geom['vertices'])))
    for a_index, b_index in spec['pairs']:  # body-body candidates: both members take the collide_mesh_mesh path
        for index in (a_index, b_index):
            if not isinstance(modules[index], DeepSupportConvex):
                modules[index] = pair_member(type(modules[index]), modules[index])
    ct.geometries = ModuleList(modules)
    ct.friction_params = Parameter(torch.tensor([g['mu'] for g in geoms]), requires_grad=True)
    n_g = len(geoms)
    # ground pairs in geometry order, then the body-body candidates (quirk Q8: Drake's order is not specified)
    # (Drake's GetCollisionCandidates has no anchored-anchored pair: the geometries of a body welded to the world do not meet
    # the ground, which sits on the world body -- O.ground_geometries)
    ground = O.ground_geometries(spec)
    ct.collision_candidates = torch.tensor([[0] * len(ground) + [a for a, _ in spec['pairs']],
                                            ground + [b for _, b in spec['pairs']]]).long()

    mt = MultibodyTerms.__new__(MultibodyTerms)
    nn.Module.__init__(mt)
    mt.lagrangian_terms = lt
    mt.contact_terms = ct

    system = MultibodyLearnableSystem.__new__(MultibodyLearnableSystem)
    # the plant's state space as generate_state_space builds it (drake_utils.py:309-335): the world model (no coordinates),
    # then one FloatingBaseSpace -- or, for a model welded to the world, FixedBaseSpace -- per model
    models = spec.get('models', [spec])
    space = state_space.ProductSpace([state_space.FixedBaseSpace(0)] + [
        (state_space.FixedBaseSpace if model['fixed_base'] else state_space.FloatingBaseSpace)(model['n_joints']) for model in models])
    System.__init__(system, space, VelocityIntegrator(space, system.sim_step, DT))
    system.multibody_terms = mt
    system.solver = RecordingSolver()
    system.dt = DT
    system.set_carry_sampler(lambda: torch.tensor([False]))
    system.max_batch_dim = 1
    return system, spec


def named_grads(system) -> dict:
    return {name: param.grad.detach().clone().numpy() for name, param in system.named_parameters()}


def named_values(system) -> dict:
    out = {name: param.detach().clone().numpy() for name, param in system.named_parameters()}
    for index, geometry in enumerate(system.multibody_terms.contact_terms.geometries):
        if isinstance(geometry, DeepSupportConvex) and hasattr(geometry, 'perturbations'):  # (not a pair_member)
            out[f'multibody_terms.contact_terms.geometries.{index}.perturbations'] = \
                geometry.perturbations.detach().clone().numpy()
    return out


def record_case(name: str, urdf: str, x: torch.Tensor, x_plus: torch.Tensor, inertia_mode: str,
                sim_steps: int = 4, sim_rows=None, mesh_representation: str = 'deep_support', prepare=None, u=None) -> None:
    """``u``: actuation inputs ``(B, n_u)`` of an actuated model, handed to the reference's ``multibody_terms``,
    ``contactnets_loss`` and ``forward_dynamics`` (its ``sim_step`` -- hence ``step`` / ``simulate`` -- always passes a ``u`` of
    width 0, multibody_learnable_system.py:311); recorded as ``u``."""
    system, _ = build_reference_system(urdf, inertia_mode, mesh_representation=mesh_representation)
    if prepare is not None:  # e.g. move the parameters off their URDF values before anything is recorded
        prepare(system)
    out = {'urdf': os.path.basename(urdf) if isinstance(urdf, str) else ' '.join(f'{k}={os.path.basename(v)}' for k, v in urdf.items()),
           'dt': DT, 'inertia_mode': inertia_mode, 'x': x.numpy(), 'x_plus': x_plus.numpy()}
    for key, value in named_values(system).items():
        out['param/' + key] = value

    # --- terms at the next state (what the loss evaluates, Q6) -------------------------
    q_plus, v_plus = system.space.q_v(x_plus)
    if u is None:
        u = torch.zeros(x.shape[:-1] + (0,))
    else:
        out['u'] = u.numpy()
    with torch.no_grad():
        D, M, J, phi, a = system.multibody_terms(q_plus, v_plus, u)
    out.update({'terms/D': D.numpy(), 'terms/M': M.numpy(), 'terms/J': J.numpy(), 'terms/phi': phi.numpy(),
                'terms/a': a.numpy()})

    # --- loss through the caller's wrapper (A17) and gradients --------------------------
    system.solver.calls.clear()
    loss_batch = system.contactnets_loss(x, u, x_plus)
    J_M, q_s, eps, f_s = system.solver.calls[-1]
    out.update({'loss': loss_batch.detach().numpy(), 'solver_loss/J': J_M.numpy(), 'solver_loss/q': q_s.numpy(),
                'solver_loss/eps': eps, 'solver_loss/f': f_s.numpy()})
    kkt = O.kkt_residuals(J_M, q_s, eps, f_s)
    out['solver_loss/kkt'] = np.array([kkt[k].max().item() for k in ('primal', 'dual', 'complementarity')])
    system.zero_grad()
    if u.shape[-1] == 0:
        mean_loss = DrakeMultibodyLearnableExperiment.contactnets_loss(None, x.unsqueeze(-2), x_plus.unsqueeze(-2),
                                                                     system)
    else:  # (the experiment's wrapper builds a u of width 0 itself, drake_experiment.py:219: the system's method with the inputs)
        mean_loss = system.contactnets_loss(x, u, x_plus).mean()
    mean_loss.backward()
    out['loss_mean'] = mean_loss.detach().numpy()
    for key, value in named_grads(system).items():
        out['grad/' + key] = value

    # --- forward dynamics / step / simulate --------------------------------------------
    system.solver.calls.clear()
    q, v = system.space.q_v(x)
    with torch.no_grad():
        v_next = system.forward_dynamics(q, v, u)
        J_M, q_s, eps, f_s = system.solver.calls[-1]
        x_next, _ = system.integrator.step(x, torch.zeros(x.shape[:-1] + (1,)))
        rows = list(range(min(8, x.shape[0]))) if sim_rows is None else sim_rows
        x_0 = x[rows].unsqueeze(-2)
        traj, _ = system.simulate(x_0, torch.zeros((len(rows), 1)), sim_steps)
    kkt = O.kkt_residuals(J_M, q_s, eps, f_s)
    out.update({'dynamics/v_next': v_next.numpy(), 'dynamics/x_next': x_next.numpy(),
                'solver_dynamics/J': J_M.numpy(), 'solver_dynamics/q': q_s.numpy(), 'solver_dynamics/eps': eps,
                'solver_dynamics/f': f_s.numpy(),
                'solver_dynamics/kkt': np.array([kkt[k].max().item()
                                                 for k in ('primal', 'dual', 'complementarity')]),
                'simulate/rows': np.array(rows), 'simulate/steps': sim_steps, 'simulate/traj': traj.numpy()})
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    print(f'{name}: B={x.shape[0]} loss mean {float(mean_loss):.6e} kkt(loss) {out["solver_loss/kkt"]} '
          f'kkt(dyn) {out["solver_dynamics/kkt"]}')


def cube_pairs(files, stride: int = 1):
    """(x, x_plus) pairs exactly as TrajectorySliceDataset builds them
    (reference dataset_management.py:43-59 with the default TrajectorySliceConfig)."""
    dataset = TrajectorySliceDataset(TrajectorySliceConfig())
    for index in files:
        dataset.add_slices_from_trajectory(torch.load(os.path.join(ASSETS, 'contactnets_cube', f'{index}.pt')))
    past = torch.stack(dataset.previous_states_slices)[::stride]
    future = torch.stack(dataset.future_states_slices)[::stride]
    return past[..., -1, :].clone(), future[..., 0, :].clone()


def elbow_pairs(n_traj: int = 6, steps: int = 120, keep_every: int = 5, seed: int = 0):
    """Synthetic elbow tosses: initial states drawn like UniformSampler(space, ELBOW_SAMPLER_RANGE,
    ELBOW_X_0) (reference state_space.py:900-948, examples/contactnets_simple.py:60-67) and
    rolled out with the reference's own simulate on the stub-built system."""
    system, _ = build_reference_system(os.path.join(ASSETS, 'contactnets_elbow.urdf'), 'reference_literal')
    x_0 = torch.tensor([1., 0., 0., 0., 0., 0., 0.21 + .015, np.pi, 0., 0., 0., 0., 0., -.075, 0.])
    ranges = torch.tensor([2 * np.pi, 2 * np.pi, 2 * np.pi, .03, .03, .015, np.pi, 6., 6., 6., .5, .5, .075, 6.])
    gen = torch.Generator().manual_seed(seed)
    space = system.space
    xs, xps = [], []
    with torch.no_grad():
        for _ in range(n_traj):
            delta = (2 * torch.rand(ranges.shape, generator=gen) - 1) * ranges
            start = space.shift_state(x_0.unsqueeze(0), delta.unsqueeze(0))
            traj, _ = system.simulate(start.unsqueeze(-2), torch.zeros((1, 1)), steps)
            traj = traj[0]
            xs.append(traj[:-1][::keep_every])
            xps.append(traj[1:][::keep_every])
    return torch.cat(xs).clone(), torch.cat(xps).clone()


def record_bench_batch(name: str, n_pairs: int = 4096, seed: int = 0) -> None:
    """BASELINE configs[1] inputs: ``n_pairs`` of the 57,812 real cube (x, x+) pairs,
    torch.Generator().manual_seed(seed) randperm; expected loss / gradients from the reference run."""
    cube = os.path.join(ASSETS, 'contactnets_cube.urdf')
    x_all, xp_all = cube_pairs(range(550))
    pick = torch.randperm(x_all.shape[0], generator=torch.Generator().manual_seed(seed))[:n_pairs]
    x, x_plus = x_all[pick].clone(), xp_all[pick].clone()
    system, _ = build_reference_system(cube, 'reference_literal')
    u = torch.zeros(x.shape[:-1] + (0,))
    loss = system.contactnets_loss(x, u, x_plus)
    system.zero_grad()
    loss.mean().backward()
    out = {'urdf': os.path.basename(cube), 'dt': DT, 'inertia_mode': 'reference_literal', 'n_total_pairs': x_all.shape[0],
           'x': x.numpy(), 'x_plus': x_plus.numpy(), 'loss': loss.detach().numpy(),
           'loss_mean': loss.mean().detach().numpy()}
    for key, value in named_values(system).items():
        out['param/' + key] = value
    for key, value in named_grads(system).items():
        out['grad/' + key] = value
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    print(f'{name}: {n_pairs} of {x_all.shape[0]} pairs, loss mean {loss.mean().item():.6e}')


def kept_subset_gradients(system, x: torch.Tensor, x_plus: torch.Tensor, chunk: int = 0) -> dict:
    """What the float32 checks at the BASELINE sizes compare with: the reference's own signed distances at x+ (their smallest
    magnitude per item), the items kept -- those off the |phi| kink of the loss, where the sign of phi and with it the
    gradient flips with float32 rounding of the inputs --, and the reference-run mean loss and gradients over the kept items."""
    with torch.no_grad():
        phi_min = torch.cat([system.multibody_terms(*system.space.q_v(xp), torch.zeros(xp.shape[:-1] + (0,)))[3].abs().min(-1).values
                             for xp in x_plus.split(chunk or x_plus.shape[0])])
    keep = phi_min >= 1e-6
    xk, xpk = x[keep], x_plus[keep]
    system.zero_grad()
    total = 0.0
    for xc, xpc in zip(xk.split(chunk or xk.shape[0]), xpk.split(chunk or xk.shape[0])):
        part = system.contactnets_loss(xc, torch.zeros(xc.shape[:-1] + (0,)), xpc).sum() / xk.shape[0]
        part.backward()
        total += part.item()
    out = {'terms/phi_min': phi_min.numpy(), 'keep': keep.numpy(), 'loss_mean_keep': np.array(total)}
    for key, value in named_grads(system).items():
        out['grad_keep/' + key] = value
    return out


def record_mesh_bench_batch(name: str = 'cube_mesh_4096', pairs_from: str = 'cube_box_4096', chunk: int = 256) -> None:
    """BASELINE configs[3] at its stated size: the reference's own DeepSupportConvex / HomogeneousICNN loss
    (geometry.py:255-325, deep_support_function.py:213-266) on the 4096 benchmark pairs of `pairs_from`, in chunks of 256 so that
    autograd's (B, 256, 256) weight gradient stays small; per-item loss, batch mean and every gradient incl. the network's.
    The pairs themselves are not stored again (the tests read them from `pairs_from`)."""
    pairs = np.load(os.path.join(GOLDEN, pairs_from + '.npz'))
    x, x_plus = torch.tensor(pairs['x']), torch.tensor(pairs['x_plus'])
    mesh = os.path.join(ASSETS, 'contactnets_cube_mesh.urdf')
    system, _ = build_reference_system(mesh, 'reference_literal')
    out = {'urdf': os.path.basename(mesh), 'dt': DT, 'inertia_mode': 'reference_literal', 'pairs_from': pairs_from}
    for key, value in named_values(system).items():
        out['param/' + key] = value
    system.zero_grad()
    losses = []
    for xc, xpc in zip(x.split(chunk), x_plus.split(chunk)):
        loss = system.contactnets_loss(xc, torch.zeros(xc.shape[:-1] + (0,)), xpc)
        (loss.sum() / x.shape[0]).backward()
        losses.append(loss.detach())
    loss = torch.cat(losses)
    out.update({'loss': loss.numpy(), 'loss_mean': loss.mean().numpy()})
    for key, value in named_grads(system).items():
        out['grad/' + key] = value
    out.update(kept_subset_gradients(system, x, x_plus, chunk))
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    print(f'{name}: {x.shape[0]} pairs of {pairs_from}, loss mean {loss.mean().item():.6e}, kept {int(out["keep"].sum())}')


def elbow_rollouts(n_traj: int = 40, steps: int = 120, seed: int = 0) -> torch.Tensor:
    """BASELINE configs[2] inputs exactly as SURVEY 8d specifies them: ``n_traj`` initial states from the
    reference's own ``UniformSampler(space, ELBOW_SAMPLER_RANGE, x_0=ELBOW_X_0)`` (reference
    state_space.py:900-948, examples/contactnets_simple.py:60-67; global generator seeded with ``seed``),
    each rolled out ``steps`` (= TRAJECTORY_LENGTHS['elbow'], :70) steps with the reference's own
    ``System.simulate`` on the stub-built system.  Returns ``(n_traj, steps + 1, 15)``."""
    system, _ = build_reference_system(os.path.join(ASSETS, 'contactnets_elbow.urdf'), 'reference_literal')
    x_0 = torch.tensor([1., 0., 0., 0., 0., 0., 0.21 + .015, np.pi, 0., 0., 0., 0., 0., -.075, 0.])
    ranges = torch.tensor([2 * np.pi, 2 * np.pi, 2 * np.pi, .03, .03, .015, np.pi, 6., 6., 6., .5, .5, .075, 6.])
    sampler = state_space.UniformSampler(system.space, ranges, x_0=x_0)
    torch.manual_seed(seed)
    starts = torch.stack([sampler.get_sample() for _ in range(n_traj)])
    with torch.no_grad():
        traj, _ = system.simulate(starts.unsqueeze(-2), torch.zeros((n_traj, 1)), steps)
    return traj


def record_elbow_bench_batch(name: str = 'elbow_box_4096', n_pairs: int = 4096, seed: int = 0) -> None:
    """BASELINE configs[2]: 4096 seeded (x, x+) pairs of 120-step elbow tosses (sliced by the reference's
    TrajectorySliceDataset), expected per-item loss, batch mean and every parameter gradient from the reference run."""
    elbow = os.path.join(ASSETS, 'contactnets_elbow.urdf')
    traj = elbow_rollouts(seed=seed)
    dataset = TrajectorySliceDataset(TrajectorySliceConfig())
    for t in traj:
        dataset.add_slices_from_trajectory(t)
    x_all = torch.stack(dataset.previous_states_slices)[..., -1, :]
    xp_all = torch.stack(dataset.future_states_slices)[..., 0, :]
    pick = torch.randperm(x_all.shape[0], generator=torch.Generator().manual_seed(seed))[:n_pairs]
    x, x_plus = x_all[pick].clone(), xp_all[pick].clone()
    system, _ = build_reference_system(elbow, 'reference_literal')
    system.zero_grad()
    mean_loss = DrakeMultibodyLearnableExperiment.contactnets_loss(None, x.unsqueeze(-2), x_plus.unsqueeze(-2), system)
    mean_loss.backward()
    with torch.no_grad():
        loss = system.contactnets_loss(x, torch.zeros(x.shape[:-1] + (0,)), x_plus)
        rows = list(range(0, n_pairs, n_pairs // 256))
        v_next = system.forward_dynamics(*system.space.q_v(x[rows]), torch.zeros((len(rows), 0)))
    out = {'urdf': os.path.basename(elbow), 'dt': DT, 'inertia_mode': 'reference_literal', 'n_total_pairs': x_all.shape[0],
           'x': x.numpy(), 'x_plus': x_plus.numpy(), 'loss': loss.numpy(), 'loss_mean': mean_loss.detach().numpy(),
           'dynamics/rows': np.array(rows), 'dynamics/v_next': v_next.numpy()}
    for key, value in named_values(system).items():
        out['param/' + key] = value
    for key, value in named_grads(system).items():
        out['grad/' + key] = value
    out.update(kept_subset_gradients(system, x, x_plus))
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    print(f'{name}: {n_pairs} of {x_all.shape[0]} pairs, loss mean {float(mean_loss):.6e}, max {loss.max().item():.3e}')


def record_slice_fixture(name: str = 'cube_trajectories_0_2') -> None:
    """The reference's own data files assets/contactnets_cube/{0,1,2}.pt as raw arrays (data, not code) together with
    the (x, x+) pairs the reference's TrajectorySliceDataset makes of them (dataset_management.py:43-59): pins the
    trainer's slice rule on real trajectories."""
    out = {}
    for index in (0, 1, 2):
        out[f'trajectory/{index}'] = torch.load(os.path.join(ASSETS, 'contactnets_cube', f'{index}.pt')).numpy()
    x, xp = cube_pairs([0, 1, 2])
    out['x'], out['x_plus'] = x.numpy(), xp.numpy()
    dataset = TrajectorySliceDataset(TrajectorySliceConfig(t_prediction=3))
    for index in (0, 1, 2):
        dataset.add_slices_from_trajectory(torch.load(os.path.join(ASSETS, 'contactnets_cube', f'{index}.pt')))
    out['window3/x_past'] = torch.stack(dataset.previous_states_slices).numpy()
    out['window3/x_future'] = torch.stack(dataset.future_states_slices).numpy()
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    print(f'{name}: {x.shape[0]} pairs, {out["window3/x_past"].shape[0]} windows of 3')


def record_cube_tosses(name: str = 'contactnets_cube_tosses') -> None:
    """The reference's whole cube-toss data set (assets/contactnets_cube/{0..549}.pt: 550 real tosses, (T, 13) float64
    states, T in [85, 139]; SURVEY section 2 row 22 -- data, not code) as ONE array file under assets/: the states of all
    trajectories one after the other plus their lengths.  `trainer.load_tosses` cuts it back into trajectories; the f1 demo
    (examples/contactnets_cube.py) and bench.py's training configuration run on its 57,812 (x, x+) pairs."""
    states, lengths = [], []
    for index in range(550):
        trajectory = torch.load(os.path.join(ASSETS, 'contactnets_cube', f'{index}.pt')).numpy()
        assert trajectory.ndim == 2 and trajectory.shape[1] == 13 and trajectory.dtype == np.float64
        states.append(trajectory)
        lengths.append(trajectory.shape[0])
    np.savez_compressed(os.path.join(REPO, 'assets', name + '.npz'), states=np.concatenate(states), lengths=np.array(lengths, dtype=np.int32),
                        dt=np.float64(DT))
    print(f'{name}: {len(lengths)} trajectories, {sum(lengths)} states, {sum(lengths) - len(lengths)} pairs')


def record_dynamics_gradients(name: str = 'dynamics_gradients') -> None:
    """gradients THROUGH the reference's own ``forward_dynamics`` / ``VelocityIntegrator.step`` /
    ``Integrator.simulate`` (multibody_learnable_system.py:293-304, experiment.py:292-320) by torch autograd, with
    respect to every learnable parameter and to the input state, for seeded linear functionals of the next state /
    of a 3-step rollout.  Only the cone solve's backward is not the reference's (sappy is absent): it is the
    implicit-function derivative of the unique optimum, ``oracle.sap_solve_diff``."""
    out = {'dt': DT}
    cases = (('cube', os.path.join(ASSETS, 'contactnets_cube.urdf'), 'cube_box_literal', 96),
             ('elbow', os.path.join(ASSETS, 'contactnets_elbow.urdf'), 'elbow_box_literal', 48))
    for key, urdf, source, count in cases:
        g = np.load(os.path.join(GOLDEN, source + '.npz'))
        rows = np.linspace(0, g['x'].shape[0] - 1, count).astype(int)
        gen = torch.Generator().manual_seed(7)
        for label, steps in (('step', 1), ('rollout3', 3)):
            system, _ = build_reference_system(urdf, 'reference_literal')
            x = torch.tensor(g['x'][rows]).clone().requires_grad_(True)
            carry = torch.zeros((count, 1))
            traj, _ = system.simulate(x.unsqueeze(-2), carry, steps)
            w = torch.rand(traj[:, 1:].shape, generator=gen) - 0.5
            total = (traj[:, 1:] * w).sum()
            system.zero_grad()
            total.backward()
            prefix = f'{key}/{label}/'
            out[prefix + 'x'] = x.detach().numpy()
            out[prefix + 'w'] = w.numpy()
            out[prefix + 'traj'] = traj.detach().numpy()
            out[prefix + 'total'] = total.detach().numpy()
            out[prefix + 'grad_x'] = x.grad.numpy()
            for pname, value in named_grads(system).items():
                out[prefix + 'grad/' + pname] = value
            print(f'{name}: {key} {label}: total {float(total):.6e}, |grad_x| max {x.grad.abs().max().item():.3e}')
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)


def general_tosses(urdf: str, n_traj: int, steps: int, keep_every: int, seed: int, mesh_representation: str = 'deep_support',
                   prepare=None):
    """(x, x_plus) pairs of seeded tosses (random attitude, 5-12 cm above the ground, spinning) rolled out by the
    reference's simulate on the stub-built system, every `keep_every`-th pair kept."""
    system, spec = build_reference_system(urdf, 'reference_literal', mesh_representation=mesh_representation)
    if prepare is not None:
        prepare(system)
    n_j = spec['n_joints']
    gen = torch.Generator().manual_seed(seed)
    quat = torch.randn((n_traj, 4), generator=gen)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    pos = torch.cat((0.05 * torch.randn((n_traj, 2), generator=gen), 0.05 + 0.07 * torch.rand((n_traj, 1), generator=gen)), -1)
    # (a prismatic joint's coordinate is metres: centimetres of travel, decimetres per second)
    sliding = torch.tensor([body['joint_kind'] == 'prismatic' for body in spec['bodies'][1:]], dtype=torch.bool)
    joints = torch.where(sliding, 0.03, 1.5) * torch.randn((n_traj, n_j), generator=gen)
    vel = torch.cat((4.0 * torch.randn((n_traj, 3), generator=gen), 0.4 * torch.randn((n_traj, 3), generator=gen),
                     torch.where(sliding, 0.3, 3.0) * torch.randn((n_traj, n_j), generator=gen)), -1)
    x_0 = torch.cat((quat, pos, joints, vel), -1)
    with torch.no_grad():
        traj, _ = system.simulate(x_0.unsqueeze(-2), torch.zeros((n_traj, 1)), steps)
    x = traj[:, :-1][:, ::keep_every].reshape(-1, traj.shape[-1]).clone()
    x_plus = traj[:, 1:][:, ::keep_every].reshape(-1, traj.shape[-1]).clone()
    return x, x_plus


def record_general_cases(n_traj: int = 8, steps: int = 36, keep_every: int = 3, seed: int = 0,
                         names=('chain3', 'vee', 'ballcube', 'mace', 'gripper', 'crank', 'slider')) -> None:
    """models beyond the cube / elbow topologies -- three-link serial chain, branching tree, several
    geometries on one body, spheres, four links on three hinges with one link bare of geometry, links whose
    inertial / collision / joint frames are all turned by an rpy, a prismatic joint (this repository's own URDFs under
    assets/) -- through the reference's own
    MultibodyTerms / contactnets_loss / forward_dynamics / simulate, exactly as `record_case` does for the reference's
    assets.  Inputs: `general_tosses`."""
    for name in names:
        urdf = os.path.join(REPO, 'assets', name + '.urdf')
        x, x_plus = general_tosses(urdf, n_traj, steps, keep_every, seed)
        record_case(name + '_literal', urdf, x, x_plus, 'reference_literal', sim_steps=3)


def record_welded_case(n_traj: int = 8, steps: int = 36, keep_every: int = 3, seed: int = 0) -> None:
    """a URDF whose `fixed` joints weld links that carry mass (assets/welded_arm.urdf: five links, two of them
    moving against each other).  The reference's parameter tree has one theta row per Drake body -- (5, 10) here -- and its
    LagrangianTerms.forward converts every row and hands all of them to the mass-matrix / force closures
    (multibody_terms.py:228-234), which here sum over the rows with a welded link riding on its host (O.mass_matrix)."""
    urdf = os.path.join(REPO, 'assets', 'welded_arm.urdf')
    x, x_plus = general_tosses(urdf, n_traj, steps, keep_every, seed)
    record_case('welded_arm_literal', urdf, x, x_plus, 'reference_literal', sim_steps=3)
    record_case('welded_arm_physical', urdf, x[::4].clone(), x_plus[::4].clone(), 'physical', sim_steps=2)


def record_polygon_cases(n_traj: int = 8, steps: int = 36, keep_every: int = 3, seed: int = 0) -> None:
    """SURVEY 8f-4, Polygon (geometry.py:220-252: a learnable vertex set, support query = the 4 vertices furthest along
    the direction) through the reference's own classes: the cube with its mesh read as the 8-vertex polygon (the
    vertices moved off the symmetric corners by seeded millimetre noise, so no two are tied), and a 6-vertex wedge with
    a hinged box flap."""
    def jitter(system):
        gen = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for geometry in system.multibody_terms.contact_terms.geometries:
                if isinstance(geometry, Polygon):
                    geometry.vertices.add_(0.003 * torch.randn(geometry.vertices.shape, generator=gen))
    for name, urdf in (('polycube', os.path.join(ASSETS, 'contactnets_cube_mesh.urdf')),
                       ('wedge', os.path.join(REPO, 'assets', 'wedge.urdf'))):
        x, x_plus = general_tosses(urdf, n_traj, steps, keep_every, seed, 'polygon', jitter)
        record_case(name + '_literal', urdf, x, x_plus, 'reference_literal', sim_steps=3, mesh_representation='polygon',
                    prepare=jitter)


def pair_tosses(urdf: str, representation: str, n_traj: int, steps: int, keep_every: int, seed: int):
    """`general_tosses` started with the joints folded so that the body-body pair is about to meet: joint angles drawn
    uniformly, kept when the pair's signed distance (a function of the joints alone) is between 0 and 8 mm."""
    system, spec = build_reference_system(urdf, 'reference_literal', mesh_representation=representation)
    n_j = spec['n_joints']
    gen = torch.Generator().manual_seed(seed)
    trial = torch.zeros((4000, 7 + n_j))
    trial[:, 0] = 1.0
    trial[:, 7:] = (2 * torch.rand((4000, n_j), generator=gen) - 1) * np.pi
    with torch.no_grad():
        phi = system.multibody_terms.contact_terms(trial)[0][:, -len(spec['pairs']):]
    # several candidates: toss i starts with candidate i mod n about to meet, none overlapping
    free = (phi > 0).all(dim=-1)
    picks = [trial[free & (phi[:, p] < 0.008)][:-(-n_traj // phi.shape[1]), 7:] for p in range(phi.shape[1])]
    joints = torch.stack([picks[i % len(picks)][i // len(picks)] for i in range(n_traj)])
    assert joints.shape[0] == n_traj
    quat = torch.randn((n_traj, 4), generator=gen)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    pos = torch.cat((0.05 * torch.randn((n_traj, 2), generator=gen), 0.06 + 0.07 * torch.rand((n_traj, 1), generator=gen)), -1)
    vel = torch.cat((4.0 * torch.randn((n_traj, 3), generator=gen), 0.4 * torch.randn((n_traj, 3), generator=gen),
                     3.0 * torch.randn((n_traj, n_j), generator=gen)), -1)
    x_0 = torch.cat((quat, pos, joints, vel), -1)
    with torch.no_grad():
        traj, _ = system.simulate(x_0.unsqueeze(-2), torch.zeros((n_traj, 1)), steps)
    x = traj[:, :-1][:, ::keep_every].reshape(-1, traj.shape[-1]).clone()
    x_plus = traj[:, 1:][:, ::keep_every].reshape(-1, traj.shape[-1]).clone()
    return x, x_plus


def record_pair_cases(n_traj: int = 10, steps: int = 36, keep_every: int = 3, seed: int = 0,
                      cases=(('clasp', 'deep_support'), ('clasp_ball', 'polygon'), ('vee_pair', 'deep_support'),
                             ('pincer', 'deep_support'), ('grasp', 'deep_support')), mesh_case: bool = True) -> None:
    """SURVEY 8f-4, body-body contact: the reference's GeometryCollider.collide_mesh_mesh / ContactTerms.forward pair
    path (geometry.py:585-643, multibody_terms.py:428-521) on a base and a tip that fold onto each other -- box against
    box, a sphere against a polygon (the pair swapped into the reference's type order), and the two arms of a branching
    tree against each other (both members move with a joint of their own; `pincer`: the same with every frame turned by an
    rpy), and a fingertip that can meet the palm or the thumb (`grasp`: two candidates) -- with fcl's direction supplied by
    DirectionSearchFcl.  Inputs: `pair_tosses`."""
    for name, representation in cases:
        urdf = os.path.join(REPO, 'assets', name + '.urdf')
        x, x_plus = pair_tosses(urdf, representation, n_traj, steps, keep_every, seed)
        record_case(name + '_literal', urdf, x, x_plus, 'reference_literal', sim_steps=3, mesh_representation=representation)
    if not mesh_case:
        return
    # the reference's own case: two DeepSupportConvex shapes (its classes unmodified; fcl sees their extracted meshes)
    urdf = os.path.join(REPO, 'assets', 'clasp_mesh.urdf')
    x, x_plus = pair_tosses(urdf, 'deep_support', 6, 36, 3, seed)
    record_case('clasp_mesh_literal', urdf, x, x_plus, 'reference_literal', sim_steps=2)


FOREST_CASES = {
    # name -> init_urdfs of the system (this repository's assets)
    'chain6': {'chain6': 'chain6.urdf'},                                  # 5 hinges, 6 boxes, 10 body-body candidates
    'rake': {'rake': 'rake.urdf'},                                        # one body, 5 geometries (2 boxes, 3 spheres)
    'two_cubes': {'cube_a': 'cube.urdf', 'cube_b': 'cube.urdf'},          # two models: two free cubes that can collide
    'pendulum_cube': {'pendulum': 'pendulum.urdf', 'cube': 'cube.urdf'},  # a fixed-base model next to a free one
    # actuated systems (record_actuated_forest): motors on three of five hinges; a motor on a fixed-base model next to a free cube
    'chain6_actuated': {'chain6_actuated': 'chain6_actuated.urdf'},
    'pendulum_actuated_cube': {'pendulum': 'pendulum_actuated.urdf', 'cube': 'cube.urdf'},
}
ACTUATED_FOREST_CASES = ('chain6_actuated', 'pendulum_actuated_cube')


def forest_tosses(name: str, n_traj: int, steps: int, keep_every: int, seed: int):
    """(x, x_plus) pairs of seeded tosses of a system of several models / a long tree, rolled out by the reference's own
    simulate on the stub-built system: every free model starts with a random attitude 5-12 cm above the ground, spinning,
    the models of a system 8 cm apart and moving towards each other; joints anywhere (the candidates of a long chain may
    start overlapping: the contact model takes that, and both sides of the parity check see the same states)."""
    urdfs = {key: os.path.join(REPO, 'assets', value) for key, value in FOREST_CASES[name].items()}
    system, spec = build_reference_system(urdfs, 'reference_literal')
    models = spec.get('models', [spec])
    gen = torch.Generator().manual_seed(seed)
    qs, vs = [], []
    for m, model in enumerate(models):
        n_j = model['n_joints']
        joints = 1.2 * torch.randn((n_traj, n_j), generator=gen)
        rates = 3.0 * torch.randn((n_traj, n_j), generator=gen)
        if n_j > 0 and model['pairs'] and len(models) == 1:
            # a long chain: joint angles drawn until no two links overlap (a function of the joints alone), half of the tosses
            # with some pair of links within 2 cm of each other -- as pair_tosses starts the short trees
            trial = torch.zeros((6000, 7 + n_j))
            trial[:, 0] = 1.0
            trial[:, 7:] = 1.6 * torch.randn((6000, n_j), generator=gen)
            with torch.no_grad():
                phi = system.multibody_terms.contact_terms(trial)[0][:, -len(model['pairs']):]
            free = (phi > 0.002).all(dim=-1)
            near = free & (phi.min(dim=-1).values < 0.02)
            picks = torch.cat((trial[near][:n_traj // 2, 7:], trial[free & ~near][:n_traj - min(n_traj // 2, int(near.sum())), 7:]))
            assert picks.shape[0] == n_traj, (int(free.sum()), int(near.sum()))
            joints = picks
        if model['fixed_base']:
            qs.append(joints)
            vs.append(rates)
            continue
        quat = torch.randn((n_traj, 4), generator=gen)
        quat = quat / quat.norm(dim=-1, keepdim=True)
        side = (m - 0.5 * (len(models) - 1)) * 0.08
        pos = torch.cat((side + 0.01 * torch.randn((n_traj, 1), generator=gen), 0.02 * torch.randn((n_traj, 1), generator=gen),
                         0.05 + 0.07 * torch.rand((n_traj, 1), generator=gen)), -1)
        lin = torch.cat((-6.0 * side + 0.3 * torch.randn((n_traj, 1), generator=gen), 0.4 * torch.randn((n_traj, 2), generator=gen)), -1)
        if name in ('pendulum_cube', 'pendulum_actuated_cube'):  # the cube is thrown at the mast / under the arm
            pos = torch.cat((0.2 - 0.1 + 0.02 * torch.randn((n_traj, 1), generator=gen), 0.1 + 0.02 * torch.randn((n_traj, 1), generator=gen),
                             0.05 + 0.05 * torch.rand((n_traj, 1), generator=gen)), -1)
            lin = torch.cat((0.8 + 0.3 * torch.randn((n_traj, 1), generator=gen), 0.3 * torch.randn((n_traj, 2), generator=gen)), -1)
        qs.append(torch.cat((quat, pos, joints), -1))
        vs.append(torch.cat((4.0 * torch.randn((n_traj, 3), generator=gen), lin, rates), -1))
    x_0 = torch.cat(qs + vs, -1)
    with torch.no_grad():
        traj, _ = system.simulate(x_0.unsqueeze(-2), torch.zeros((n_traj, 1)), steps)
    x = traj[:, :-1][:, ::keep_every].reshape(-1, traj.shape[-1]).clone()
    x_plus = traj[:, 1:][:, ::keep_every].reshape(-1, traj.shape[-1]).clone()
    return urdfs, x, x_plus


def record_forest_cases(n_traj: int = 8, steps: int = 36, keep_every: int = 3, seed: int = 0,
                        names=tuple(name for name in FOREST_CASES if name not in ACTUATED_FOREST_CASES)) -> None:
    """SURVEY 8f-3, what the reference's generality reaches beyond one short tree: several models in one system (init_urdfs with
    more than one entry: a ProductSpace of the models' spaces, candidates between the models), a model welded to the world
    (FixedBaseSpace), five joints, five geometries on a body, ten candidates -- through the reference's own MultibodyTerms /
    contactnets_loss / forward_dynamics / simulate exactly as record_case does for its assets."""
    for name in names:
        urdfs, x, x_plus = forest_tosses(name, n_traj, steps, keep_every, seed)
        record_case(name + '_literal', urdfs if len(urdfs) > 1 or name == 'pendulum_cube' else next(iter(urdfs.values())), x, x_plus,
                    'reference_literal', sim_steps=3)


def record_actuated_forest(n_traj: int = 6, steps: int = 36, keep_every: int = 3, seed: int = 0) -> None:
    """B u (reference multibody_terms.py:142-146) on systems beyond one short tree: a six-link chain with motors on three of its
    hinges (listed out of joint order: column k of u is the k-th <transmission>), and a fixed-base pendulum with a motor on its
    pivot next to a free cube (the plant's actuators are the models' one after the other) -- seeded torques of up to 0.05 N m
    through the reference's own contactnets_loss / forward_dynamics / MultibodyTerms with a non-empty u."""
    for name in ACTUATED_FOREST_CASES:
        urdfs, x, x_plus = forest_tosses(name, n_traj, steps, keep_every, seed)
        _, spec = build_reference_system(urdfs, 'reference_literal')
        n_u = len(spec['actuators'])
        u = 0.1 * (torch.rand((x.shape[0], n_u), generator=torch.Generator().manual_seed(13)) - 0.5)
        record_case(name + '_literal', urdfs if len(urdfs) > 1 else next(iter(urdfs.values())), x, x_plus, 'reference_literal', sim_steps=3, u=u)


def record_actuated_elbow(name: str = 'elbow_actuated_literal') -> None:
    """The B u path (reference multibody_terms.py:142-146, 235-236; multibody_learnable_system.py:199-203): the elbow with a
    <transmission> on its hinge (this repository's assets/elbow_actuated.urdf: none of the reference's URDFs has an actuator),
    the synthetic elbow pairs and seeded torques of up to 0.5 N m through the reference's own contactnets_loss / forward_dynamics /
    MultibodyTerms with a non-empty u."""
    ex, exp_ = elbow_pairs()
    u = (torch.rand((ex.shape[0], 1), generator=torch.Generator().manual_seed(11)) - 0.5)
    record_case(name, os.path.join(REPO, 'assets', 'elbow_actuated.urdf'), ex, exp_, 'reference_literal', u=u)


def record_elbow_mesh() -> None:
    """contactnets_elbow_mesh.urdf: a DeepSupportConvex on each link (two independent networks), on every 4th of the
    synthetic elbow pairs."""
    ex, exp_ = elbow_pairs()
    record_case('elbow_mesh_literal', os.path.join(ASSETS, 'contactnets_elbow_mesh.urdf'), ex[::4].clone(),
                exp_[::4].clone(), 'reference_literal', sim_steps=3)


def main() -> None:
    cube = os.path.join(ASSETS, 'contactnets_cube.urdf')
    record_bench_batch('cube_box_4096')
    x, xp = cube_pairs([0, 1, 2])
    record_case('cube_box_literal', cube, x, xp, 'reference_literal')
    record_case('cube_box_physical', cube, x[::4], xp[::4], 'physical')
    record_case('cube_box_config1', cube, x[:1], xp[:1], 'reference_literal')  # BASELINE configs[0]
    ex, exp_ = elbow_pairs()
    record_case('elbow_box_literal', os.path.join(ASSETS, 'contactnets_elbow.urdf'), ex, exp_, 'reference_literal')
    mx, mxp = cube_pairs([3], stride=2)
    record_case('cube_mesh_literal', os.path.join(ASSETS, 'contactnets_cube_mesh.urdf'), mx, mxp,
                'reference_literal')
    record_elbow_bench_batch()
    record_mesh_bench_batch()
    record_actuated_elbow()
    record_slice_fixture()
    record_dynamics_gradients()
    record_general_cases()
    record_welded_case()
    record_elbow_mesh()
    record_polygon_cases()
    record_pair_cases()
    record_forest_cases()
    record_actuated_forest()


if __name__ == '__main__':
    if len(sys.argv) > 1:  # python oracle/gen_golden.py record_elbow_bench_batch record_slice_fixture ...
        for _fn in sys.argv[1:]:
            globals()[_fn]()
    else:
        main()
