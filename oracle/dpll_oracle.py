"""CPU oracle for the dair_pll contact-dynamics hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU (float64 by default) restatement of the reference
algorithm on the path named by BASELINE.json `north_star`:

    MultibodyTerms(q, v, u) -> Anitescu cone-QP solve -> contactnets_loss (fwd, autograd bwd)
    forward_dynamics -> VelocityIntegrator.step -> simulate

It is the *checker*.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; nothing under ``dair_pll_amd/`` does (the product path
fails loudly when the HIP library is missing instead of falling back to this file).

Pinning status (see DESIGN.md "Oracle"):
  * All torch arithmetic that lives under /root/reference/dair_pll is pinned: the
    reference's own ``contactnets_loss``, ``forward_dynamics``, ``Integrator.simulate``,
    ``ContactTerms.forward``, ``LagrangianTerms.forward``, ``GeometryCollider``, ``Box``,
    ``HomogeneousICNN``, ``InertialParameterConverter`` and ``quaternion``/``state_space``
    were executed in the authoring container (``oracle/gen_golden.py``) and their outputs
    are committed as ``tests/golden/*.npz``; this file reproduces them to <=1e-12.
  * PARITY UNPINNED for the two third-party pieces the reference delegates to and that are
    absent from /root/reference and from this image:
      - ``sappy.SAPSolver`` (git+https://github.com/mshalm/sappy.git, no version pin,
        reference setup.py:40-43) -- the cone QP.  Restated here from its published
        problem statement (unique minimiser of a strictly convex QP, call sites
        multibody_learnable_system.py:181-184, 295-298) and verified by KKT residuals.
      - ``pydrake`` + ``drake_pytorch`` (unpinned, setup.py:33-43) -- symbolic M(q), F(q,v)
        and geometry kinematics.  Restated here as textbook spatial-vector rigid-body
        dynamics following the *definitions* at multibody_terms.py:123-157, 355-376 and
        verified by independent physical invariants (tests/test_oracle_physics.py).

Conventions (reference state_space.py:412-424): q = [quat wxyz, p_W, joint angles],
v = [omega_body, v_W, joint rates]; spatial vectors are [angular; linear] in BODY
coordinates at the body origin.
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

GRAVITY_Z = -9.81  # Drake's default UniformGravityField (used through CalcGravityGeneralizedForces,
#                    reference multibody_terms.py:142-144)
GROUND_MU = 1.0  # reference drake_utils.py:280-288
N_QUERY = 4  # witness points per convex geometry, reference geometry.py:47-48, 490
LOSS_EPS = 1e-3  # reference multibody_learnable_system.py:130
DYNAMICS_EPS = 1e-4  # reference multibody_learnable_system.py:283, 298
INVALID_FORCE = 1e3  # reference multibody_learnable_system.py:187


# --------------------------------------------------------------------------------------
# URDF -> model spec (replaces Drake's parser for the assets the path uses)
# --------------------------------------------------------------------------------------
def _floats(text: Optional[str], n: int, default: float = 0.0) -> List[float]:
    if text is None:
        return [default] * n
    vals = [float(t) for t in text.split()]
    assert len(vals) == n, f'expected {n} floats, got {text!r}'
    return vals


def _load_obj_vertices(path: str) -> List[List[float]]:
    verts = []
    with open(path, 'r', encoding='utf8') as handle:
        for line in handle:
            parts = line.split()
            if len(parts) >= 4 and parts[0] == 'v':
                verts.append([float(parts[1]), float(parts[2]), float(parts[3])])
    return verts


def rpy_matrix(rpy: List[float]) -> List[List[float]]:
    """URDF ``rpy``: fixed-axis roll (x), pitch (y), yaw (z), i.e. R = Rz(yaw) Ry(pitch) Rx(roll) -- the convention
    Drake's parser applies to every ``<origin>`` (RollPitchYaw)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr]]


def _origin_rotation(origin) -> List[List[float]]:
    return rpy_matrix(_floats(origin.get('rpy') if origin is not None else None, 3))


_IDENTITY = [[1., 0., 0.], [0., 1., 0.], [0., 0., 1.]]


def _fold_welds(links: Dict, order: List[str], joints: List, welds: List):
    """Links welded to another link by a `fixed` joint.  In Drake such a link stays a body of its own -- with its own spatial
    inertia, hence its own row of the reference's inertial_parameters (multibody_terms.py:161-207) -- that has no coordinate:
    it rides on the link it is welded to.  Here it is taken off the list of moving bodies: its frame in its host (the first
    link up the welds that is not itself welded on) is the product of the joint <origin>s on the way; its collision geometries
    and the joints hanging off it are re-expressed in the host's frame; its inertia is NOT merged into the host's -- it
    stays a row of its own, carried by the host (mass_matrix / lagrangian_forces sum over the rows).
    Returns ({welded link with mass: (host, origin, rot, the link's own dict)}, {every welded link: host}); `order` and `joints`
    are edited in place."""
    def mat(a, b):
        return [[sum(a[i][k] * b[k][j] for k in range(3)) for j in range(3)] for i in range(3)]

    def at(rot, origin, point):
        return [origin[i] + sum(rot[i][k] * point[k] for k in range(3)) for i in range(3)]

    direct = {child: (parent, origin, rot) for parent, child, origin, rot in welds}
    frames = {}
    for name in direct:
        host, origin, rot = direct[name]
        while host in direct:  # welded onto a welded link: one more frame on the way up
            up, up_origin, up_rot = direct[host]
            origin, rot, host = at(up_rot, up_origin, origin), mat(up_rot, rot), up
        frames[name] = (host, origin, rot)
    carried = {}
    for name, (host, origin, rot) in frames.items():
        link = links[name]
        for geom in link['geoms']:
            moved = dict(geom)
            moved['origin'] = at(rot, origin, geom['origin'])
            moved['rot'] = mat(rot, geom['rot'])
            links[host]['geoms'].append(moved)
        if link['mass'] > 0.0:
            carried[name] = (host, origin, rot, link)
        else:
            assert link['mass'] == 0.0 and not any(link['inertia_cm']), 'a welded link needs a positive mass or no inertia at all'
        order.remove(name)
    for index, (parent, child, j_origin, axis, j_rot, kind) in enumerate(joints):
        if parent in frames:
            host, origin, rot = frames[parent]
            joints[index] = (host, child, at(rot, origin, j_origin), axis, mat(rot, j_rot), kind)
    return carried, {name: frame[0] for name, frame in frames.items()}


def parse_urdf(path: str, mesh_representation: str = 'deep_support') -> Dict:
    """Parses one floating-base serial chain with revolute joints and box / mesh collision
    geometry.  Mirrors what the reference obtains from Drake: bodies with (m, com, I_cm)
    (multibody_terms.py:161-207), one joint per non-root link, collision boxes
    (geometry.py:486-490) or meshes (geometry.py:499-504), mu_static per geometry
    (drake_utils.py:192-197).  mesh_representation='polygon': a mesh element becomes a Polygon over
    the OBJ's vertices (geometry.py:220-252) instead of a DeepSupportConvex."""
    root = ET.parse(path).getroot()
    links = {}
    order = []
    for link in root.findall('link'):
        name = link.get('name')
        if name == 'world':
            continue
        inertial = link.find('inertial')
        origin = inertial.find('origin')
        inertia = inertial.find('inertia')
        # the inertia tensor is given in the inertial frame; Drake hands the reference the body-frame one
        # (CalcSpatialInertiaInBodyFrame, multibody_terms.py:161-207): I_B = R I R^T
        r_bi = _origin_rotation(origin)
        ixx, iyy, izz, ixy, ixz, iyz = [float(inertia.get(k)) for k in ('ixx', 'iyy', 'izz', 'ixy', 'ixz', 'iyz')]
        i_in = [[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]]
        i_b = [[sum(r_bi[a][k] * i_in[k][l] * r_bi[b][l] for k in range(3) for l in range(3)) for b in range(3)]
               for a in range(3)]
        body = {
            'name': name,
            'mass': float(inertial.find('mass').get('value')),
            'com': _floats(origin.get('xyz') if origin is not None else None, 3),
            'inertia_cm': [i_b[0][0], i_b[1][1], i_b[2][2], i_b[0][1], i_b[0][2], i_b[1][2]],
            'geoms': [],
            'parent': -1,
            'joint_origin': None,
            'joint_axis': None,
            'joint_rot': None,  # orientation of the joint (= child at angle 0) frame in the parent frame
            'joint_kind': 'revolute',  # or 'prismatic': the child slides along the axis
        }
        for col in link.findall('collision'):
            c_origin = col.find('origin')
            mu = None
            for element in col.iter():
                if element.tag.endswith('mu_static'):
                    mu = float(element.get('value'))
            assert mu is not None, 'collision without drake:mu_static'
            geometry = col.find('geometry')
            geom = {'origin': _floats(c_origin.get('xyz') if c_origin is not None else None, 3), 'mu': mu, 'link': name,
                    'rot': _origin_rotation(c_origin)}  # R_BG (inspector.GetPoseInFrame, multibody_terms.py:356)
            if geometry.find('box') is not None:
                size = _floats(geometry.find('box').get('size'), 3)
                geom.update(kind='box', half=[0.5 * s for s in size])
            elif geometry.find('sphere') is not None:
                geom.update(kind='sphere', radius=float(geometry.find('sphere').get('radius')))
            elif geometry.find('mesh') is not None:
                filename = geometry.find('mesh').get('filename')
                mesh_path = os.path.join(os.path.dirname(os.path.abspath(path)), filename)
                geom.update(kind='mesh' if mesh_representation == 'deep_support' else 'polygon', file=filename,
                            vertices=_load_obj_vertices(mesh_path))
            else:
                raise TypeError('unsupported collision geometry')
            body['geoms'].append(geom)
        links[name] = body
        order.append(name)
    children = set()
    joints = []
    welds = []
    for joint in root.findall('joint'):
        j_origin = joint.find('origin')
        parent = joint.find('parent').get('link')
        child = joint.find('child').get('link')
        if parent == 'world' and joint.get('type') == 'fixed':
            # a model welded to the world (Drake welds a link named `world`; the reference gives the model a FixedBaseSpace,
            # drake_utils.py:329-332): its root has no coordinates; the mount is the joint's <origin>
            links[child]['mount'] = (_floats(j_origin.get('xyz') if j_origin is not None else None, 3), _origin_rotation(j_origin))
            continue
        assert joint.get('type') in ('continuous', 'revolute', 'prismatic', 'fixed'), 'only revolute, prismatic and fixed joints'
        if joint.get('type') == 'fixed':
            # a link welded to another: Drake keeps it as a body of its own with no coordinate (see _fold_welds below)
            welds.append((parent, child, _floats(j_origin.get('xyz') if j_origin is not None else None, 3), _origin_rotation(j_origin)))
            continue
        axis = _floats(joint.find('axis').get('xyz'), 3) if joint.find('axis') is not None else [1., 0., 0.]
        norm = math.sqrt(sum(a * a for a in axis))
        joints.append((parent, child, _floats(j_origin.get('xyz') if j_origin is not None else None, 3),
                       [a / norm for a in axis], _origin_rotation(j_origin),
                       'prismatic' if joint.get('type') == 'prismatic' else 'revolute'))
        children.add(child)
    link_order = list(order)
    carried, hosts = _fold_welds(links, order, joints, welds)
    children = {child for _, child, *_ in joints}
    roots = [n for n in order if n not in children]
    assert len(roots) == 1, 'one chain per file (reference drake_utils.py:309-335)'
    # breadth-first order from the root: body 0 is the floating base.
    sorted_names = [roots[0]]
    for name in sorted_names:
        for parent, child, origin, axis, rotation, kind in joints:
            if parent == name:
                links[child]['parent'] = sorted_names.index(parent)
                links[child]['joint_origin'] = origin
                links[child]['joint_axis'] = axis
                links[child]['joint_rot'] = rotation
                links[child]['joint_kind'] = kind
                sorted_names.append(child)
    assert len(sorted_names) == len(order)
    bodies = [links[n] for n in sorted_names]
    spec = {'name': root.get('name'), 'bodies': bodies, 'n_joints': len(bodies) - 1, 'ground_mu': GROUND_MU}
    spec['fixed_base'] = 'mount' in bodies[0]
    # the rows of the reference's inertial_parameters: one per Drake body (multibody_terms.py:161-207, drake_utils.py:129-146).
    # Without welded links these are the bodies, in body order; with them every link in document order -- Drake's body index
    # order -- each with its own (m, com, I_cm), the body that carries it and its frame there.
    if carried:
        spec['inertia_rows'] = []
        for name in link_order:
            if name in sorted_names:
                body = links[name]
                spec['inertia_rows'].append({'name': name, 'body': sorted_names.index(name), 'mass': body['mass'], 'com': body['com'],
                                             'inertia_cm': body['inertia_cm'], 'origin': [0., 0., 0.], 'rot': _IDENTITY})
            elif name in carried:
                host, origin, rot, link = carried[name]
                spec['inertia_rows'].append({'name': name, 'body': sorted_names.index(host), 'mass': link['mass'], 'com': link['com'],
                                             'inertia_cm': link['inertia_cm'], 'origin': origin, 'rot': rot})
    spec['pairs'] = collision_candidates(root, spec)
    # actuators: one JointActuator per <transmission> in file order (Drake's parser), each a generalized force on its joint's
    # coordinate -- the B of reference multibody_terms.py:142-146.  Entry k = index of the actuated joint (joint j drives body j + 1)
    joint_child = {joint.get('name'): joint.find('child').get('link') for joint in root.findall('joint')}
    spec['actuators'] = [sorted_names.index(joint_child[t.find('joint').get('name')]) - 1 for t in root.findall('transmission')]
    return spec


def system_spec(urdfs, mesh_representation: str = 'deep_support') -> Dict:
    """One spec for the models of a system -- ``init_urdfs: Dict[str, str]`` of the reference's constructor
    (multibody_learnable_system.py:51-54), or one path.  Bodies of all models in order, every body with the place of its joint's
    coordinates in the ProductSpace state (drake_utils.py:309-335, state_space.py:650-730: per model [quaternion, position,
    joint coordinates] / [omega_body, v_world, joint rates]; a fixed-base model: joint coordinates only); candidates inside
    each model as collision_candidates finds them plus every geometry of one model against every geometry of another."""
    if isinstance(urdfs, str):
        urdfs = {'model': urdfs}
    specs = [parse_urdf(path, mesh_representation) for path in urdfs.values()]
    if len(specs) == 1 and not specs[0]['fixed_base']:
        return specs[0]
    bodies, own, model_of_geom = [], set(), []
    q_off = v_off = geom_off = 0
    for m, spec in enumerate(specs):
        first = len(bodies)
        fixed = spec['fixed_base']
        base_q, base_v = (0, 0) if fixed else (7, 6)
        for index, body in enumerate(spec['bodies']):
            entry = dict(body)
            entry['model'] = m
            if index == 0:
                entry.update(parent=-1, q_index=q_off, v_index=v_off, fixed=fixed)
            else:
                entry.update(parent=first + body['parent'], q_index=q_off + base_q + index - 1, v_index=v_off + base_v + index - 1)
            bodies.append(entry)
        own |= {(geom_off + a, geom_off + b) for a, b in spec['pairs']}
        n_geoms = sum(len(body['geoms']) for body in spec['bodies'])
        model_of_geom += [m] * n_geoms
        geom_off += n_geoms
        q_off += base_q + spec['n_joints']
        v_off += base_v + spec['n_joints']
    rows, first = [], 0
    for spec in specs:
        rows += [dict(row, body=first + row['body']) for row in inertia_rows(spec)]
        first += len(spec['bodies'])
    merged = {'name': '+'.join(urdfs.keys()), 'bodies': bodies, 'n_joints': sum(spec['n_joints'] for spec in specs),
              'ground_mu': GROUND_MU, 'n_q': q_off, 'n_v': v_off, 'models': specs, 'fixed_base': False}
    if any('inertia_rows' in spec for spec in specs):
        merged['inertia_rows'] = rows
    # actuators of the plant: the models' <transmission>s one after the other (Drake's JointActuator order); an entry is the index
    # of the actuated joint as lagrangian_forces reads it: joint j drives body j + 1 of the merged list
    merged['actuators'], first = [], 0
    for spec in specs:
        merged['actuators'] += [first + joint for joint in spec.get('actuators', [])]
        first += len(spec['bodies'])
    table = geometry_table(merged)
    anchored = anchored_bodies(merged)
    pairs = []
    for ia in range(1, len(table)):
        for ib in range(ia + 1, len(table)):
            swap = _TYPE_ORDER[table[ia]['kind']] > _TYPE_ORDER[table[ib]['kind']]
            pair = (ib, ia) if swap else (ia, ib)
            if table[ia]['body'] in anchored and table[ib]['body'] in anchored:
                continue  # (both welded to the world: Drake filters anchored-anchored candidates)
            if model_of_geom[ia - 1] != model_of_geom[ib - 1] or pair in own:
                pairs.append(pair)
    merged['pairs'] = pairs
    return merged


def state_sizes(spec: Dict) -> Tuple[int, int]:
    """(n_q, n_v) of a spec: one floating-base model, or what system_spec recorded"""
    return spec.get('n_q', 7 + spec['n_joints']), spec.get('n_v', 6 + spec['n_joints'])


PAIR_TIE = 1e-12  # metres: support values closer than this are a tie (witness vertex of a body-body contact)
_TYPE_ORDER = {'plane': 0, 'polygon': 1, 'box': 2, 'sphere': 3, 'mesh': 4}  # reference geometry.py:46


def collision_candidates(root, spec: Dict) -> List[Tuple[int, int]]:
    """Body-body pairs of ContactTerms.collision_candidates (reference multibody_terms.py:286-297) as indices into
    geometry_table: what Drake's GetCollisionCandidates (drake_utils.py:178-184) keeps beyond the ground pairs --
    geometries of two bodies that no joint connects (Drake filters adjacent bodies) and no drake:collision_filter_group
    excludes (assets/contactnets_elbow.urdf:74-78) -- each pair swapped into the reference's type order (:294-297)."""
    groups, ignores = {}, []
    for element in root:
        if element.tag.endswith('collision_filter_group'):
            groups[element.get('name')] = {m.get('link') for m in element if m.tag.endswith('member')}  # (groups name LINKS)
            ignores += [(element.get('name'), i.get('name')) for i in element
                        if i.tag.endswith('ignored_collision_filter_group')]
    excluded = set()
    for first, second in ignores:
        for a in groups.get(first, ()):
            for b in groups.get(second, ()):
                excluded |= {(a, b), (b, a)}
    table = geometry_table(spec)
    adjacent = set()
    for joint in root.findall('joint'):
        ends = (joint.find('parent').get('link'), joint.find('child').get('link'))
        adjacent |= {ends, ends[::-1]}
    pairs = []
    for ia in range(1, len(table)):
        for ib in range(ia + 1, len(table)):
            ba, bb = table[ia]['body'], table[ib]['body']
            if ba == bb or (table[ia]['link'], table[ib]['link']) in excluded:
                continue
            # Drake's default filter is between the two LINKS a joint connects (welded-together links, which are one body here,
            # are filtered as a set: ba == bb above) -- a link welded onto a joint's parent is NOT adjacent to the joint's child
            if (table[ia]['link'], table[ib]['link']) in adjacent:
                continue
            swap = _TYPE_ORDER[table[ia]['kind']] > _TYPE_ORDER[table[ib]['kind']]
            pairs.append((ib, ia) if swap else (ia, ib))
    return pairs


def anchored_bodies(spec: Dict) -> set:
    """Bodies welded to the world: the root of a fixed-base model (links welded to it are folded into it when the URDF is
    parsed).  Drake calls their geometries ANCHORED and filters every anchored-anchored pair out of GetCollisionCandidates
    (drake_utils.py:178-184) -- the ground half-space sits on the world body, so an anchored geometry has no ground contacts and
    no candidate with the anchored geometry of another fixed-base model."""
    return {index for index, body in enumerate(spec['bodies'])
            if body['parent'] < 0 and (body.get('fixed') or spec.get('fixed_base'))}


def ground_geometries(spec: Dict) -> List[int]:
    """indices into geometry_table of the geometries that collide with the ground: those of bodies that can move"""
    anchored = anchored_bodies(spec)
    return [index for index, geom in enumerate(geometry_table(spec)) if index > 0 and geom['body'] not in anchored]


def geometry_table(spec: Dict) -> List[Dict]:
    """Geometry list in the order [ground, body geometries...] (quirk Q8: the reference takes
    Drake's GetCollisionCandidates order, drake_utils.py:178-184; the A/B swap at
    multibody_terms.py:294-297 puts the Plane first in every pair)."""
    table = [{'kind': 'plane', 'body': -1, 'origin': [0., 0., 0.], 'mu': spec['ground_mu']}]
    for index, body in enumerate(spec['bodies']):
        for geom in body['geoms']:
            entry = dict(geom)
            entry['body'] = index
            table.append(entry)
    return table


# --------------------------------------------------------------------------------------
# small SO(3) helpers (reference quaternion.py, tensor_utils.py:137-162)
# --------------------------------------------------------------------------------------
def skew(v: Tensor) -> Tensor:
    zero = torch.zeros_like(v[..., 0])
    return torch.stack((torch.stack((zero, -v[..., 2], v[..., 1]), -1),
                        torch.stack((v[..., 2], zero, -v[..., 0]), -1),
                        torch.stack((-v[..., 1], v[..., 0], zero), -1)), -2)


def quat_to_rot(quat: Tensor) -> Tensor:
    """R such that R p == quaternion.rotate(quat, p) (reference quaternion.py:150-164); a
    homogeneous quadratic in quat, NOT normalised (quirk Q2)."""
    w = quat[..., 0:1].unsqueeze(-1)
    xyz = quat[..., 1:]
    eye = torch.eye(3, dtype=quat.dtype).expand(quat.shape[:-1] + (3, 3))
    outer = xyz.unsqueeze(-1) * xyz.unsqueeze(-2)
    return 2 * outer + (w * w - (xyz * xyz).sum(-1)[..., None, None]) * eye + 2 * w * skew(xyz)


def quat_multiply(q: Tensor, r: Tensor) -> Tensor:
    """reference quaternion.py:89-105"""
    qw, qv = q[..., :1], q[..., 1:]
    rw, rv = r[..., :1], r[..., 1:]
    return torch.cat((qw * rw - (qv * rv).sum(-1, keepdim=True),
                      qw * rv + rw * qv + torch.cross(qv, rv, dim=-1)), -1)


def quat_exp(r: Tensor) -> Tensor:
    """reference quaternion.py:276-309 with sinc of :208-229"""
    angle = r.norm(dim=-1, keepdim=True)
    half = angle / 2
    safe = torch.where(half.abs() > 0, half, torch.ones_like(half))
    sinc = torch.where(half.abs() > 0, torch.sin(safe) / safe, torch.ones_like(half))
    return torch.cat((torch.cos(half), r * sinc / 2), -1)


def axis_rotation(axis: Tensor, angle: Tensor) -> Tensor:
    """Rodrigues rotation about a fixed unit axis; angle (B,) -> (B,3,3)."""
    k = skew(axis)
    eye = torch.eye(3, dtype=angle.dtype)
    s = torch.sin(angle)[..., None, None]
    c = torch.cos(angle)[..., None, None]
    return eye + s * k + (1 - c) * (k @ k)


# --------------------------------------------------------------------------------------
# inertial parameterisations (reference inertia.py)
# --------------------------------------------------------------------------------------
def theta_to_pi_o(theta: Tensor) -> Tensor:
    """log-Cholesky theta -> pi_o, reference inertia.py:206-234."""
    alpha, d1, d2, d3, s12, s23, s13, t1, t2, t3 = theta.unbind(-1)
    e1, e2, e3 = torch.exp(d1), torch.exp(d2), torch.exp(d3)
    rows = (t1 * t1 + t2 * t2 + t3 * t3 + 1,
            t1 * e1,
            t1 * s12 + t2 * e2,
            t1 * s13 + t2 * s23 + t3 * e3,
            s12 * s12 + s23 * s23 + s13 * s13 + e2 * e2 + e3 * e3,
            s13 * s13 + s23 * s23 + e1 * e1 + e3 * e3,
            s12 * s12 + e1 * e1 + e2 * e2,
            -s12 * e1,
            -s13 * e1,
            -s12 * s13 - s23 * e2)
    return torch.exp(2 * alpha).unsqueeze(-1) * torch.stack(rows, -1)


def _inertia_matrix(vec6: Tensor) -> Tensor:
    xx, yy, zz, xy, xz, yz = vec6.unbind(-1)
    return torch.stack((torch.stack((xx, xy, xz), -1), torch.stack((xy, yy, yz), -1),
                        torch.stack((xz, yz, zz), -1)), -2)


def _inertia_vector(mat: Tensor) -> Tensor:
    return torch.stack((mat[..., 0, 0], mat[..., 1, 1], mat[..., 2, 2], mat[..., 0, 1], mat[..., 0, 2],
                        mat[..., 1, 2]), -1)


def pi_o_to_pi_cm(pi_o: Tensor) -> Tensor:
    """parallel-axis shift origin -> com, reference inertia.py:305-331 (108-145)."""
    mass = pi_o[..., 0:1]
    com = pi_o[..., 1:4] / mass
    s = skew(com)
    i_cm = _inertia_matrix(pi_o[..., 4:]) + mass.unsqueeze(-1) * (s @ s)
    return torch.cat((mass, com * mass, _inertia_vector(i_cm)), -1)


def pi_cm_to_pi_o(pi_cm: Tensor) -> Tensor:
    """reference inertia.py:334-360."""
    mass = pi_cm[..., 0:1]
    com = pi_cm[..., 1:4] / mass
    s = skew(com)
    i_o = _inertia_matrix(pi_cm[..., 4:]) - mass.unsqueeze(-1) * (s @ s)
    return torch.cat((mass, com * mass, _inertia_vector(i_o)), -1)


def pi_o_to_theta(pi_o: Tensor) -> Tensor:
    """local inverse of theta_to_pi_o, reference inertia.py:237-302."""
    ea_e1 = torch.sqrt(0.5 * (pi_o[..., 5] + pi_o[..., 6] - pi_o[..., 4]))
    ea_s12 = -pi_o[..., 7] / ea_e1
    ea_s13 = -pi_o[..., 8] / ea_e1
    ea_e2 = torch.sqrt(pi_o[..., 6] - ea_e1**2 - ea_s12**2)
    ea_s23 = (-pi_o[..., 9] - ea_s12 * ea_s13) / ea_e2
    ea_e3 = torch.sqrt(pi_o[..., 5] - ea_e1**2 - ea_s13**2 - ea_s23**2)
    ea_t1 = pi_o[..., 1] / ea_e1
    ea_t2 = (pi_o[..., 2] - ea_t1 * ea_s12) / ea_e2
    ea_t3 = (pi_o[..., 3] - ea_t1 * ea_s13 - ea_t2 * ea_s23) / ea_e3
    ea = torch.sqrt(pi_o[..., 0] - ea_t1**2 - ea_t2**2 - ea_t3**2)
    return torch.stack((torch.log(ea), torch.log(ea_e1 / ea), torch.log(ea_e2 / ea), torch.log(ea_e3 / ea),
                        ea_s12 / ea, ea_s23 / ea, ea_s13 / ea, ea_t1 / ea, ea_t2 / ea, ea_t3 / ea), -1)


def pi_cm_to_theta(pi_cm: Tensor) -> Tensor:
    return pi_o_to_theta(pi_cm_to_pi_o(pi_cm))


def theta_to_spatial_inertia(theta: Tensor) -> Tensor:
    """theta -> [m, p, I_cm / m]: what the reference hands to the drake_pytorch closures
    (multibody_terms.py:228-230 via inertia.py:363-366, 377-382)."""
    pi_cm = pi_o_to_pi_cm(theta_to_pi_o(theta))
    return torch.cat((pi_cm[..., 0:1], pi_cm[..., 1:] / pi_cm[..., 0:1]), -1)


def spatial_inertia_6x6(inertia: Tensor, inertia_mode: str) -> Tensor:
    """6x6 spatial inertia about the BODY ORIGIN from the reference's closure argument
    ``inertia = [m, p, Ivec]``.

    quirk Q1: the symbolic variables fed with ``Ivec`` were bound as a full central
    RotationalInertia (multibody_terms.py:190-201, MakeFromCentralInertia) while the value
    substituted is I_cm / m (inertia.py:377-382).  ``reference_literal`` therefore uses
    Ivec as the central rotational inertia; ``physical`` multiplies the mass back in."""
    mass = inertia[..., 0]
    com = inertia[..., 1:4]
    i_cm = _inertia_matrix(inertia[..., 4:])
    if inertia_mode == 'physical':
        i_cm = i_cm * mass[..., None, None]
    else:
        assert inertia_mode == 'reference_literal'
    s = skew(com)
    m3 = mass[..., None, None]
    i_o = i_cm - m3 * (s @ s)
    eye = torch.eye(3, dtype=inertia.dtype).expand(i_o.shape)
    top = torch.cat((i_o, m3 * s), -1)
    bottom = torch.cat((m3 * s.transpose(-1, -2), m3 * eye), -1)
    return torch.cat((top, bottom), -2)


# --------------------------------------------------------------------------------------
# chain kinematics in body coordinates (restates what Drake's symbolic plant provides,
# reference multibody_terms.py:123-146 and 355-376)
# --------------------------------------------------------------------------------------
def motion_cross(a: Tensor, b: Tensor) -> Tensor:
    """spatial motion cross product a x b, vectors [angular; linear]."""
    aw, av = a[..., :3], a[..., 3:]
    bw, bv = b[..., :3], b[..., 3:]
    return torch.cat((torch.cross(aw, bw, dim=-1), torch.cross(aw, bv, dim=-1) + torch.cross(av, bw, dim=-1)),
                     -1)


def force_cross(v: Tensor, f: Tensor) -> Tensor:
    """spatial force cross product v x* f, f = [moment; force]."""
    vw, vv = v[..., :3], v[..., 3:]
    fn, fl = f[..., :3], f[..., 3:]
    return torch.cat((torch.cross(vw, fn, dim=-1) + torch.cross(vv, fl, dim=-1), torch.cross(vw, fl, dim=-1)),
                     -1)


def chain_kinematics(spec: Dict, q: Tensor, v: Optional[Tensor] = None):
    """Per-body world rotation R_b, world origin o_b, body-frame spatial Jacobian S_b
    (6 x n_v, so V_b = S_b v), and -- if v is given -- V_b and the velocity-product
    (bias) spatial acceleration A_b (value of dV_b/dt at zero generalised acceleration)."""
    n_bodies = len(spec['bodies'])
    _, n_v = state_sizes(spec)
    batch = q.shape[:-1]
    dtype = q.dtype
    rot: List[Tensor] = []
    org: List[Tensor] = []
    jac: List[Tensor] = []
    vel: List[Tensor] = []
    acc: List[Tensor] = []
    for index, body in enumerate(spec['bodies']):
        # where the body's joint sits in q / v: one floating-base model (joint j drives body j + 1, breadth-first order), or
        # the places system_spec recorded for the models of a ProductSpace
        qi = body.get('q_index', 0 if body['parent'] < 0 else 7 + index - 1)
        vi = body.get('v_index', 0 if body['parent'] < 0 else 6 + index - 1)
        if body['parent'] < 0 and (body.get('fixed') or spec.get('fixed_base')):
            # welded to the world at its mount: no coordinates, no velocity
            xyz, mount = body['mount']
            r_b = torch.tensor(mount, dtype=dtype).expand(batch + (3, 3))
            o_b = torch.tensor(xyz, dtype=dtype).expand(batch + (3,))
            s_b = torch.zeros(batch + (6, n_v), dtype=dtype)
            if v is not None:
                v_b = torch.zeros(batch + (6,), dtype=dtype)
                a_b = torch.zeros(batch + (6,), dtype=dtype)
        elif body['parent'] < 0:
            r_b = quat_to_rot(q[..., qi:qi + 4])
            o_b = q[..., qi + 4:qi + 7]
            s_b = torch.zeros(batch + (6, n_v), dtype=dtype)
            s_b[..., 0, vi] = 1.
            s_b[..., 1, vi + 1] = 1.
            s_b[..., 2, vi + 2] = 1.
            s_b[..., 3:6, vi + 3:vi + 6] = r_b.transpose(-1, -2)
            if v is not None:
                v_b = (s_b @ v.unsqueeze(-1)).squeeze(-1)
                zero3 = torch.zeros(batch + (3,), dtype=dtype)
                a_b = torch.cat((zero3, -torch.cross(v_b[..., :3], v_b[..., 3:], dim=-1)), -1)
        else:
            parent = body['parent']
            axis = torch.tensor(body['joint_axis'], dtype=dtype)
            p_j = torch.tensor(body['joint_origin'], dtype=dtype)
            r_joint = torch.tensor(body['joint_rot'], dtype=dtype)
            if body['joint_kind'] == 'prismatic':
                # child frame = joint frame moved along the axis (given in the joint frame) by the joint coordinate
                r_pc = r_joint.expand(batch + (3, 3))
                p_j = p_j + q[..., qi].unsqueeze(-1) * (r_joint @ axis)
                s_col = torch.cat((torch.zeros(3, dtype=dtype), axis))
            else:
                # child frame = joint frame (the <origin> of the joint, rpy included) turned about the axis by the angle
                r_pc = r_joint @ axis_rotation(axis, q[..., qi])
                s_col = torch.cat((axis, torch.zeros(3, dtype=dtype)))
            e_cp = r_pc.transpose(-1, -2)
            r_b = rot[parent] @ r_pc
            o_b = org[parent] + (rot[parent] @ p_j.unsqueeze(-1)).squeeze(-1)
            # motion transform parent -> child coordinates at the child origin
            x_top = torch.cat((e_cp, torch.zeros_like(e_cp)), -1)
            x_bottom = torch.cat((-e_cp @ skew(p_j), e_cp), -1)
            x_cp = torch.cat((x_top, x_bottom), -2)
            s_b = x_cp @ jac[parent]
            s_b = s_b.clone()
            s_b[..., :, vi] = s_b[..., :, vi] + s_col
            if v is not None:
                v_b = (s_b @ v.unsqueeze(-1)).squeeze(-1)
                rate = v[..., vi].unsqueeze(-1)
                a_b = (x_cp @ acc[parent].unsqueeze(-1)).squeeze(-1) + motion_cross(v_b, s_col * rate)
        rot.append(r_b)
        org.append(o_b)
        jac.append(s_b)
        if v is not None:
            vel.append(v_b)
            acc.append(a_b)
    assert len(rot) == n_bodies
    return rot, org, jac, vel, acc


def inertia_rows(spec: Dict) -> List[Dict]:
    """The Drake bodies that carry inertia, in the order of the rows of the reference's inertial_parameters
    (multibody_terms.py:161-207): what parse_urdf / system_spec recorded for models with welded links, else the bodies."""
    if 'inertia_rows' in spec:
        return spec['inertia_rows']
    return [{'name': body['name'], 'body': index, 'mass': body['mass'], 'com': body['com'], 'inertia_cm': body['inertia_cm'],
             'origin': [0., 0., 0.], 'rot': _IDENTITY} for index, body in enumerate(spec['bodies'])]


def _carried_frame(row: Dict, dtype) -> Optional[Tensor]:
    """motion transform from the coordinates of the body that carries a welded link to the link's own (at the link's origin);
    None for the body's own link"""
    if row['rot'] == _IDENTITY and not any(row['origin']):
        return None
    e_cp = torch.tensor(row['rot'], dtype=dtype).t()
    p = torch.tensor(row['origin'], dtype=dtype)
    return torch.cat((torch.cat((e_cp, torch.zeros(3, 3, dtype=dtype)), -1), torch.cat((-e_cp @ skew(p), e_cp), -1)), -2)


def mass_matrix(spec: Dict, q: Tensor, inertia: Tensor, inertia_mode: str) -> Tensor:
    """M(q) = sum_b S_b^T I_b S_b; equals gamma^T M_drake gamma of reference
    multibody_terms.py:131 (same kinetic energy, dair_pll velocity coordinates).
    ``inertia``: (*, n_bodies, 10) as passed by LagrangianTerms.forward (:228-234)."""
    _, _, jac, _, _ = chain_kinematics(spec, q)
    total = None
    for index, row in enumerate(inertia_rows(spec)):  # (a welded link: a Drake body without a coordinate, riding on its host)
        i6 = spatial_inertia_6x6(inertia[..., index, :], inertia_mode)
        x_lb = _carried_frame(row, q.dtype)
        s_l = jac[row['body']] if x_lb is None else x_lb @ jac[row['body']]
        term = s_l.transpose(-1, -2) @ i6 @ s_l
        total = term if total is None else total + term
    return total


def lagrangian_forces(spec: Dict, q: Tensor, v: Tensor, inertia: Tensor, inertia_mode: str, u: Optional[Tensor] = None) -> Tensor:
    """F(q, v, u) = gamma^T(-C + B u + tau_g) of reference multibody_terms.py:142-146:
    F = -sum_b S_b^T (I_b (A_b - G_b) + V_b x* I_b V_b), G_b = gravity as a spatial accel; + B u: input k of `u` on the coordinate
    of joint spec['actuators'][k] (gamma is the identity on joint coordinates).  `u` of width 0 (what every caller of the
    reference passes for its unactuated systems, and sim_step always) adds nothing."""
    rot, _, jac, vel, acc = chain_kinematics(spec, q, v)
    g_world = torch.tensor([0., 0., GRAVITY_Z], dtype=q.dtype)
    total = None
    for index, row in enumerate(inertia_rows(spec)):
        i6 = spatial_inertia_6x6(inertia[..., index, :], inertia_mode)
        b = row['body']
        x_lb = _carried_frame(row, q.dtype)
        if x_lb is None:
            r_l, s_l, v_l, a_l = rot[b], jac[b], vel[b], acc[b]
        else:  # a welded link: a joint without a coordinate -- its motion is its host's, seen from its own frame
            r_l = rot[b] @ torch.tensor(row['rot'], dtype=q.dtype)
            s_l = x_lb @ jac[b]
            v_l = (x_lb @ vel[b].unsqueeze(-1)).squeeze(-1)
            a_l = (x_lb @ acc[b].unsqueeze(-1)).squeeze(-1)
        g_body = (r_l.transpose(-1, -2) @ g_world.unsqueeze(-1)).squeeze(-1)
        grav = torch.cat((torch.zeros_like(g_body), g_body), -1)
        momentum = (i6 @ v_l.unsqueeze(-1)).squeeze(-1)
        wrench = (i6 @ (a_l - grav).unsqueeze(-1)).squeeze(-1) + force_cross(v_l, momentum)
        term = -(s_l.transpose(-1, -2) @ wrench.unsqueeze(-1)).squeeze(-1)
        total = term if total is None else total + term
    if u is not None and u.shape[-1] > 0:
        actuators = spec.get('actuators', [])
        assert u.shape[-1] == len(actuators), 'u must have one column per <transmission> of the model'
        columns = torch.zeros(len(actuators), total.shape[-1], dtype=q.dtype)
        for k, joint in enumerate(actuators):
            columns[k, spec['bodies'][joint + 1].get('v_index', 6 + joint)] = 1.0
        total = total + u @ columns
    return total


def geometry_kinematics(spec: Dict, q: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """The three closures of reference multibody_terms.py:299-310 evaluated at q:
    R_WG (*, n_g, 3, 3), p_WoGo_W (*, n_g, 3) and the world-frame spatial Jacobian [w; v]
    of each geometry origin in dair_pll velocity coordinates (*, n_g, 6, n_v) (:355-376)."""
    rot, org, jac, _, _ = chain_kinematics(spec, q)
    batch = q.shape[:-1]
    _, n_v = state_sizes(spec)
    rots, trans, jacs = [], [], []
    for geom in geometry_table(spec):
        if geom['body'] < 0:
            rots.append(torch.eye(3, dtype=q.dtype).expand(batch + (3, 3)))
            trans.append(torch.zeros(batch + (3,), dtype=q.dtype))
            jacs.append(torch.zeros(batch + (6, n_v), dtype=q.dtype))
            continue
        b = geom['body']
        c = torch.tensor(geom['origin'], dtype=q.dtype)
        r_b = rot[b]
        rots.append(r_b @ torch.tensor(geom['rot'], dtype=q.dtype))
        trans.append(org[b] + (r_b @ c.unsqueeze(-1)).squeeze(-1))
        ang = jac[b][..., :3, :]
        lin = jac[b][..., 3:, :] - skew(c) @ ang
        jacs.append(torch.cat((r_b @ ang, r_b @ lin), -2))
    return torch.stack(rots, -3), torch.stack(trans, -2), torch.stack(jacs, -3)


# --------------------------------------------------------------------------------------
# collision geometry (reference geometry.py, deep_support_function.py)
# --------------------------------------------------------------------------------------
_UNIT_BOX = torch.tensor([[-1., -1., -1.], [-1., -1., 1.], [-1., 1., -1.], [-1., 1., 1.], [1., -1., -1.],
                          [1., -1., 1.], [1., 1., -1.], [1., 1., 1.]])  # reference geometry.py:39-41


def box_vertices(length_params: Tensor) -> Tensor:
    """reference geometry.py:393-403: unit corners times |length_params|."""
    return _UNIT_BOX.to(length_params.dtype) * torch.abs(length_params).reshape(1, 3)


def topk_support(directions: Tensor, vertices: Tensor, n_query: int = N_QUERY) -> Tensor:
    """reference geometry.py:162-202: the n_query vertices with the largest d . s.  The
    reference's torch.topk(sorted=False) leaves the order unspecified (quirk Q3); here it is
    descending (ties: lower vertex index first).  vertices (*, N, 3) or (N, 3)."""
    if vertices.dim() == 2:
        vertices = vertices.expand(directions.shape[:-1] + vertices.shape)
    dots = (directions.unsqueeze(-2) * vertices).sum(-1)
    order = torch.sort(dots, dim=-1, descending=True, stable=True).indices[..., :n_query]
    return torch.gather(vertices, -2, order.unsqueeze(-1).expand(order.shape + (3,)))


def surface_directions() -> Tensor:
    """reference deep_support_function.py:12-16: unit directions through the boundary nodes of an 8 x 8 x 8 grid on [-1, 1]^3"""
    line = torch.linspace(-1, 1, steps=8, dtype=torch.float64)
    grid = torch.cartesian_prod(line, line, line)
    surface = grid[grid.abs().max(dim=-1).values >= 1.0]
    return surface / surface.norm(dim=-1, keepdim=True)


def rotation_matrix_from_one_vector(directions: Tensor, axis: int = 2) -> Tensor:
    """reference tensor_utils.py:305-366 (after Drake's MakeFromOneVector): R with R[:, axis] = d."""
    a = directions / directions.norm(dim=-1, keepdim=True)
    flat = a.reshape(-1, 3)
    rows = torch.arange(flat.shape[0])
    i = torch.abs(flat).min(dim=-1).indices
    j, k = (i + 1) % 3, (i + 2) % 3
    a_i, a_j, a_k = flat[rows, i], flat[rows, j], flat[rows, k]
    mag = torch.sqrt(1 - a_i * a_i)
    corr = -a_i / mag
    col_b = torch.zeros_like(flat)
    col_b[rows, j] = -a_k / mag
    col_b[rows, k] = a_j / mag
    col_c = torch.zeros_like(flat)
    col_c[rows, i] = mag
    col_c[rows, j] = corr * a_j
    col_c[rows, k] = corr * a_k
    columns = [None, None, None]
    columns[axis], columns[(axis + 1) % 3], columns[(axis + 2) % 3] = flat, col_b, col_c
    return torch.stack(columns, -1).reshape(directions.shape + (3,))


def _closest_on_triangles(tri):
    """closest point to the origin on each triangle (T, 3, 3) (Ericson, Real-Time Collision Detection 5.1.5)"""
    import numpy as np
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    ab, ac, ap = b - a, c - a, -a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = -b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = -c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    va, vb, vc = d3 * d6 - d5 * d4, d5 * d2 - d1 * d6, d1 * d4 - d3 * d2
    out = np.zeros_like(a)
    done = np.zeros(len(a), dtype=bool)

    def put(mask, value):
        nonlocal done
        mask = mask & ~done
        out[mask] = value[mask]
        done |= mask
    with np.errstate(divide='ignore', invalid='ignore'):
        put((d1 <= 0) & (d2 <= 0), a)
        put((d3 >= 0) & (d4 <= d3), b)
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab)
        put((d6 >= 0) & (d5 <= d6), c)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac)
        put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[:, None] * (c - b))
        denom = 1.0 / (va + vb + vc)
        put(np.ones(len(a), dtype=bool), a + ab * (vb * denom)[:, None] + ac * (vc * denom)[:, None])
    return out


def pair_direction_exact(verts_a, verts_b):
    """fcl's role in collide_mesh_mesh (reference geometry.py:603-625), exactly: the unit direction from A to B that
    maximises the separation  min_b d.b - max_a d.a  of two convex vertex sets given in one frame -- the direction
    between the nearest points when they are apart (what fcl.distance's nearest points give, :621-625), the direction
    of minimum penetration when they overlap (the canonical stand-in for fcl.collide's first contact normal, :615-618).
    Computed on the Minkowski difference B - A: its convex hull (qhull) either contains the origin (closest facet) or
    not (closest point of its triangles).  numpy (n_a, 3), (n_b, 3) -> (3,)."""
    import numpy as np
    from scipy.spatial import ConvexHull, QhullError
    diff = (verts_b[None, :, :] - verts_a[:, None, :]).reshape(-1, 3)
    if diff.shape[0] == 1:
        return diff[0] / np.linalg.norm(diff[0])
    centred = diff - diff.mean(0)
    rank = np.linalg.matrix_rank(centred, tol=1e-12 * max(1.0, np.abs(diff).max()))
    if rank < 3:
        raise NotImplementedError('flat Minkowski difference')
    hull = ConvexHull(diff)
    normals, offsets = hull.equations[:, :3], hull.equations[:, 3]
    if (offsets <= 0).all():  # origin inside: facet plane closest to the origin, direction against its outward normal
        return -normals[np.argmax(offsets)]
    closest = _closest_on_triangles(diff[hull.simplices])
    best = closest[np.argmin((closest ** 2).sum(-1))]
    return best / np.linalg.norm(best)


def icnn_support_point(weights: Dict[str, Tensor], directions: Tensor, negative_slope: float = 0.5) -> Tensor:
    """HomogeneousICNN.forward (reference deep_support_function.py:238-266): returns
    d f / d direction of the depth-D network of :213-236 with |W_h|, |w_out| (:189-194) and
    LeakyReLU masks treated as constants (:196-211).  ``weights`` keys: 'input_weights.i'
    (3, W), 'hidden_weights.i' (W, W), 'output_weight' (W,)."""
    depth = sum(1 for key in weights if key.startswith('input_weights.'))
    acts = []
    pre = directions @ weights['input_weights.0']
    acts.append(torch.where(pre > 0, pre, negative_slope * pre))
    for layer in range(1, depth):
        pre = acts[-1] @ torch.abs(weights[f'hidden_weights.{layer - 1}']) + directions @ weights[
            f'input_weights.{layer}']
        acts.append(torch.where(pre > 0, pre, negative_slope * pre))

    def mask(act: Tensor) -> Tensor:
        return torch.where(act <= 0, torch.full_like(act, negative_slope), torch.ones_like(act)).detach()

    hidden_jac = torch.abs(weights['output_weight']) * mask(acts[-1])
    result = torch.zeros_like(directions)
    for layer in range(depth - 1, 0, -1):
        result = result + hidden_jac @ weights[f'input_weights.{layer}'].transpose(-1, -2)
        hidden_jac = (hidden_jac @ torch.abs(weights[f'hidden_weights.{layer - 1}']).transpose(-1, -2)) * mask(
            acts[layer - 1])
    return result + hidden_jac @ weights['input_weights.0'].transpose(-1, -2)


def icnn_value(weights: Dict[str, Tensor], directions: Tensor, negative_slope: float = 0.5) -> Tensor:
    """network_activations output f(d) (reference deep_support_function.py:213-236)."""
    depth = sum(1 for key in weights if key.startswith('input_weights.'))
    pre = directions @ weights['input_weights.0']
    act = torch.where(pre > 0, pre, negative_slope * pre)
    for layer in range(1, depth):
        pre = act @ torch.abs(weights[f'hidden_weights.{layer - 1}']) + directions @ weights[
            f'input_weights.{layer}']
        act = torch.where(pre > 0, pre, negative_slope * pre)
    return act @ torch.abs(weights['output_weight'])


def mesh_support(weights: Dict[str, Tensor], perturbations: Tensor, directions: Tensor) -> Tensor:
    """DeepSupportConvex.get_vertices + support_points (reference geometry.py:309-325,
    162-202): support points of the normalised perturbed directions; all n_query of them
    are returned (top-4 of 4), ordered by descending d . s."""
    perturbed = directions.unsqueeze(-2) + perturbations
    perturbed = perturbed / perturbed.norm(dim=-1, keepdim=True)
    vertices = icnn_support_point(weights, perturbed)
    return topk_support(directions, vertices, perturbations.shape[0])


# --------------------------------------------------------------------------------------
# the QP solver: restatement of sappy.SAPSolver.apply(J, q, eps)   (PARITY UNPINNED)
# --------------------------------------------------------------------------------------
def lorentz_project(z: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Projection of (*, k, 3) vectors [t1, t2, n] onto {n >= |t|}; also returns the region
    masks (inside, polar) and |t|."""
    t = z[..., :2]
    n = z[..., 2]
    r = t.norm(dim=-1)
    inside = r <= n
    polar = (r <= -n) & ~inside
    safe_r = torch.where(r > 0, r, torch.ones_like(r))
    s = 0.5 * (n + r)
    mid = torch.cat((t * (s / safe_r).unsqueeze(-1), s.unsqueeze(-1)), -1)
    out = torch.where(inside.unsqueeze(-1), z, torch.where(polar.unsqueeze(-1), torch.zeros_like(z), mid))
    return out, inside, polar, r


def lorentz_project_jacobian(z: Tensor) -> Tensor:
    """Generalised Jacobian (*, k, 3, 3) of lorentz_project."""
    t = z[..., :2]
    n = z[..., 2]
    r = t.norm(dim=-1)
    inside = r <= n
    polar = (r <= -n) & ~inside
    safe_r = torch.where(r > 0, r, torch.ones_like(r))
    that = t / safe_r.unsqueeze(-1)
    ratio = (0.5 * (n + r) / safe_r)[..., None, None]
    eye2 = torch.eye(2, dtype=z.dtype)
    tt = that.unsqueeze(-1) * that.unsqueeze(-2)
    block = ratio * eye2 + (0.5 - ratio) * tt
    top = torch.cat((block, 0.5 * that.unsqueeze(-1)), -1)
    bottom = torch.cat((0.5 * that, 0.5 * torch.ones_like(n).unsqueeze(-1)), -1).unsqueeze(-2)
    mid = torch.cat((top, bottom), -2)
    eye3 = torch.eye(3, dtype=z.dtype).expand(mid.shape)
    return torch.where(inside[..., None, None], eye3,
                       torch.where(polar[..., None, None], torch.zeros_like(mid), mid))


def sap_solve(J: Tensor, q: Tensor, eps: float, tol: float = 1e-13, max_iter: int = 100,
              return_info: bool = False):
    """argmin over f in (Lorentz cone)^k of 1/2 f^T (J J^T + eps I) f + q^T f.

    J (*, 3k, n), q (*, 3k), per-contact order [t_x, t_y, n] (reference
    tensor_utils.py:460-497), unit friction cones (mu is folded into J, reference
    multibody_terms.py:424).  Strictly convex => unique minimiser, so any convergent method
    gives the reference's answer up to its own tolerance.  Method: Newton on the
    unconstrained primal  l(x) = 1/2 |x|^2 + eps/2 sum_c |P_K(-(J_c x + q_c)/eps)|^2
    (x in R^n, f_c = P_K(.)), derivative-based line search (root of l'(alpha) by
    safeguarded Newton).  Stops at |grad| <= tol * (1 + max(|x|, |J^T gamma|)) or when the
    gradient norm stops decreasing at rounding level."""
    batch = J.shape[:-2]
    m, n = J.shape[-2:]
    k = m // 3
    Jf = J.reshape((-1, m, n)).detach()
    qf = q.reshape((-1, m)).detach()
    nb = Jf.shape[0]
    x = torch.zeros((nb, n), dtype=J.dtype)
    iters = torch.zeros(nb, dtype=torch.long)
    active = torch.ones(nb, dtype=torch.bool)
    best = torch.full((nb,), float('inf'), dtype=J.dtype)
    stall = torch.zeros(nb, dtype=torch.long)
    eye = torch.eye(n, dtype=J.dtype)
    tiny = torch.finfo(J.dtype).eps

    for _ in range(max_iter):
        idx = torch.nonzero(active).squeeze(-1)
        if idx.numel() == 0:
            break
        Ja, qa, xa = Jf[idx], qf[idx], x[idx]
        z = (-((Ja @ xa.unsqueeze(-1)).squeeze(-1) + qa) / eps).reshape(-1, k, 3)
        gamma = lorentz_project(z)[0].reshape(-1, m)
        jt_gamma = (Ja.transpose(-1, -2) @ gamma.unsqueeze(-1)).squeeze(-1)
        grad = xa - jt_gamma
        gnorm = grad.norm(dim=-1)
        scale = 1.0 + torch.maximum(xa.norm(dim=-1), jt_gamma.norm(dim=-1))
        improved = gnorm < 0.5 * best[idx]
        stall[idx] = torch.where(improved, torch.zeros_like(stall[idx]), stall[idx] + 1)
        best[idx] = torch.minimum(best[idx], gnorm)
        done = (gnorm <= tol * scale) | ((stall[idx] >= 3) & (gnorm <= 1e3 * tol * scale))
        active[idx[done]] = False
        keep = ~done
        if not keep.any():
            break
        idx, Ja, xa, z, grad = idx[keep], Ja[keep], xa[keep], z[keep], grad[keep]
        dp = lorentz_project_jacobian(z)  # (b, k, 3, 3)
        J3 = Ja.reshape(-1, k, 3, n)
        hess = eye + (J3.transpose(-1, -2) @ dp @ J3).sum(-3) / eps
        d = -torch.linalg.solve(hess, grad.unsqueeze(-1)).squeeze(-1)
        jd = (Ja @ d.unsqueeze(-1)).squeeze(-1)  # (b, m)
        jd3 = jd.reshape(-1, k, 3)
        dz = -jd3 / eps
        xd = (xa * d).sum(-1)
        dd = (d * d).sum(-1)
        slope0 = (grad * d).sum(-1)  # l'(0) < 0

        def dphi(alpha: Tensor, sel: Tensor) -> Tuple[Tensor, Tensor]:
            za = z[sel] + alpha[:, None, None] * dz[sel]
            ga = lorentz_project(za)[0]
            first = xd[sel] + alpha * dd[sel] - (ga * jd3[sel]).sum((-1, -2))
            dpa = lorentz_project_jacobian(za)
            second = dd[sel] + ((dpa @ jd3[sel].unsqueeze(-1)).squeeze(-1) * jd3[sel]).sum((-1, -2)) / eps
            return first, second

        nbk = xd.shape[0]
        alpha = torch.ones(nbk, dtype=J.dtype)
        lo = torch.zeros(nbk, dtype=J.dtype)
        hi = torch.full((nbk,), float('inf'), dtype=J.dtype)
        searching = torch.ones(nbk, dtype=torch.bool)
        for _ in range(100):
            sel = torch.nonzero(searching).squeeze(-1)
            if sel.numel() == 0:
                break
            a_s = alpha[sel]
            first, second = dphi(a_s, sel)
            ok = first.abs() <= 1e-9 * slope0[sel].abs()
            lo_s = torch.where(first < 0, a_s, lo[sel])
            hi_s = torch.where(first >= 0, a_s, hi[sel])
            newton = a_s - first / second
            mid = torch.where(torch.isinf(hi_s), 2 * a_s, 0.5 * (lo_s + hi_s))
            bad = ~((newton > lo_s) & (newton < hi_s))
            nxt = torch.where(bad, mid, newton)
            ok = ok | ((hi_s - lo_s) <= 4 * tiny * hi_s)
            lo[sel], hi[sel] = lo_s, hi_s
            alpha[sel] = torch.where(ok, a_s, nxt)
            searching[sel[ok]] = False
        x[idx] = xa + alpha.unsqueeze(-1) * d
        iters[idx] += 1
    z = (-((Jf @ x.unsqueeze(-1)).squeeze(-1) + qf) / eps).reshape(-1, k, 3)
    force = lorentz_project(z)[0].reshape(batch + (m,))
    if return_info:
        return force, x.reshape(batch + (n,)), iters.reshape(batch)
    return force


def sap_solve_diff(J: Tensor, q: Tensor, eps: float) -> Tensor:
    """``sap_solve`` that is differentiable in ``J`` and ``q``: what ``sappy.SAPSolver.apply`` is to the
    reference's ``forward_dynamics`` (multibody_learnable_system.py:293-298, not detached there; sappy's own
    backward is third party and UNPINNED -- this is the canonical answer, the implicit-function derivative of the
    unique optimum).

    With the converged primal optimum x* (grad l(x*; J, q) = 0) and the generalised Hessian H of l at x*, both held
    constant, one more Newton step written with autograd-visible ``J, q``,

        x(J, q) = x* - H^-1 grad l(x*; J, q),        f = P_K(-(J x(J, q) + q) / eps),

    has the value of the converged solution and -- because the Newton map's derivative at a fixed point is
    -H^-1 d(grad l)/d(J, q) -- exactly the implicit-function derivative dx*/d(J, q).  The projection's derivative is
    autograd's through ``lorentz_project`` (its generalised Jacobian region by region)."""
    batch = J.shape[:-2]
    m, n = J.shape[-2:]
    k = m // 3
    with torch.no_grad():
        _, x_star, _ = sap_solve(J, q, eps, return_info=True)
        z_star = (-((J @ x_star.unsqueeze(-1)).squeeze(-1) + q) / eps).reshape(batch + (k, 3))
        dp = lorentz_project_jacobian(z_star)
        J3 = J.reshape(batch + (k, 3, n))
        hess = torch.eye(n, dtype=J.dtype) + (J3.transpose(-1, -2) @ dp @ J3).sum(-3) / eps

    def gamma(x: Tensor) -> Tensor:
        z = (-((J @ x.unsqueeze(-1)).squeeze(-1) + q) / eps).reshape(batch + (k, 3))
        return lorentz_project(z)[0].reshape(batch + (m,))

    grad_l = x_star - (J.transpose(-1, -2) @ gamma(x_star).unsqueeze(-1)).squeeze(-1)
    x = x_star - torch.linalg.solve(hess, grad_l.unsqueeze(-1)).squeeze(-1)
    return gamma(x)


def kkt_residuals(J: Tensor, q: Tensor, eps: float, f: Tensor) -> Dict[str, Tensor]:
    """KKT certificate of the dual QP: f in K, r = (J J^T + eps I) f + q in K* = K, f . r = 0."""
    m = J.shape[-2]
    k = m // 3
    r = (J @ (J.transpose(-1, -2) @ f.unsqueeze(-1))).squeeze(-1) + eps * f + q
    f3 = f.reshape(f.shape[:-1] + (k, 3))
    r3 = r.reshape(r.shape[:-1] + (k, 3))
    primal = (f3[..., :2].norm(dim=-1) - f3[..., 2]).clamp(min=0).amax(-1)
    dual = (r3[..., :2].norm(dim=-1) - r3[..., 2]).clamp(min=0).amax(-1)
    comp = (f * r).sum(-1).abs()
    return {'primal': primal, 'dual': dual, 'complementarity': comp}


def sappy_reorder_matrix(k: int, dtype=torch.float64) -> Tensor:
    """lambda = P lambda_s, reference tensor_utils.py:460-497."""
    mat = torch.zeros((3 * k, 3 * k), dtype=dtype)
    for c in range(k):
        mat[c, 3 * c + 2] = 1
        mat[k + 2 * c, 3 * c] = 1
        mat[k + 2 * c + 1, 3 * c + 1] = 1
    return mat


# --------------------------------------------------------------------------------------
# the system
# --------------------------------------------------------------------------------------
class OracleSystem:
    """CPU restatement of MultibodyLearnableSystem for one URDF (+ ground plane).

    Parameters mirror the reference ``state_dict``: ``theta`` (n_bodies, 10),
    ``friction`` (n_geometries,), per-geometry ``length_params`` (1, 3) for boxes or ICNN
    weights for meshes."""

    def __init__(self, urdf, dt: float, inertia_mode: str = 'reference_literal',
                 dtype=torch.float64, mesh_seed: int = 0, mesh_params: Optional[Dict] = None,
                 mesh_representation: str = 'deep_support'):
        """``urdf``: one path, or ``{name: path}`` -- the models of one system (reference ``init_urdfs``)"""
        self.spec = system_spec(urdf, mesh_representation)
        self.dt = dt
        self.dtype = dtype
        self.inertia_mode = inertia_mode
        self.geoms = geometry_table(self.spec)
        self.n_joints = self.spec['n_joints']
        self.n_q, self.n_v = state_sizes(self.spec)
        self.n_x = self.n_q + self.n_v
        # witness points per geometry: 4 (box: geometry.py:490; mesh: :47-48), 1 for a sphere (:440-452)
        self.ground_geoms = ground_geometries(self.spec)  # (an anchored geometry has no ground contacts)
        self.n_contacts = sum(1 if self.geoms[g]['kind'] == 'sphere' else N_QUERY for g in self.ground_geoms) + len(self.spec['pairs'])
        pi_cm = torch.tensor([[b['mass']] + [b['mass'] * c for c in b['com']] + b['inertia_cm']
                              for b in inertia_rows(self.spec)], dtype=torch.float64)
        # theta_0 = pi_o_to_theta(drake inertia), reference multibody_terms.py:186-188
        self.theta = pi_cm_to_theta(pi_cm).to(dtype)
        self.friction = torch.tensor([g['mu'] for g in self.geoms], dtype=dtype)
        self.geom_params: List[Optional[Dict[str, Tensor]]] = []
        for index, geom in enumerate(self.geoms):
            if geom['kind'] == 'plane':
                self.geom_params.append(None)
            elif geom['kind'] == 'box':
                self.geom_params.append({'length_params': torch.tensor([geom['half']], dtype=dtype)})
            elif geom['kind'] == 'sphere':
                self.geom_params.append({'length_param': torch.tensor(geom['radius'], dtype=dtype)})
            elif geom['kind'] == 'polygon':
                self.geom_params.append({'vertices': torch.tensor(geom['vertices'], dtype=dtype)})
            else:
                if mesh_params is not None and index in mesh_params:
                    self.geom_params.append({k: v.to(dtype) for k, v in mesh_params[index].items()})
                else:
                    self.geom_params.append(init_mesh_params(geom['vertices'], mesh_seed, dtype))

    # -- parameter plumbing ------------------------------------------------------------
    def named_parameters(self) -> Dict[str, Tensor]:
        """names follow the reference state_dict."""
        out = {'multibody_terms.lagrangian_terms.inertial_parameters': self.theta,
               'multibody_terms.contact_terms.friction_params': self.friction}
        for index, params in enumerate(self.geom_params):
            if params is None:
                continue
            for key, value in params.items():
                if key == 'perturbations':
                    continue
                prefix = f'multibody_terms.contact_terms.geometries.{index}.'
                out[prefix + (key if key in ('length_params', 'length_param', 'vertices') else 'network.' + key)] = value
        return out

    def requires_grad_(self, flag: bool = True) -> 'OracleSystem':
        for value in self.named_parameters().values():
            value.requires_grad_(flag)
        return self

    def zero_grad(self) -> None:
        for value in self.named_parameters().values():
            value.grad = None

    # -- state space (reference state_space.py:171-192) ----------------------------------
    def q_v(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        return x[..., :self.n_q], x[..., self.n_q:]

    # -- terms -----------------------------------------------------------------------
    def lagrangian_terms(self, q: Tensor, v: Tensor, u: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """LagrangianTerms.forward, reference multibody_terms.py:214-237."""
        inertia = theta_to_spatial_inertia(self.theta)
        inertia = inertia.expand(q.shape[:-1] + inertia.shape)
        M = mass_matrix(self.spec, q, inertia, self.inertia_mode)
        F = lagrangian_forces(self.spec, q, v, inertia, self.inertia_mode, u)
        return M, torch.linalg.solve(M, F)

    def support_points(self, geom_index: int, directions: Tensor) -> Tensor:
        params = self.geom_params[geom_index]
        if 'length_params' in params:
            return topk_support(directions, box_vertices(params['length_params']))
        if 'length_param' in params:  # Sphere.support_points, reference geometry.py:440-452: ONE witness point
            return (directions * torch.abs(params['length_param'])).unsqueeze(-2)
        if 'vertices' in params:  # Polygon.get_vertices (:241-243): the static vertex set, signed parameters
            return topk_support(directions, params['vertices'])
        weights = {k: v for k, v in params.items() if k != 'perturbations'}
        return mesh_support(weights, params['perturbations'], directions)

    def vertex_set(self, geom_index: int) -> Tuple[Tensor, Tensor]:
        """(vertices (N, 3), margin) of a geometry for the direction search: a sphere is its centre plus the radius"""
        params = self.geom_params[geom_index]
        if 'length_params' in params:
            return box_vertices(params['length_params']), torch.zeros((), dtype=self.dtype)
        if 'length_param' in params:
            return torch.zeros((1, 3), dtype=self.dtype), torch.abs(params['length_param'])
        if 'vertices' in params:
            return params['vertices'], torch.zeros((), dtype=self.dtype)
        # DeepSupportConvex.get_fcl_geometry (reference geometry.py:343-358): the mesh extract_mesh builds from the network,
        # i.e. its distinct support points over the 296 surface directions (deep_support_function.py:12-16, 92-113)
        weights = {k: v.detach() for k, v in params.items() if k != 'perturbations'}
        points = icnn_support_point(weights, surface_directions().to(self.dtype))
        seen, unique = set(), []
        for point in points:
            key = point.numpy().tobytes()
            if key not in seen:
                seen.add(key)
                unique.append(point)
        return torch.stack(unique), torch.zeros((), dtype=self.dtype)

    def support_single(self, geom_index: int, directions: Tensor) -> Tensor:
        """what `geometry.network(directions)` is to collide_mesh_mesh (reference geometry.py:627-629): ONE support point
        per direction -- the vertex furthest along it (ties: lowest index) plus, for a sphere, the radius along it"""
        params = self.geom_params[geom_index]
        if 'output_weight' in params:  # DeepSupportConvex: the network itself, geometry.network(directions) (:627-629)
            return icnn_support_point({k: v for k, v in params.items() if k != 'perturbations'}, directions)
        vertices, margin = self.vertex_set(geom_index)
        dots = (directions @ vertices.transpose(-1, -2)).detach()
        # the lowest index among the vertices within PAIR_TIE of the furthest one: a direction that is a face or edge
        # normal of the shape itself ties several vertices up to rounding (a support-function network has no such ties)
        best = torch.zeros(dots.shape[:-1], dtype=torch.long)
        value = dots[..., 0].clone()
        for u in range(1, dots.shape[-1]):
            better = dots[..., u] > value + PAIR_TIE
            best = torch.where(better, torch.full_like(best, u), best)
            value = torch.where(better, dots[..., u], value)
        return vertices[best] + margin * directions

    def collide_pair(self, a_index: int, b_index: int, R_AB: Tensor, p_AoBo_A: Tensor):
        """GeometryCollider.collide_mesh_mesh, reference geometry.py:585-643, with fcl's direction replaced by
        pair_direction_exact and `network(d)` by the geometry's own support function."""
        import numpy as np
        batch = p_AoBo_A.shape[:-1]
        R = R_AB.reshape(-1, 3, 3)
        p = p_AoBo_A.reshape(-1, 3)
        va, _ = self.vertex_set(a_index)
        vb, _ = self.vertex_set(b_index)
        directions = torch.zeros_like(p)
        with torch.no_grad():  # "collision directions are piecewise constant" (:597-600)
            for n in range(p.shape[0]):
                vb_in_a = vb.detach() @ R[n].detach().t() + p[n].detach()
                directions[n] = torch.tensor(pair_direction_exact(va.detach().numpy().astype(np.float64),
                                                                  vb_in_a.numpy().astype(np.float64)), dtype=p.dtype)
        directions = directions / directions.norm(dim=-1, keepdim=True)
        R_AC = rotation_matrix_from_one_vector(directions, 2)
        p_AoAc_A = self.support_single(a_index, directions)
        p_BoBc_B = self.support_single(b_index, -(directions.unsqueeze(-2) @ R).squeeze(-2))
        p_BoBc_A = (p_BoBc_B.unsqueeze(-2) @ R.transpose(-1, -2)).squeeze(-2)
        p_AcBc_A = -p_AoAc_A + p + p_BoBc_A
        phi = (p_AcBc_A * R_AC[..., 2]).sum(-1)
        return (phi.reshape(batch + (1,)), R_AC.reshape(batch + (1, 3, 3)), p_AoAc_A.reshape(batch + (1, 3)),
                p_BoBc_B.reshape(batch + (1, 3)))

    def contact_terms(self, q: Tensor) -> Tuple[Tensor, Tensor]:
        """ContactTerms.forward, reference multibody_terms.py:428-521: the plane-vs-convex pairs (geometry.py:554-582)
        first, then the body-body candidates (geometry.py:585-643)."""
        R_WC, p_WoCo_W, Jv_V_WC_W = geometry_kinematics(self.spec, q)
        mu_all = torch.abs(self.friction)  # :321-324
        phis, jacs, mus = [], [], []
        for b_index in self.ground_geoms:
            a_index = 0
            mu = 2 * mu_all[a_index] * mu_all[b_index] / (mu_all[a_index] + mu_all[b_index])  # :471
            R_WA = R_WC[..., a_index, :, :]
            R_WB = R_WC[..., b_index, :, :]
            R_AW = R_WA.transpose(-1, -2)
            R_AB = R_AW @ R_WB
            p_AoBo_A = (R_AW @ (p_WoCo_W[..., b_index, :] - p_WoCo_W[..., a_index, :]).unsqueeze(-1)).squeeze(-1)
            # collide_plane_convex (geometry.py:554-582)
            directions_b = -R_AB[..., 2, :]
            p_BoBc_B = self.support_points(b_index, directions_b)  # (*, 4, 3)
            p_AoBc_A = p_BoBc_B @ R_AB.transpose(-1, -2) + p_AoBo_A.unsqueeze(-2)
            phi_i = p_AoBc_A[..., 2]
            p_AoAc_A = torch.cat((p_AoBc_A[..., :2], torch.zeros_like(p_AoBc_A[..., 2:])), -1)
            R_FW = R_AW.unsqueeze(-3)  # R_AC = I  (:581)
            # assemble_velocity_jacobian (:385-399), tensor_utils.py:257-302
            p_AoAc_W = p_AoAc_A @ R_AW
            p_BoBc_W = p_BoBc_B @ R_WB.transpose(-1, -2)
            eye = torch.eye(3, dtype=q.dtype).expand(p_BoBc_W.shape + (3,))
            J_A = torch.cat((-skew(p_AoAc_W), eye), -1) @ Jv_V_WC_W[..., a_index, :, :].unsqueeze(-3)
            J_B = torch.cat((-skew(p_BoBc_W), eye), -1) @ Jv_V_WC_W[..., b_index, :, :].unsqueeze(-3)
            jacs.append(R_FW @ (J_B - J_A))  # (*, 4, 3, n_v)
            phis.append(phi_i)
            mus.append(mu.repeat(phi_i.shape[-1]))
        for a_index, b_index in self.spec['pairs']:
            mu = 2 * mu_all[a_index] * mu_all[b_index] / (mu_all[a_index] + mu_all[b_index])
            R_WA = R_WC[..., a_index, :, :]
            R_WB = R_WC[..., b_index, :, :]
            R_AW = R_WA.transpose(-1, -2)
            R_AB = R_AW @ R_WB
            p_AoBo_A = (R_AW @ (p_WoCo_W[..., b_index, :] - p_WoCo_W[..., a_index, :]).unsqueeze(-1)).squeeze(-1)
            phi_i, R_AF, p_AoAc_A, p_BoBc_B = self.collide_pair(a_index, b_index, R_AB, p_AoBo_A)
            R_FW = R_AF.transpose(-1, -2) @ R_AW.unsqueeze(-3)  # :494-495
            p_AoAc_W = p_AoAc_A @ R_AW
            p_BoBc_W = p_BoBc_B @ R_WB.transpose(-1, -2)
            eye = torch.eye(3, dtype=q.dtype).expand(p_BoBc_W.shape + (3,))
            J_A = torch.cat((-skew(p_AoAc_W), eye), -1) @ Jv_V_WC_W[..., a_index, :, :].unsqueeze(-3)
            J_B = torch.cat((-skew(p_BoBc_W), eye), -1) @ Jv_V_WC_W[..., b_index, :, :].unsqueeze(-3)
            jacs.append(R_FW @ (J_B - J_A))
            phis.append(phi_i)
            mus.append(mu.repeat(1))
        phi = torch.cat(phis, -1)
        Jc = torch.cat(jacs, -3)  # (*, k, 3, n_v)
        mu_rep = torch.cat(mus)
        # relative_velocity_to_contact_jacobian (:402-426)
        J_n = Jc[..., 2, :]
        J_t = (mu_rep.reshape(-1, 1, 1) * Jc[..., :2, :]).reshape(Jc.shape[:-3] + (-1, Jc.shape[-1]))
        return phi, torch.cat((J_n, J_t), -2)

    def multibody_terms(self, q: Tensor, v: Tensor, u: Optional[Tensor] = None):
        """MultibodyTerms.forward, reference multibody_terms.py:584-609."""
        M, a = self.lagrangian_terms(q, v, u)
        phi, J = self.contact_terms(q)
        D = J @ torch.linalg.solve(M, J.transpose(-1, -2))
        return D, M, J, phi, a

    # -- the loss --------------------------------------------------------------------
    def contactnets_loss(self, x: Tensor, x_plus: Tensor, return_force: bool = False, u: Optional[Tensor] = None):
        """reference multibody_learnable_system.py:104-197."""
        _, v = self.q_v(x)
        q_plus, v_plus = self.q_v(x_plus)
        dt, eps = self.dt, LOSS_EPS
        D, M, J, phi, a = self.multibody_terms(q_plus, v_plus, u)
        k = phi.shape[-1]
        P = sappy_reorder_matrix(k, x.dtype)
        J_t = J[..., k:, :]
        phi_then_zero = torch.cat((phi, torch.zeros(phi.shape[:-1] + (2 * k,), dtype=x.dtype)), -1)
        sliding_velocities = (J_t @ v_plus.unsqueeze(-1)).squeeze(-1)
        sliding_speeds = sliding_velocities.reshape(phi.shape[:-1] + (k, 2)).norm(dim=-1)
        Q = D + eps * torch.eye(3 * k, dtype=x.dtype)
        J_M = P.t() @ (J @ torch.linalg.cholesky(torch.inverse(M)))
        dv = v_plus - (v + a * dt)
        q_pred = -(J @ dv.unsqueeze(-1)).squeeze(-1)
        q_comp = torch.abs(phi_then_zero)
        q_diss = dt * torch.cat((sliding_speeds, sliding_velocities), -1)
        qv = q_pred + q_comp + q_diss
        penalty = (torch.clamp(-phi, min=0)**2).sum(-1)
        constant = 0.5 * (dv.unsqueeze(-2) @ M @ dv.unsqueeze(-1)).reshape(dv.shape[:-1]) + penalty
        force_s = sap_solve(J_M, (qv.unsqueeze(-2) @ P).squeeze(-2), eps)
        force = (force_s.unsqueeze(-2) @ P.t()).squeeze(-2).detach()
        invalid = ((force.abs() > INVALID_FORCE) | force.isnan() | force.isinf()).any(-1)
        constant = torch.where(invalid, torch.zeros_like(constant), constant)
        force = torch.where(invalid.unsqueeze(-1), torch.zeros_like(force), force)
        loss = 0.5 * (force.unsqueeze(-2) @ Q @ force.unsqueeze(-1)).reshape(constant.shape) + \
            (force * qv).sum(-1) + constant
        if return_force:
            return loss, force
        return loss

    # -- dynamics ----------------------------------------------------------------------
    def forward_dynamics(self, q: Tensor, v: Tensor, return_impulse: bool = False, u: Optional[Tensor] = None):
        """reference multibody_learnable_system.py:199-304 (the eps=1e6 contact filter at
        :262-269 keeps every contact; Q, q at :288-291 are dead code)."""
        dt = self.dt
        D, M, J, phi, a = self.multibody_terms(q, v, u)
        k = phi.shape[-1]
        P = sappy_reorder_matrix(k, q.dtype)
        J_M = P.t() @ (J @ torch.linalg.cholesky(torch.inverse(M)))
        phi_then_zero = torch.cat((phi, torch.zeros(phi.shape[:-1] + (2 * k,), dtype=q.dtype)), -1)
        v_minus = v + dt * a
        q_full = (J @ v_minus.unsqueeze(-1)).squeeze(-1) + phi_then_zero / dt
        q_s = (q_full.unsqueeze(-2) @ P).squeeze(-2)
        # not detached in the reference (:293-298): with a graph to build, the solve is differentiated implicitly
        solve = sap_solve_diff if (torch.is_grad_enabled() and (J_M.requires_grad or q_s.requires_grad)) else sap_solve
        impulse_s = solve(J_M, q_s, DYNAMICS_EPS)
        impulse = (impulse_s.unsqueeze(-2) @ P.t()).squeeze(-2)
        v_plus = v_minus + torch.linalg.solve(M, (J.transpose(-1, -2) @ impulse.unsqueeze(-1))).squeeze(-1)
        if return_impulse:
            return v_plus, impulse
        return v_plus

    def step(self, x: Tensor) -> Tensor:
        """VelocityIntegrator.step (reference integrator.py:153-162) with
        FloatingBaseSpace.exponential (state_space.py:466-486); no quaternion
        re-normalisation."""
        q, v = self.q_v(x)
        v_next = self.forward_dynamics(q, v)
        dq = v_next * self.dt
        # ProductSpace.exponential (state_space.py:709-719): every floating base by the quaternion exponential, everything
        # else (positions, joint coordinates) additively
        q_next = q.clone()
        for index, body in enumerate(self.spec['bodies']):
            qi = body.get('q_index', 0 if body['parent'] < 0 else 7 + index - 1)
            vi = body.get('v_index', 0 if body['parent'] < 0 else 6 + index - 1)
            if body['parent'] < 0 and (body.get('fixed') or self.spec.get('fixed_base')):
                continue
            if body['parent'] < 0:
                q_next[..., qi:qi + 4] = quat_multiply(q[..., qi:qi + 4], quat_exp(dq[..., vi:vi + 3]))
                q_next[..., qi + 4:qi + 7] = q[..., qi + 4:qi + 7] + dq[..., vi + 3:vi + 6]
            else:
                q_next[..., qi] = q[..., qi] + dq[..., vi]
        return torch.cat((q_next, v_next), -1)

    def simulate(self, x_0: Tensor, steps: int) -> Tensor:
        """Integrator.simulate (reference integrator.py:75-99): (*, n_x) -> (*, steps+1, n_x)."""
        traj = [x_0]
        x = x_0
        for _ in range(steps):
            x = self.step(x)
            traj.append(x)
        return torch.stack(traj, -2)


def init_mesh_params(vertices: List[List[float]], seed: int, dtype=torch.float64, depth: int = 2,
                     width: int = 256, negative_slope: float = 0.5, perturbation: float = 0.4) -> Dict[str, Tensor]:
    """Random initial DeepSupportConvex parameters with the distribution of the reference
    constructors (geometry.py:303-307, deep_support_function.py:147-183).  The draw order is
    the oracle's own (seeded generator); golden fixtures carry the actual values."""
    gen = torch.Generator().manual_seed(seed)
    verts = torch.tensor(vertices, dtype=torch.float64)
    scale = float((verts.max(0).values - verts.min(0).values).norm() / 2)
    out: Dict[str, Tensor] = {}
    scale_hidden = 2 * (2.0 / (1 + negative_slope**2))**0.5 / width
    for layer in range(depth - 1):
        out[f'hidden_weights.{layer}'] = (2 * (torch.rand((width, width), generator=gen, dtype=torch.float64) - 0.5)
                                          * scale_hidden)
    for layer in range(depth):
        # kaiming_uniform_ on a (3, width) tensor: fan_in = width, bound = sqrt(6 / fan_in)
        bound = math.sqrt(6.0 / width)
        weight = (2 * torch.rand((3, width), generator=gen, dtype=torch.float64) - 1) * bound
        if layer > 0:
            weight = weight * 2**(-0.5)
        out[f'input_weights.{layer}'] = weight
    scale_out = scale * 2 * (2.0 / (width * (1 + negative_slope**2)))**0.5
    out['output_weight'] = 2 * (torch.rand(width, generator=gen, dtype=torch.float64) - 0.5) * scale_out
    out['perturbations'] = torch.cat((torch.zeros((1, 3), dtype=torch.float64),
                                      perturbation * (torch.rand((N_QUERY - 1, 3), generator=gen,
                                                                 dtype=torch.float64) - 0.5)))
    return {k: v.to(dtype) for k, v in out.items()}
