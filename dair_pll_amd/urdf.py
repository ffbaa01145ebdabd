"""URDF -> :class:`ModelSpec` for the contact-dynamics kernels (host logic, no Drake).

The reference obtains the same facts from Drake's parser and symbolic plant
(``dair_pll/drake_utils.py:248-335``, ``dair_pll/multibody_terms.py:161-207, 355-376``): per body
mass / centre of mass / central inertia, the joint tree, collision geometry with
``drake:mu_static``, and a ground half-space with friction 1.0 added to every plant
(``drake_utils.py:280-288``).  The kernels support one floating-base tree of up to three revolute or prismatic
joints with up to three box / sphere / polygon / learned-mesh collision geometries on any of its bodies, touching the
ground and -- up to four candidate pairs -- each other (the elbow's links are collision filtered,
``assets/contactnets_elbow.urdf``).  The cube and elbow systems of the reference's ContactNets example
-- a serial chain with exactly one box per body -- run on builds specialised for them
(:meth:`ModelSpec.is_fast`).
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

GROUND_MU = 1.0  # dair_pll/drake_utils.py:280-288
GRAVITY_Z = -9.81  # Drake's default UniformGravityField
MAX_JOINTS = 3  # dpll_core.hpp kMaxJoints
MAX_GEOMS = 3  # dpll_core.hpp kMaxGeoms
MAX_POLYGON_VERTICES = 8  # dpll_core.hpp kMaxPolyVerts
MAX_PAIRS = 4  # dpll_core.hpp kMaxPairs
# GeometryCollider orders a pair by type (geometry.py:46, 66-74; multibody_terms.py:294-297)
TYPE_ORDER = {'polygon': 1, 'box': 2, 'sphere': 3, 'mesh': 4}


@dataclass
class GeomSpec:
    kind: str  # 'box' | 'sphere' | 'mesh' (DeepSupportConvex) | 'polygon' (the mesh's vertex set, geometry.py:220-252)
    origin: List[float]
    mu: float
    half_lengths: Optional[List[float]] = None
    radius: Optional[float] = None
    mesh_file: Optional[str] = None
    vertices: Optional[List[List[float]]] = None
    rotation: List[List[float]] = field(default_factory=lambda: _identity())  # R_BG: the geometry frame in its body's
    link: Optional[str] = None  # the URDF link the <collision> element belongs to (its body's name unless the link is welded on)


@dataclass
class BodySpec:
    name: str
    mass: float
    com: List[float]
    inertia_cm: List[float]  # ixx iyy izz ixy ixz iyz about the centre of mass
    parent: int = -1
    joint_origin: Optional[List[float]] = None
    joint_axis: Optional[List[float]] = None
    # the joint frame (= this body's frame at joint angle 0) in the parent's frame: the rpy of the joint's <origin>
    joint_rotation: List[List[float]] = field(default_factory=lambda: _identity())
    joint_kind: str = 'revolute'  # or 'prismatic': this body slides along the axis (given in the joint frame)
    geoms: List[GeomSpec] = field(default_factory=list)


@dataclass
class InertiaRow:
    """One row of the reference's ``inertial_parameters`` -- one Drake body (``multibody_terms.py:161-207``): a URDF link ALONE,
    its mass, centre of mass and central inertia in its own frame, the kernel body that carries it and the link's frame in that
    body's (the identity for the body's own link; a link a ``fixed`` joint welds on sits at the joint's ``<origin>``)."""
    name: str
    body: int
    mass: float
    com: List[float]
    inertia_cm: List[float]
    origin: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    rotation: List[List[float]] = field(default_factory=lambda: _identity())


@dataclass
class ModelSpec:
    name: str
    bodies: List[BodySpec]
    ground_mu: float = GROUND_MU
    gravity_z: float = GRAVITY_Z
    # body-body collision candidates as (geometry a, geometry b), indices into :meth:`geoms`: geometries of two bodies that
    # no joint connects and no collision filter group excludes (what Drake's GetCollisionCandidates keeps beyond the
    # ground pairs, drake_utils.py:178-184), the pair ordered by geometry type as the reference orders it
    pairs: List[Tuple[int, int]] = field(default_factory=list)
    # links folded into another by a `fixed` joint: name -> name of the link that stands for it (collision filter groups
    # may still name them)
    welded: dict = field(default_factory=dict)
    # the root link is welded to the world (a URDF whose root is a link named `world`: Drake welds it, and the reference gives
    # the model a FixedBaseSpace, drake_utils.py:329-332): no base coordinates
    # the links as Drake sees them: every <link> in document order (the order of Drake's body indices, hence of the rows of the
    # reference's inertial_parameters), a welded link's frame in its host body's frame, every link's OWN inertia
    link_order: List[str] = field(default_factory=list)
    welded_frames: dict = field(default_factory=dict)   # welded link -> (origin, rotation) in the frame of the body it ends up in
    own_inertia: dict = field(default_factory=dict)     # link -> (mass, com, inertia_cm) before anything was folded into it
    fixed_base: bool = False
    mount_origin: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])   # of a fixed base: its root frame in the world
    mount_rotation: List[List[float]] = field(default_factory=lambda: _identity())
    # actuators: the URDF's <transmission> elements in file order, each on one joint (Drake adds a JointActuator per
    # transmission; the reference's lagrangian_forces carries B u, multibody_terms.py:142-146).  Entry k = index of the
    # actuated joint (joint j drives body j + 1) -- the width of the `u` the model takes is len(actuators)
    actuators: List[int] = field(default_factory=list)

    @property
    def n_u(self) -> int:
        return len(self.actuators)

    def inertia_rows(self) -> List[InertiaRow]:
        """The rows of ``inertial_parameters``.  A model without welded links: one per body, in body order (what every round
        so far had).  A model whose ``fixed`` joints weld links that carry mass: one per LINK in document order, as the
        reference's parameter tree has them (``drake_utils.py:129-146``: Drake's bodies in index order) -- a kernel body's
        inertia is then the sum of its links' (``csrc/dpll_weld.hip``).  A welded link without mass and inertia (a frame) has
        no row: the reference's ``theta`` has no finite value for it (``log m``)."""
        index = {body.name: i for i, body in enumerate(self.bodies)}
        heavy = []
        for name in self.welded:
            mass, _, inertia = self.own_inertia[name]
            if mass > 0.0:
                heavy.append(name)
            elif mass < 0.0 or any(v != 0.0 for v in inertia):
                raise ValueError(f'link {name}: a welded link needs a positive mass or no inertia at all')
        if not heavy:
            return [InertiaRow(body.name, i, body.mass, list(body.com), list(body.inertia_cm)) for i, body in enumerate(self.bodies)]
        rows = []
        for name in self.link_order:
            if name in index:
                mass, com, inertia = self.own_inertia[name]
                rows.append(InertiaRow(name, index[name], mass, list(com), list(inertia)))
            elif name in heavy:
                mass, com, inertia = self.own_inertia[name]
                origin, rotation = self.welded_frames[name]
                rows.append(InertiaRow(name, index[self.welded[name]], mass, list(com), list(inertia), list(origin), rotation))
        return rows

    def has_welded_rows(self) -> bool:
        return len(self.inertia_rows()) != len(self.bodies)

    @property
    def n_joints(self) -> int:
        return len(self.bodies) - 1

    @property
    def n_bodies(self) -> int:
        return len(self.bodies)

    @property
    def n_q(self) -> int:
        return (0 if self.fixed_base else 7) + self.n_joints

    @property
    def n_v(self) -> int:
        return (0 if self.fixed_base else 6) + self.n_joints

    def geoms(self):
        """``[(body index, GeomSpec)]`` in body order: the order of ``friction_params[1:]`` and of the contact blocks"""
        return [(index, geom) for index, body in enumerate(self.bodies) for geom in body.geoms]

    @property
    def n_contacts(self) -> int:
        """witness points per geometry: 4 for a box / mesh / polygon (geometry.py:47-51, 490), 1 for a sphere (:440-452);
        one per body-body candidate (geometry.py:639-643)"""
        return sum(1 if geom.kind == 'sphere' else 4 for _, geom in self.geoms()) + len(self.pairs)

    def contact_slots(self) -> List[int]:
        """indices of the real contacts among the kernels' 4 slots per geometry (a sphere uses slot 0 only)"""
        out = []
        for g, (_, geom) in enumerate(self.geoms()):
            out += [4 * g] if geom.kind == 'sphere' else [4 * g + s for s in range(4)]
        return out + [4 * MAX_GEOMS + p for p in range(len(self.pairs))]  # pair p: slot p of the group behind the geometries

    def body_alignment(self) -> List[List[List[float]]]:
        """``A_b``: the orientation of body b's frame in the root's when every joint angle is zero (the product of the
        joint rotations down the tree).  The kernels work in body frames that all coincide at zero angles; a vector with
        coordinates ``v`` in the URDF's frame of body b has coordinates ``A_b v`` there (``_capi.make_desc``)."""
        out = []
        for body in self.bodies:  # (a fixed base: the root's kernel frame is the world's, so its alignment is the mount's rotation)
            out.append((self.mount_rotation if self.fixed_base else _identity()) if body.parent < 0
                       else _matmul(out[body.parent], body.joint_rotation))
        return out

    def rotated(self) -> bool:
        """some joint or collision <origin> carries a rotation"""
        eye = _identity()
        return any(_differs(b.joint_rotation, eye) or any(_differs(g.rotation, eye) for g in b.geoms) for b in self.bodies)

    def is_fast(self) -> bool:
        """the cube / elbow topology the specialised builds are written for: a serial chain of at most one joint with
        exactly one box (or mesh) per body, no frame turned against its parent's"""
        kinds = {geom.kind for _, geom in self.geoms()}
        return (self.n_joints <= 1 and not self.rotated() and all(b.joint_kind == 'revolute' for b in self.bodies)
                and not self.actuators  # (B u is built into the general build only)
                and not self.has_welded_rows()  # (rows of welded links: composed for the general / forest builds)
                and all(len(b.geoms) == 1 for b in self.bodies) and kinds in ({'box'}, {'mesh'})
                and all(b.parent == i - 1 for i, b in enumerate(self.bodies) if i > 0) and not self.pairs)

    def friction_init(self) -> List[float]:
        """``friction_params`` initial value: ground first, then every geometry in body order
        (``dair_pll/multibody_terms.py:314-317``)."""
        return [self.ground_mu] + [g.mu for b in self.bodies for g in b.geoms]


def _vec(text: Optional[str], n: int = 3) -> List[float]:
    if text is None:
        return [0.0] * n
    vals = [float(tok) for tok in text.split()]
    if len(vals) != n:
        raise ValueError(f'expected {n} numbers, got {text!r}')
    return vals


def _identity() -> List[List[float]]:
    return [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]


def _matmul(a, b):
    return [[sum(a[i][k] * b[k][j] for k in range(3)) for j in range(3)] for i in range(3)]


def _transpose(a):
    return [[a[j][i] for j in range(3)] for i in range(3)]


def _matvec(a, v):
    return [sum(a[i][k] * v[k] for k in range(3)) for i in range(3)]


def _differs(a, b, tol: float = 0.0) -> bool:
    return any(abs(a[i][j] - b[i][j]) > tol for i in range(3) for j in range(3))


def _rotation(element) -> List[List[float]]:
    """the rotation of an <origin>: URDF rpy = fixed-axis roll (x), pitch (y), yaw (z), R = Rz(yaw) Ry(pitch) Rx(roll),
    as Drake's parser reads it (the reference sees the result through the plant: CalcSpatialInertiaInBodyFrame,
    inspector.GetPoseInFrame, multibody_terms.py:161-207, 355-376)"""
    roll, pitch, yaw = _vec(element.get('rpy') if element is not None else None)
    cr, sr, cp, sp, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    return [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr]]


def _obj_vertices(path: str) -> List[List[float]]:
    out = []
    with open(path, 'r', encoding='utf8') as handle:
        for line in handle:
            tok = line.split()
            if len(tok) >= 4 and tok[0] == 'v':
                out.append([float(tok[1]), float(tok[2]), float(tok[3])])
    return out


def parse_urdf(path: str, mesh_representation: str = 'deep_support') -> ModelSpec:
    """``mesh_representation``: what a ``<mesh>`` collision element becomes -- ``'deep_support'``, the reference's
    choice (``PydrakeToCollisionGeometryFactory.convert_mesh``, geometry.py:497-504: a ``DeepSupportConvex`` initialised from
    the OBJ's bounding box), or ``'polygon'``: a ``Polygon`` over the OBJ's vertices (geometry.py:220-252; the reference
    has the class but no front end builds one)."""
    if mesh_representation not in ('deep_support', 'polygon'):
        raise ValueError("mesh_representation must be 'deep_support' or 'polygon'")
    root = ET.parse(path).getroot()
    by_name = {}
    order = []
    for link in root.findall('link'):
        inertial = link.find('inertial')
        if link.get('name') == 'world':  # (Drake's world frame: a joint from it welds a model to the world, see below)
            continue
        if inertial is None:
            raise ValueError(f'link {link.get("name")} has no <inertial>')
        origin = inertial.find('origin')
        inertia = inertial.find('inertia')
        ixx, iyy, izz, ixy, ixz, iyz = [float(inertia.get(k)) for k in ('ixx', 'iyy', 'izz', 'ixy', 'ixz', 'iyz')]
        r_bi = _rotation(origin)  # the tensor is given in the inertial frame: I_B = R I R^T in the body frame
        i_b = _matmul(_matmul(r_bi, [[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]]), _transpose(r_bi))
        body = BodySpec(name=link.get('name'),
                        mass=float(inertial.find('mass').get('value')),
                        com=_vec(origin.get('xyz') if origin is not None else None),
                        inertia_cm=[i_b[0][0], i_b[1][1], i_b[2][2], i_b[0][1], i_b[0][2], i_b[1][2]])
        for col in link.findall('collision'):
            c_origin = col.find('origin')
            mu = None
            for element in col.iter():
                if element.tag.endswith('mu_static'):
                    mu = float(element.get('value'))
            if mu is None:
                raise ValueError('collision geometry without drake:mu_static')
            geometry = col.find('geometry')
            xyz = _vec(c_origin.get('xyz') if c_origin is not None else None)
            first = len(body.geoms)
            if geometry.find('box') is not None:
                size = _vec(geometry.find('box').get('size'))
                body.geoms.append(GeomSpec('box', xyz, mu, half_lengths=[0.5 * s for s in size]))
            elif geometry.find('sphere') is not None:
                body.geoms.append(GeomSpec('sphere', xyz, mu, radius=float(geometry.find('sphere').get('radius'))))
            elif geometry.find('mesh') is not None:
                filename = geometry.find('mesh').get('filename')
                mesh_path = os.path.join(os.path.dirname(os.path.abspath(path)), filename)
                kind = 'mesh' if mesh_representation == 'deep_support' else 'polygon'
                body.geoms.append(GeomSpec(kind, xyz, mu, mesh_file=filename, vertices=_obj_vertices(mesh_path)))
            else:
                raise NotImplementedError('only <box>, <sphere> and <mesh> collision geometry is supported')
            body.geoms[first].rotation = _rotation(c_origin)
            body.geoms[first].link = body.name
        by_name[body.name] = body
        order.append(body.name)
    joints = []
    children = set()
    mount = None
    for joint in root.findall('joint'):
        if joint.get('type') not in ('continuous', 'revolute', 'prismatic', 'fixed'):  # (limits are not modelled, as in the reference)
            raise NotImplementedError(f'joint type {joint.get("type")!r} is not supported')
        j_origin = joint.find('origin')
        if joint.find('parent').get('link') == 'world':
            # a model welded to the world: its root has no coordinates (the reference gives such a model a FixedBaseSpace,
            # drake_utils.py:329-332); the joint's <origin> is where -- and how turned -- the root sits
            if joint.get('type') != 'fixed':
                raise NotImplementedError('a joint from the world must be fixed (a free model needs no joint to the world)')
            mount = (joint.find('child').get('link'), _vec(j_origin.get('xyz') if j_origin is not None else None), _rotation(j_origin))
            continue
        axis = _vec(joint.find('axis').get('xyz')) if joint.find('axis') is not None else [1.0, 0.0, 0.0]
        norm = math.sqrt(sum(a * a for a in axis))
        joints.append((joint.find('parent').get('link'), joint.find('child').get('link'),
                       _vec(j_origin.get('xyz') if j_origin is not None else None), [a / norm for a in axis],
                       _rotation(j_origin), joint.get('type') if joint.get('type') in ('prismatic', 'fixed') else 'revolute'))
        children.add(joints[-1][1])
    welded, frames = {}, {}
    link_order = list(order)
    own_inertia = {name: (by_name[name].mass, list(by_name[name].com), list(by_name[name].inertia_cm)) for name in order}
    joints = _weld_fixed_joints(by_name, order, joints, welded, frames)
    children = {child for _, child, *_ in joints}
    roots = [name for name in order if name not in children]
    if len(roots) != 1:
        raise ValueError('expected exactly one root link per URDF (dair_pll/drake_utils.py:309-335)')
    chain = [roots[0]]
    for name in chain:
        for parent, child, origin, axis, rotation, kind in joints:
            if parent == name:
                body = by_name[child]
                body.parent = chain.index(parent)
                body.joint_origin = origin
                body.joint_axis = axis
                body.joint_rotation = rotation
                body.joint_kind = kind
                chain.append(child)
    if len(chain) != len(order):
        raise ValueError('disconnected links')
    spec = ModelSpec(name=root.get('name'), bodies=[by_name[name] for name in chain], welded=welded, link_order=link_order,
                     welded_frames=frames, own_inertia=own_inertia)
    if mount is not None:
        if welded.get(mount[0], mount[0]) != chain[0]:
            raise ValueError('the link welded to the world must be the root of the model')
        spec.fixed_base, spec.mount_origin, spec.mount_rotation = True, mount[1], mount[2]
    spec.pairs = _collision_candidates(root, spec)
    # actuators (<transmission><joint name=...>): the joint's place among the model's moving joints = its child's body index - 1
    joint_child = {joint.get('name'): joint.find('child').get('link') for joint in root.findall('joint')}
    body_index = {body.name: index for index, body in enumerate(spec.bodies)}
    for transmission in root.findall('transmission'):
        named = transmission.find('joint')
        if named is None or named.get('name') not in joint_child:
            raise ValueError('a <transmission> must name one joint of the model')
        child = joint_child[named.get('name')]
        if child not in body_index or body_index[child] == 0:
            raise NotImplementedError(f'transmission on joint {named.get("name")!r}: a fixed joint cannot be actuated')
        spec.actuators.append(body_index[child] - 1)
    return spec


def _weld_fixed_joints(by_name, order, joints, welded, frames):
    """A `fixed` joint welds its child to its parent: the two links are ONE rigid body.  Drake keeps a welded link as a
    body of its own (with inertial parameters of its own: the reference would learn them separately,
    multibody_terms.py:161-207); the kernels' bodies are the links that move against each other, so here the child is folded
    into its parent at parse time -- mass, centre of mass and central inertia combined (parallel axes), its collision
    geometries and the joints hanging off it re-expressed in the parent's frame.  The parameter tree keeps the reference's
    shape all the same: ``frames`` records where every welded link sits in the body it ends up in, and
    :meth:`ModelSpec.inertia_rows` lists the links' own inertias, one learnable row each.  Returns the remaining joints."""
    def central(body):
        ixx, iyy, izz, ixy, ixz, iyz = body.inertia_cm
        return [[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]]

    def shifted(inertia, mass, offset):  # inertia about a point displaced by `offset` from the centre of mass
        d2 = sum(o * o for o in offset)
        return [[inertia[i][j] + mass * ((d2 if i == j else 0.0) - offset[i] * offset[j]) for j in range(3)] for i in range(3)]

    pending = [j for j in joints if j[5] == 'fixed']
    joints = [j for j in joints if j[5] != 'fixed']
    while pending:
        # innermost first: a weld whose child carries no further weld
        pick = next(j for j in pending if not any(other[0] == j[1] for other in pending))
        pending.remove(pick)
        parent_name, child_name, origin, _, rotation, _ = pick
        parent, child = by_name[parent_name], by_name[child_name]
        com_child = [origin[i] + sum(rotation[i][k] * child.com[k] for k in range(3)) for i in range(3)]
        inertia_child = _matmul(_matmul(rotation, central(child)), _transpose(rotation))
        mass = parent.mass + child.mass
        com = [(parent.mass * parent.com[i] + child.mass * com_child[i]) / mass for i in range(3)]
        total = [[a + b for a, b in zip(ra, rb)] for ra, rb in zip(
            shifted(central(parent), parent.mass, [com[i] - parent.com[i] for i in range(3)]),
            shifted(inertia_child, child.mass, [com[i] - com_child[i] for i in range(3)]))]
        parent.mass, parent.com = mass, com
        parent.inertia_cm = [total[0][0], total[1][1], total[2][2], total[0][1], total[0][2], total[1][2]]
        for geom in child.geoms:
            geom.origin = [origin[i] + sum(rotation[i][k] * geom.origin[k] for k in range(3)) for i in range(3)]
            geom.rotation = _matmul(rotation, geom.rotation)
            parent.geoms.append(geom)
        moved = []
        for j_parent, j_child, j_origin, j_axis, j_rotation, j_kind in joints + pending:
            if j_parent == child_name:  # a joint that hung off the welded link now hangs off its parent
                j_origin = [origin[i] + sum(rotation[i][k] * j_origin[k] for k in range(3)) for i in range(3)]
                j_rotation = _matmul(rotation, j_rotation)
                j_parent = parent_name
            moved.append((j_parent, j_child, j_origin, j_axis, j_rotation, j_kind))
        joints = [j for j in moved if j[5] != 'fixed']
        pending = [j for j in moved if j[5] == 'fixed']
        order.remove(child_name)
        del by_name[child_name]
        welded[child_name] = parent_name
        frames[child_name] = (list(origin), rotation)
        for name, target in list(welded.items()):
            if target == child_name:  # a link welded to the child earlier (innermost first) moves on with it
                welded[name] = parent_name
                inner_origin, inner_rotation = frames[name]
                frames[name] = ([origin[i] + sum(rotation[i][k] * inner_origin[k] for k in range(3)) for i in range(3)],
                                _matmul(rotation, inner_rotation))
    return joints


def _collision_candidates(root, spec: ModelSpec) -> List[Tuple[int, int]]:
    """Geometry pairs on two different bodies that can collide: Drake filters the two LINKS a joint connects, links welded
    together (one body here), and the pairs a ``drake:collision_filter_group`` lists under
    ``drake:ignored_collision_filter_group`` (``assets/contactnets_elbow.urdf:74-78`` of the reference) -- groups name links, so a
    link welded onto a body is filtered on its own: the geometries it brought along, not the whole body."""
    groups, ignores = {}, []
    for element in root:
        if element.tag.endswith('collision_filter_group'):
            name = element.get('name')
            groups[name] = {m.get('link') for m in element if m.tag.endswith('member')}
            ignores += [(name, i.get('name')) for i in element if i.tag.endswith('ignored_collision_filter_group')]
    excluded = set()
    for first, second in ignores:
        for a in groups.get(first, ()):
            for b in groups.get(second, ()):
                excluded |= {(a, b), (b, a)}
    adjacent = set()
    for joint in root.findall('joint'):
        ends = (joint.find('parent').get('link'), joint.find('child').get('link'))
        adjacent |= {ends, ends[::-1]}
    pairs = []
    geoms = spec.geoms()
    for ga, (ba, geom_a) in enumerate(geoms):
        for gb, (bb, geom_b) in enumerate(geoms):
            links = ((geom_a.link or spec.bodies[ba].name), (geom_b.link or spec.bodies[bb].name))
            if gb <= ga or ba == bb or links in excluded or links in adjacent:
                continue
            pairs.append((gb, ga) if TYPE_ORDER[geom_a.kind] > TYPE_ORDER[geom_b.kind] else (ga, gb))
    return pairs


# limits of the forest build (csrc/dpll_forest.hpp): several models per system, one wave per item, everything in LDS
FOREST_MAX_BODIES = 16
FOREST_MAX_GEOMS = 12
FOREST_MAX_PAIRS = 16
FOREST_MAX_CONTACTS = 64
FOREST_MAX_V = 32


@dataclass
class SystemSpec:
    """The models of one system -- ``init_urdfs: Dict[str, str]`` of the reference's constructor
    (``multibody_learnable_system.py:51-54``), one floating-base (or fixed-base) tree each (``drake_utils.py:309-335``) -- in
    dict order, with the body-body collision candidates of the whole plant: inside a model as :func:`_collision_candidates`
    finds them, and every geometry of one model against every geometry of another (Drake filters nothing between models)."""
    names: List[str]
    models: List[ModelSpec]
    pairs: List[Tuple[int, int]] = field(default_factory=list)  # (never between two geometries welded to the world)

    @property
    def bodies(self) -> List[BodySpec]:
        return [body for spec in self.models for body in spec.bodies]

    @property
    def n_bodies(self) -> int:
        return sum(len(spec.bodies) for spec in self.models)

    @property
    def n_joints(self) -> int:
        return sum(spec.n_joints for spec in self.models)

    def is_fast(self) -> bool:
        return False

    def inertia_rows(self) -> List[InertiaRow]:
        """the models' rows one after the other (``drake_utils.py:129-146``), ``body`` counted over the system"""
        out, first = [], 0
        for spec in self.models:
            for row in spec.inertia_rows():
                out.append(InertiaRow(row.name, first + row.body, row.mass, row.com, row.inertia_cm, row.origin, row.rotation))
            first += len(spec.bodies)
        return out

    def has_welded_rows(self) -> bool:
        return any(spec.has_welded_rows() for spec in self.models)

    @property
    def n_u(self) -> int:
        return sum(spec.n_u for spec in self.models)

    def contact_slots(self) -> List[int]:
        """the forest build's contacts ARE the model's (no padding slots)"""
        return list(range(self.n_contacts))

    def body_model(self) -> List[int]:
        """model index of every body of the system"""
        return [m for m, spec in enumerate(self.models) for _ in spec.bodies]

    def geoms(self):
        """``[(body index in the system, GeomSpec)]``: the order of ``friction_params[1:]`` and of the contact blocks"""
        out, first = [], 0
        for spec in self.models:
            out += [(first + index, geom) for index, geom in spec.geoms()]
            first += len(spec.bodies)
        return out

    @property
    def n_q(self) -> int:
        return sum((0 if spec.fixed_base else 7) + spec.n_joints for spec in self.models)

    @property
    def n_v(self) -> int:
        return sum((0 if spec.fixed_base else 6) + spec.n_joints for spec in self.models)

    def anchored_bodies(self) -> set:
        """Bodies welded to the world: the root of a fixed-base model (links welded to it were folded into it at parse time).
        Drake calls their geometries anchored and filters every anchored-anchored pair out of ``GetCollisionCandidates``
        (``drake_utils.py:178-184``); the ground half-space sits on the world body, so an anchored geometry has NO ground
        contacts and no candidate with the anchored geometry of another fixed-base model."""
        out, first = set(), 0
        for spec in self.models:
            if spec.fixed_base:
                out.add(first)
            first += len(spec.bodies)
        return out

    def ground_geoms(self) -> List[int]:
        """indices into :meth:`geoms` of the geometries that collide with the ground (those of bodies that can move)"""
        anchored = self.anchored_bodies()
        return [g for g, (body, _) in enumerate(self.geoms()) if body not in anchored]

    @property
    def n_contacts(self) -> int:
        geoms = self.geoms()
        return sum(1 if geoms[g][1].kind == 'sphere' else 4 for g in self.ground_geoms()) + len(self.pairs)

    def friction_init(self) -> List[float]:
        return [self.models[0].ground_mu] + [geom.mu for _, geom in self.geoms()]


def build_system_spec(models) -> SystemSpec:
    """``models``: ``{name: ModelSpec}`` in the order of the reference's ``init_urdfs``"""
    names, specs = list(models.keys()), list(models.values())
    system = SystemSpec(names, specs)
    offsets, count = [], 0
    for spec in specs:
        offsets.append(count)
        count += len(spec.geoms())
    own = set()
    for offset, spec in zip(offsets, specs):
        own |= {(offset + a, offset + b) for a, b in spec.pairs}
    model_of = [m for m, spec in enumerate(specs) for _ in spec.geoms()]
    geoms = system.geoms()
    anchored = system.anchored_bodies()
    pairs = []
    for ga in range(len(geoms)):
        for gb in range(ga + 1, len(geoms)):
            swap = TYPE_ORDER[geoms[ga][1].kind] > TYPE_ORDER[geoms[gb][1].kind]
            pair = (gb, ga) if swap else (ga, gb)
            if geoms[ga][0] in anchored and geoms[gb][0] in anchored:
                continue  # both welded to the world: no such candidate in Drake
            if model_of[ga] != model_of[gb] or pair in own:
                pairs.append(pair)
    system.pairs = pairs
    return system


def check_forest_supported(system: SystemSpec) -> None:
    """What the forest build takes; anything else fails loudly at construction."""
    geoms = system.geoms()
    if system.n_bodies > FOREST_MAX_BODIES:
        raise NotImplementedError(f'at most {FOREST_MAX_BODIES} bodies per system')
    if not 1 <= len(geoms) <= FOREST_MAX_GEOMS:
        raise NotImplementedError(f'between 1 and {FOREST_MAX_GEOMS} collision geometries per system')
    if len(system.pairs) > FOREST_MAX_PAIRS:
        raise NotImplementedError(f'at most {FOREST_MAX_PAIRS} body-body collision candidates (exclude the others with a '
                                  'drake:collision_filter_group)')
    if system.n_contacts > FOREST_MAX_CONTACTS or system.n_v > FOREST_MAX_V:
        raise NotImplementedError(f'at most {FOREST_MAX_CONTACTS} contacts and {FOREST_MAX_V} velocities per system')
    for spec in system.models:
        for index, body in enumerate(spec.bodies):
            if index > 0 and not 0 <= body.parent < index:
                raise NotImplementedError('links must be listed after their parent')
    for _, geom in geoms:
        if geom.kind == 'mesh':
            raise NotImplementedError('learned shapes (DeepSupportConvex) run on the general build: one model of at most '
                                      f'{MAX_JOINTS} joints and {MAX_GEOMS} geometries')
        if geom.kind == 'polygon' and not 4 <= len(geom.vertices) <= MAX_POLYGON_VERTICES:
            raise NotImplementedError(f'a polygon has 4 to {MAX_POLYGON_VERTICES} vertices (support queries return 4 of them)')


def check_supported(spec: ModelSpec) -> None:
    """What the HIP kernels are written for; anything else fails loudly at construction."""
    if spec.n_joints > MAX_JOINTS:
        raise NotImplementedError(f'at most {MAX_JOINTS} joints')
    geoms = spec.geoms()
    if spec.actuators and any(geom.kind == 'mesh' for _, geom in geoms):
        raise NotImplementedError('actuated joints on a model with learned shapes (DeepSupportConvex) are not built')
    if not 1 <= len(geoms) <= MAX_GEOMS:
        raise NotImplementedError(f'between 1 and {MAX_GEOMS} collision geometries')
    for index, body in enumerate(spec.bodies):
        if index > 0 and not 0 <= body.parent < index:
            raise NotImplementedError('links must be listed after their parent')
    for _, geom in geoms:
        if geom.kind == 'polygon' and not 4 <= len(geom.vertices) <= MAX_POLYGON_VERTICES:
            raise NotImplementedError(f'a polygon has 4 to {MAX_POLYGON_VERTICES} vertices (support queries return 4 of them)')
    if len(spec.pairs) > MAX_PAIRS:
        raise NotImplementedError(f'at most {MAX_PAIRS} body-body collision candidates (exclude the others with a '
                                  'drake:collision_filter_group)')
    # a learned shape (DeepSupportConvex) collides with the ground or with another learned shape: the reference's
    # GeometryCollider.collide has no case for a mesh against a box / sphere / polygon (TypeError, geometry.py:543-551)
    for a, b in spec.pairs:
        if (geoms[a][1].kind == 'mesh') != (geoms[b][1].kind == 'mesh'):
            raise NotImplementedError('a body-body candidate between a mesh (DeepSupportConvex) and another kind of geometry: '
                                      'the reference has no collider for it either (exclude it with a drake:collision_filter_group)')
