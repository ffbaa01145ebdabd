"""ctypes binding of ``libdpll_hip.so`` (``include/dpll.h``).

There is no CPU fallback: if the library is missing or a call fails this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p
from typing import Optional

from .urdf import ModelSpec, _differs, _matmul, _matvec, _transpose, check_supported  # noqa: F401

MAX_JOINTS = 3
MAX_BODIES = 4
MAX_GEOMS = 3
MAX_PAIRS = 4
GEN_SLOTS = MAX_GEOMS + 1  # geometry slots of the general build (DPLL_GEN_SLOTS): the geometries + the pairs' group
GEOM_KINDS = {'box': 0, 'sphere': 1, 'polygon': 2, 'mesh': 3}  # dpll_geom_kind
JOINT_KINDS = {'revolute': 0, 'prismatic': 1}
GEOM_BLOCK = 24  # DPLL_GEOM_BLOCK: numbers per geometry in the general build's `lengths` block
F32, F64 = 0, 1
ABI_VERSION = 25  # dpll_abi_version() of include/dpll.h as bound below
INERTIA_MODES = {'reference_literal': 0, 'physical': 1}
MAX_WELD_ROWS = 64  # DPLL_MAX_WELD_ROWS
INERTIA_COMPOSED = 2  # DPLL_INERTIA_COMPOSED: set by make_desc / make_forest_desc for models with welded links, never by a caller

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DPLL_HIP_LIBRARY: a diagnostic knob -- tools/diag A/B runs load another build of the same library; never a fallback)
LIB_PATH = os.environ.get('DPLL_HIP_LIBRARY') or os.path.join(_HERE, 'csrc', 'libdpll_hip.so')


class ModelDesc(ctypes.Structure):
    """``dpll_model_desc_t``"""
    _fields_ = [('n_joints', c_int32), ('inertia_mode', c_int32), ('dt', c_double), ('gravity_z', c_double),
                ('joint_origin', (c_double * 3) * MAX_JOINTS), ('joint_axis', (c_double * 3) * MAX_JOINTS),
                ('geom_origin', (c_double * 3) * MAX_GEOMS), ('parent', c_int32 * MAX_JOINTS), ('n_geoms', c_int32),
                ('geom_body', c_int32 * MAX_GEOMS), ('geom_kind', c_int32 * MAX_GEOMS), ('geom_nverts', c_int32 * MAX_GEOMS), ('n_pairs', c_int32),
                ('pair_a', c_int32 * MAX_PAIRS), ('pair_b', c_int32 * MAX_PAIRS), ('rotated', c_int32),
                ('body_rot', ((c_double * 3) * 3) * MAX_BODIES), ('geom_rot', ((c_double * 3) * 3) * MAX_GEOMS),
                ('joint_kind', c_int32 * MAX_JOINTS), ('n_u', c_int32), ('act_joint', c_int32 * MAX_JOINTS), ('reserved', c_int32)]


# ---- the forest build (csrc/dpll_forest.hpp): several models in one system, any number of joints / geometries / candidates ----
FOREST_MAX_BODIES = 16
FOREST_MAX_GEOMS = 12
FOREST_MAX_PAIRS = 16
FOREST_MAX_CONTACTS = 64
FOREST_MAX_V = 32
JOINT_FLOATING, JOINT_FIXED = 2, 3


class ForestDesc(ctypes.Structure):
    """``dpll_forest_desc_t``"""
    _fields_ = [('n_bodies', c_int32), ('n_geoms', c_int32), ('n_pairs', c_int32), ('n_contacts', c_int32), ('n_q', c_int32),
                ('n_v', c_int32), ('inertia_mode', c_int32), ('rotated', c_int32), ('max_depth', c_int32), ('n_u', c_int32),
                ('dt', c_double), ('gravity_z', c_double),
                ('parent', c_int32 * FOREST_MAX_BODIES), ('joint_kind', c_int32 * FOREST_MAX_BODIES),
                ('q_index', c_int32 * FOREST_MAX_BODIES), ('v_index', c_int32 * FOREST_MAX_BODIES), ('depth', c_int32 * FOREST_MAX_BODIES),
                ('joint_origin', (c_double * 3) * FOREST_MAX_BODIES), ('joint_axis', (c_double * 3) * FOREST_MAX_BODIES),
                ('body_rot', ((c_double * 3) * 3) * FOREST_MAX_BODIES), ('dof_body', c_int32 * FOREST_MAX_V),
                ('geom_body', c_int32 * FOREST_MAX_GEOMS), ('geom_kind', c_int32 * FOREST_MAX_GEOMS), ('geom_nverts', c_int32 * FOREST_MAX_GEOMS),
                ('geom_origin', (c_double * 3) * FOREST_MAX_GEOMS), ('geom_rot', ((c_double * 3) * 3) * FOREST_MAX_GEOMS),
                ('pair_a', c_int32 * FOREST_MAX_PAIRS), ('pair_b', c_int32 * FOREST_MAX_PAIRS),
                ('contact_geom', c_int32 * FOREST_MAX_CONTACTS), ('contact_slot', c_int32 * FOREST_MAX_CONTACTS),
                ('act_body', c_int32 * FOREST_MAX_V)]


class SolverOpts(ctypes.Structure):
    """``dpll_solver_opts_t``"""
    _fields_ = [('max_iter', c_int32), ('max_ls', c_int32), ('tol', c_double), ('stall_tol', c_double),
                ('ls_tol', c_double), ('n_stages', c_int32), ('stage_max_iter', c_int32), ('stage_factor', c_double),
                ('stage_tol', c_double), ('stage_ls_tol', c_double), ('stage_max_ls', c_int32), ('fast_ls', c_int32),
                ('warm_start', c_int32), ('wide', c_int32), ('loss_stage_factor', c_double), ('loss_n_stages', c_int32),
                ('f64_refine', c_int32), ('mesh_gemm', c_int32), ('portfolio', c_int32),
                ('race_stages', c_int32 * 3), ('race_flags', c_int32 * 3), ('race_factor', c_double * 3)]


class Params(ctypes.Structure):
    """``dpll_params_t``"""
    _fields_ = [('theta', c_void_p), ('friction', c_void_p), ('lengths', c_void_p), ('u', c_void_p), ('ld_u', ctypes.c_int64)]


class AdamState(ctypes.Structure):
    """``dpll_adam_t``"""
    _fields_ = [('params', c_void_p), ('exp_avg', c_void_p), ('exp_avg_sq', c_void_p), ('state', c_void_p), ('lr', c_double),
                ('beta1', c_double), ('beta2', c_double), ('eps', c_double), ('weight_decay', c_double)]


class MeshParams(ctypes.Structure):
    """``dpll_mesh_params_t``"""
    _fields_ = [('hidden_weight', c_void_p), ('input_weight0', c_void_p), ('input_weight1', c_void_p),
                ('output_weight', c_void_p), ('perturbations', c_void_p)]


def make_desc(spec: ModelSpec, dt: float, inertia_mode: str = 'reference_literal') -> ModelDesc:
    """``dpll_model_desc_t`` of a parsed model.  The cube / elbow topologies (a serial chain of at most one joint with
    one box -- or one mesh -- per body) use the fast builds (``n_geoms = 0``); everything else the general build."""
    check_supported(spec)
    desc = ModelDesc()
    desc.n_joints = spec.n_joints
    # (links welded on with rows of their own: the kernels are handed the bodies' composed inertial vectors, csrc/dpll_weld.hip)
    desc.inertia_mode = INERTIA_COMPOSED if spec.has_welded_rows() else INERTIA_MODES[inertia_mode]
    desc.dt = dt
    desc.gravity_z = spec.gravity_z
    # The kernels' body frames all coincide at zero joint angles.  A URDF whose joint <origin>s carry a rotation is
    # re-expressed in such frames: body b's is its URDF frame turned back by A_b (ModelSpec.body_alignment), so the joint
    # data is rotated by A here, the inertial parameters -- learnable, in the URDF's frames like the reference's -- by
    # body_rot in the kernels, and a geometry's own orientation R_BG becomes geom_rot = A_b R_BG (its origin is given
    # in that geometry frame).
    eye = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    align = spec.body_alignment()
    for index, body in enumerate(spec.bodies):
        for r in range(3):
            for c in range(3):
                desc.body_rot[index][r][c] = align[index][r][c]
        if index > 0:
            desc.parent[index - 1] = body.parent
            desc.joint_kind[index - 1] = JOINT_KINDS[body.joint_kind]
            origin, hinge = _matvec(align[body.parent], body.joint_origin), _matvec(align[index], body.joint_axis)
            for axis in range(3):
                desc.joint_origin[index - 1][axis] = origin[axis]
                desc.joint_axis[index - 1][axis] = hinge[axis]
    for index in range(len(spec.bodies), MAX_BODIES):
        for r in range(3):
            desc.body_rot[index][r][r] = 1.0
    geoms = spec.geoms()
    for g in range(MAX_GEOMS):
        for r in range(3):
            desc.geom_rot[g][r][r] = 1.0
    for g, (body_index, geom) in enumerate(geoms):
        desc.geom_body[g] = body_index
        desc.geom_kind[g] = GEOM_KINDS.get(geom.kind, 0)
        desc.geom_nverts[g] = len(geom.vertices) if geom.kind == 'polygon' else 0
        frame = _matmul(align[body_index], geom.rotation)
        origin = _matvec(_transpose(geom.rotation), geom.origin)
        for r in range(3):
            desc.geom_origin[g][r] = origin[r]
            for c in range(3):
                desc.geom_rot[g][r][c] = frame[r][c]
    desc.rotated = ((1 if any(_differs(a, eye) for a in align) else 0)
                    | (2 if any(_differs(_matmul(align[b], geom.rotation), eye) for b, geom in geoms) else 0))
    desc.n_geoms = 0 if spec.is_fast() else len(geoms)
    for p, (a, b) in enumerate(spec.pairs):
        desc.pair_a[p], desc.pair_b[p] = a, b
    desc.n_pairs = len(spec.pairs)
    desc.n_u = len(spec.actuators)  # (B u of the reference's lagrangian_forces, multibody_terms.py:142-146: general build)
    for k, joint in enumerate(spec.actuators):
        desc.act_joint[k] = joint
    return desc


def make_forest_desc(system_spec, dt: float, inertia_mode: str = 'reference_literal') -> ForestDesc:
    """``dpll_forest_desc_t`` of a :class:`dair_pll_amd.urdf.SystemSpec` (one or several models).  State layout of the
    reference's ``ProductSpace`` (``drake_utils.py:309-335``, ``state_space.py:650-730``): ``q`` = the models' coordinates one
    after the other, each ``[quaternion wxyz, position, joint coordinates]`` (a fixed-base model: joint coordinates only), ``v``
    likewise ``[omega_body, v_world, joint rates]``.  Frames are re-expressed as in :func:`make_desc`: the bodies' frames of
    one model coincide at zero joint angles."""
    from .urdf import check_forest_supported
    check_forest_supported(system_spec)
    desc = ForestDesc()
    # (links welded on with rows of their own: the kernels are handed the bodies' composed inertial vectors, csrc/dpll_weld.hip)
    desc.inertia_mode = INERTIA_COMPOSED if system_spec.has_welded_rows() else INERTIA_MODES[inertia_mode]
    desc.dt = dt
    desc.gravity_z = system_spec.models[0].gravity_z
    eye = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    n_b = q_off = v_off = 0
    body_rot_turned = geom_rot_turned = False
    geoms = []  # (global body index, GeomSpec, alignment of its body)
    for spec in system_spec.models:
        align = spec.body_alignment()
        first = n_b
        n_joints = spec.n_joints
        fixed = getattr(spec, 'fixed_base', False)
        for index, body in enumerate(spec.bodies):
            b = first + index
            for r in range(3):
                for c in range(3):
                    desc.body_rot[b][r][c] = align[index][r][c]
            body_rot_turned = body_rot_turned or _differs(align[index], eye)
            if index == 0:
                desc.parent[b] = -1
                desc.joint_kind[b] = JOINT_FIXED if fixed else JOINT_FLOATING
                desc.q_index[b], desc.v_index[b], desc.depth[b] = q_off, v_off, 0
                for axis in range(3):  # (a fixed base: where its root sits in the world)
                    desc.joint_origin[b][axis] = spec.mount_origin[axis] if fixed else 0.0
            else:
                desc.parent[b] = first + body.parent
                desc.joint_kind[b] = JOINT_KINDS[body.joint_kind]
                desc.q_index[b] = q_off + (0 if fixed else 7) + index - 1
                desc.v_index[b] = v_off + (0 if fixed else 6) + index - 1
                desc.depth[b] = desc.depth[first + body.parent] + 1
                origin, hinge = _matvec(align[body.parent], body.joint_origin), _matvec(align[index], body.joint_axis)
                for axis in range(3):
                    desc.joint_origin[b][axis] = origin[axis]
                    desc.joint_axis[b][axis] = hinge[axis]
            for geom in body.geoms:
                geoms.append((b, geom, align[index]))
        for i in range(0 if fixed else 6):
            desc.dof_body[v_off + i] = first
        for j in range(n_joints):
            desc.dof_body[v_off + (0 if fixed else 6) + j] = first + 1 + j
        # actuators: the models' <transmission>s one after the other (the plant's actuator order); joint j drives body j + 1
        for joint in spec.actuators:
            desc.act_body[desc.n_u] = first + 1 + joint
            desc.n_u += 1
        n_b += len(spec.bodies)
        q_off += (0 if fixed else 7) + n_joints
        v_off += (0 if fixed else 6) + n_joints
    desc.n_bodies, desc.n_q, desc.n_v = n_b, q_off, v_off
    desc.max_depth = max(desc.depth[b] for b in range(n_b))
    contact = 0
    ground = set(system_spec.ground_geoms())
    for g, (b, geom, align_b) in enumerate(geoms):
        desc.geom_body[g] = b
        desc.geom_kind[g] = GEOM_KINDS[geom.kind]
        desc.geom_nverts[g] = len(geom.vertices) if geom.kind == 'polygon' else 0
        frame = _matmul(align_b, geom.rotation)
        origin = _matvec(_transpose(geom.rotation), geom.origin)
        geom_rot_turned = geom_rot_turned or _differs(frame, eye)
        for r in range(3):
            desc.geom_origin[g][r] = origin[r]
            for c in range(3):
                desc.geom_rot[g][r][c] = frame[r][c]
        if g not in ground:  # a geometry welded to the world does not meet the ground (urdf.SystemSpec.anchored_bodies)
            continue
        for slot in range(1 if geom.kind == 'sphere' else 4):
            desc.contact_geom[contact], desc.contact_slot[contact] = g, slot
            contact += 1
    desc.n_geoms = len(geoms)
    for p, (a, b) in enumerate(system_spec.pairs):
        desc.pair_a[p], desc.pair_b[p] = a, b
        desc.contact_geom[contact], desc.contact_slot[contact] = -1, p
        contact += 1
    desc.n_pairs = len(system_spec.pairs)
    desc.n_contacts = contact
    desc.rotated = (1 if body_rot_turned else 0) | (2 if geom_rot_turned else 0)
    return desc


def weld_transform(rotation, origin):
    """``X`` (10 x 10) with ``iota_body = X iota_link``: an inertial vector ``[m, m c, I_o (xx, yy, zz, xy, xz, yz)]`` given in a
    link's frame, about the link's origin, taken to the frame of the body the link is welded on -- a point ``p`` of the link
    sits at ``R p + t`` there.  ``m`` stays, ``h = m c -> R h + m t``, ``I_o -> R I_o R^T + m (|t|^2 1 - t t^T) + 2 (t . R h) 1
    - t (R h)^T - (R h) t^T`` (the integral of ``|p|^2 1 - p p^T``): linear in the vector, which is why a body's inertia is a sum
    over its links that the gradient passes through unchanged (``include/dpll.h``: dpll_weld_compose)."""
    import numpy as np
    R, t = np.asarray(rotation, dtype=np.float64), np.asarray(origin, dtype=np.float64)
    X = np.zeros((10, 10))
    slots = [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]
    for j in range(10):
        e = np.zeros(10)
        e[j] = 1.0
        m, h = e[0], e[1:4]
        I = np.array([[e[4], e[7], e[8]], [e[7], e[5], e[9]], [e[8], e[9], e[6]]])
        Rh = R @ h
        I_new = (R @ I @ R.T + m * (t @ t * np.eye(3) - np.outer(t, t))
                 + 2.0 * (t @ Rh) * np.eye(3) - np.outer(t, Rh) - np.outer(Rh, t))
        X[0, j] = m
        X[1:4, j] = Rh + m * t
        X[4:, j] = [I_new[a, b] for a, b in slots]
    return X


class DpllError(RuntimeError):
    pass


_lib: Optional[ctypes.CDLL] = None


def library() -> ctypes.CDLL:
    """Loads the HIP library or raises -- callers never get a silent fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DpllError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        f'or `make -C {os.path.join(_HERE, "csrc")}`; there is no CPU fallback')
    lib = ctypes.CDLL(LIB_PATH)
    lib.dpll_last_error.restype = c_char_p
    lib.dpll_abi_version.restype = c_int
    if lib.dpll_abi_version() != ABI_VERSION:  # a stale build with another argument layout must not load silently
        raise DpllError(f'{LIB_PATH} has ABI version {lib.dpll_abi_version()}, this binding needs {ABI_VERSION}: rebuild it '
                        f'(`make -C {os.path.join(_HERE, "csrc")}`)')
    lib.dpll_model_create.argtypes = [POINTER(ModelDesc), POINTER(c_void_p)]
    lib.dpll_forest_model_create.argtypes = [POINTER(ForestDesc), POINTER(c_void_p)]
    lib.dpll_model_destroy.argtypes = [c_void_p]
    lib.dpll_model_destroy.restype = None
    lib.dpll_model_set_solver.argtypes = [c_void_p, c_int, POINTER(SolverOpts)]
    lib.dpll_model_get_solver.argtypes = [c_void_p, c_int, POINTER(SolverOpts)]
    for name in ('dpll_n_x', 'dpll_n_contacts', 'dpll_param_count'):
        getattr(lib, name).argtypes = [c_void_p]
    lib.dpll_workspace_bytes.argtypes = [c_void_p, c_int64]
    lib.dpll_workspace_bytes.restype = c_int64
    lib.dpll_racing_copies.argtypes = [c_void_p, c_int, c_int64, c_int]
    lib.dpll_racing_copies.restype = c_int
    lib.dpll_contactnets_loss.argtypes = [c_void_p, c_int, POINTER(Params), c_void_p, c_int64, c_void_p, c_int64,
                                          c_int64, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int64, c_void_p]
    lib.dpll_profile_contactnets_loss.argtypes = [c_void_p, c_int, POINTER(Params), c_void_p, c_int64, c_void_p, c_int64,
                                                  c_int64, c_double, c_void_p, c_void_p, c_int64, c_void_p, c_int32,
                                                  POINTER(ctypes.c_float), POINTER(ctypes.c_float)]
    lib.dpll_step.argtypes = [c_void_p, c_int, POINTER(Params), c_void_p, c_int64, c_int64, c_void_p, c_int64,
                              c_void_p, c_void_p]
    lib.dpll_step_backward.argtypes = [c_void_p, c_int, POINTER(Params), c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                       c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]
    lib.dpll_simulate.argtypes = [c_void_p, c_int, POINTER(Params), c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                  c_void_p]
    lib.dpll_mesh_param_count.argtypes = [c_void_p]
    lib.dpll_mesh_workspace_bytes.argtypes = [c_void_p, c_int64, c_int]
    lib.dpll_mesh_workspace_bytes.restype = c_int64
    lib.dpll_contactnets_loss_mesh.argtypes = [c_void_p, c_int, POINTER(Params), POINTER(MeshParams), c_void_p, c_int64,
                                               c_void_p, c_int64, c_int64, c_void_p, c_double, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
    lib.dpll_profile_contactnets_loss_mesh.argtypes = [c_void_p, c_int, POINTER(Params), POINTER(MeshParams), c_void_p, c_int64,
                                                       c_void_p, c_int64, c_int64, c_double, c_void_p, c_void_p, c_int64,
                                                       c_void_p, c_int32, POINTER(ctypes.c_float)]
    lib.dpll_step_backward_mesh.argtypes = [c_void_p, c_int, POINTER(Params), POINTER(MeshParams), c_void_p, c_int64, c_void_p,
                                            c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]
    lib.dpll_terms_mesh.argtypes = [c_void_p, c_int, POINTER(Params), POINTER(MeshParams), c_void_p, c_int64, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
    lib.dpll_step_mesh.argtypes = [c_void_p, c_int, POINTER(Params), POINTER(MeshParams), c_void_p, c_int64, c_int64,
                                   c_void_p, c_int64, c_void_p, c_int64, c_void_p]
    lib.dpll_simulate_mesh.argtypes = [c_void_p, c_int, POINTER(Params), POINTER(MeshParams), c_void_p, c_int64, c_int64, c_int64,
                                       c_void_p, c_void_p, c_int64, c_void_p]
    lib.dpll_mesh_support_points.argtypes = [c_void_p, c_int, POINTER(MeshParams), c_void_p, c_int64, c_int64, c_void_p,
                                             c_void_p, c_int64, c_void_p]
    lib.dpll_ar_handle_bytes.restype = c_int64
    lib.dpll_ar_create.argtypes = [c_int, c_int, c_void_p, POINTER(c_void_p)]
    lib.dpll_ar_connect.argtypes = [c_void_p, c_void_p]
    lib.dpll_ar_allreduce.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p]
    lib.dpll_ar_status.argtypes = [c_void_p]
    lib.dpll_contactnets_loss_allreduce.argtypes = [c_void_p, c_int, POINTER(Params), c_void_p, c_int64, c_void_p, c_int64,
                                                    c_int64, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int64,
                                                    c_void_p, c_void_p]
    lib.dpll_contactnets_train_step.argtypes = [c_void_p, c_int, POINTER(Params), c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                                c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                                POINTER(AdamState), c_void_p]
    lib.dpll_contactnets_train_step_mesh.argtypes = [c_void_p, c_int, POINTER(Params), POINTER(MeshParams), c_void_p, c_int64, c_void_p, c_int64,
                                                     c_int64, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int64, POINTER(AdamState), c_void_p]
    lib.dpll_ar_destroy.argtypes = [c_void_p]
    lib.dpll_ar_destroy.restype = None
    lib.dpll_terms.argtypes = [c_void_p, c_int, POINTER(Params), c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p]
    lib.dpll_weld_compose.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.dpll_weld_compose_backward.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int, c_void_p]
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        message = library().dpll_last_error()
        raise DpllError(f'dpll call failed ({status}): {message.decode() if message else "?"}')


EXPORTED_SYMBOLS = ['dpll_last_error', 'dpll_abi_version', 'dpll_model_create', 'dpll_forest_model_create', 'dpll_model_destroy',
                    'dpll_model_set_solver', 'dpll_model_get_solver', 'dpll_n_x', 'dpll_n_contacts',
                    'dpll_param_count', 'dpll_workspace_bytes', 'dpll_racing_copies', 'dpll_contactnets_loss', 'dpll_profile_contactnets_loss', 'dpll_step', 'dpll_step_backward', 'dpll_simulate',
                    'dpll_terms', 'dpll_mesh_param_count', 'dpll_mesh_workspace_bytes', 'dpll_contactnets_loss_mesh',
                    'dpll_profile_contactnets_loss_mesh',
                    'dpll_step_mesh', 'dpll_simulate_mesh', 'dpll_mesh_support_points', 'dpll_ar_handle_bytes', 'dpll_ar_create', 'dpll_ar_connect',
                    'dpll_ar_allreduce', 'dpll_ar_status', 'dpll_ar_destroy', 'dpll_contactnets_loss_allreduce', 'dpll_terms_mesh', 'dpll_step_backward_mesh', 'dpll_contactnets_train_step', 'dpll_contactnets_train_step_mesh',
                    'dpll_weld_compose', 'dpll_weld_compose_backward']
