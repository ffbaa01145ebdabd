"""Host build of the kernels' per-item math (one lane per item) -- TEST INFRASTRUCTURE ONLY.

``dair_pll_amd`` never imports this; it exists so the arithmetic in ``csrc/dpll_core.hpp`` can be
checked against the oracle (and run under sanitizers) in the CPU-only container.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_int64, c_void_p

import numpy as np

from dair_pll_amd._capi import ModelDesc, SolverOpts

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))
_SRC = os.path.join(_HERE, 'hostsim.cpp')
_CORE = os.path.join(_REPO, 'dair_pll_amd', 'csrc', 'dpll_core.hpp')
_LIB = os.path.join(_HERE, 'libhostsim.so')
_lib = None


def build(force: bool = False, sanitize: bool = False) -> str:
    out = _LIB if not sanitize else os.path.join(_HERE, 'libhostsim_asan.so')
    parts = ('dpll_core', 'dpll_terms', 'dpll_solver', 'dpll_contact', 'dpll_loss', 'dpll_step', 'dpll_icnn', 'dpll_weld')  # (the umbrella and its parts)
    newest = max([os.path.getmtime(_SRC)] + [os.path.getmtime(_CORE.replace('dpll_core', part)) for part in parts])
    if force or not os.path.exists(out) or os.path.getmtime(out) < newest:
        flags = ['-O0', '-fsanitize=address,undefined', '-fno-omit-frame-pointer'] if sanitize else ['-O2']  # -O0: 35 s build instead of 4.5 min
        subprocess.check_call(['g++', '-std=c++17', '-shared', '-fPIC', '-Wall', '-Wno-unknown-pragmas', *flags, '-o',
                               out, _SRC])
    return out


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        assert _lib.hostsim_sizeof_model_desc() == ctypes.sizeof(ModelDesc)
    return _lib


_actuation = None  # (kept alive while the library points at it)


def set_actuation(u) -> None:
    """actuation inputs ``(B, n_u)`` of the next loss / step / step_backward calls on an actuated general model (``None``: none)"""
    global _actuation
    if u is None:
        _actuation = None
        lib().hostsim_set_actuation(None, ctypes.c_int64(0))
    else:
        _actuation = np.ascontiguousarray(u, dtype=np.float64)
        lib().hostsim_set_actuation(_actuation.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(_actuation.shape[1]))


def weld_compose(inertia_mode: int, host, transforms, theta_rows, n_bodies: int) -> np.ndarray:
    """``dpll_weld_compose`` on the host: (n_rows, 10) theta rows -> (n_bodies, 10) inertial vectors"""
    host = np.ascontiguousarray(host, dtype=np.int32)
    transforms = np.ascontiguousarray(transforms, dtype=np.float64)
    theta_rows = np.ascontiguousarray(theta_rows, dtype=np.float64)
    iota = np.zeros((n_bodies, 10))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib().hostsim_weld_compose(ctypes.c_int(inertia_mode), ctypes.c_int(len(host)), ctypes.c_int(n_bodies), p(host), p(transforms), p(theta_rows), p(iota))
    return iota


def weld_backward(inertia_mode: int, host, transforms, theta_rows, grad_iota) -> np.ndarray:
    """``dpll_weld_compose_backward`` on the host: d loss / d (bodies' inertial vectors) -> d loss / d (theta rows)"""
    host = np.ascontiguousarray(host, dtype=np.int32)
    transforms = np.ascontiguousarray(transforms, dtype=np.float64)
    theta_rows = np.ascontiguousarray(theta_rows, dtype=np.float64)
    grad_iota = np.ascontiguousarray(grad_iota, dtype=np.float64)
    out = np.zeros_like(theta_rows)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib().hostsim_weld_backward(ctypes.c_int(inertia_mode), ctypes.c_int(len(host)), p(host), p(transforms), p(theta_rows), p(grad_iota), p(out))
    return out


def default_opts(dtype) -> SolverOpts:
    if np.dtype(dtype) == np.float64:
        return SolverOpts(max_iter=100, max_ls=50, tol=1e-13, stall_tol=1e-10, ls_tol=0.9, n_stages=6, stage_max_iter=3,
                          stage_factor=3.0, stage_tol=0.3, stage_ls_tol=0.9, stage_max_ls=50, fast_ls=1, warm_start=0, wide=-1,
                          loss_stage_factor=2.5, loss_n_stages=0, f64_refine=1, mesh_gemm=0, portfolio=1)
    return SolverOpts(max_iter=60, max_ls=30, tol=1e-6, stall_tol=1e-5, ls_tol=0.9, n_stages=6, stage_max_iter=3,
                      stage_factor=3.0, stage_tol=0.3, stage_ls_tol=0.9, stage_max_ls=50, fast_ls=1, warm_start=0, wide=-1,
                          loss_stage_factor=2.5, loss_n_stages=0, f64_refine=1, mesh_gemm=0, portfolio=1)


def n_geom_slots(desc: ModelDesc) -> int:
    """geometry slots of the build that serves this model: one per body (fast builds) or always 4 (general build: three
    geometries + the slot of a body-body pair)"""
    return 4 if desc.n_geoms > 0 else desc.n_joints + 1


def geo_stride(desc: ModelDesc) -> int:
    """numbers per geometry in the `lengths` block: 3 (fast builds) or 24 (general build: room for a polygon's 8 vertices)"""
    return 24 if desc.n_geoms > 0 else 3


def general_params(spec):
    """(theta, friction, lengths) at their URDF values in the kernels' layout: friction (1 + slots,), lengths (slots, stride)
    -- a box's half lengths, a sphere's radius in column 0, a polygon's vertices row-major; unused slots of the general build
    are padded (friction 1, lengths 0)."""
    from dair_pll_amd.inertia import pi_cm_to_theta
    fast = spec.is_fast()
    slots, stride = (spec.n_joints + 1, 3) if fast else (4, 24)
    theta = np.stack([pi_cm_to_theta(np.array([b.mass] + [b.mass * c for c in b.com] + list(b.inertia_cm))) for b in spec.bodies])
    friction = np.ones(1 + slots)
    lengths = np.zeros((slots, stride))
    friction[0] = spec.ground_mu
    for g, (_, geom) in enumerate(spec.geoms()):
        friction[1 + g] = geom.mu
        if geom.kind == 'box':
            lengths[g, :3] = geom.half_lengths
        elif geom.kind == 'sphere':
            lengths[g, 0] = geom.radius
        else:
            flat = np.asarray(geom.vertices, dtype=np.float64).reshape(-1)
            lengths[g, :flat.size] = flat
    return theta, friction, lengths


def _ptr(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None


def loss(desc: ModelDesc, theta, friction, lengths, x, x_plus, dtype=np.float64, scale=None, weights=None, opts=None,
         mixed=True, want_grad=True):
    """returns dict(loss, grad (float64, [theta|friction|lengths]), force, iters)"""
    dtype = np.dtype(dtype)
    n_b = desc.n_joints + 1
    n_g = n_geom_slots(desc)
    k = 4 * n_g
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=dtype))
    theta, friction, lengths, x, x_plus = map(arr, (theta, friction, lengths, x, x_plus))
    assert friction.size == 1 + n_g and lengths.size == geo_stride(desc) * n_g
    batch = x.shape[0]
    scale = 1.0 / batch if scale is None else scale
    out_loss = np.zeros(batch, dtype=dtype)
    grad = np.zeros(10 * n_b + 1 + (1 + geo_stride(desc)) * n_g, dtype=np.float64) if want_grad else None
    force = np.zeros((batch, 3 * k), dtype=dtype)
    iters = np.zeros(batch, dtype=np.int32)
    weights = arr(weights) if weights is not None else None
    opts = opts or default_opts(dtype)
    args = [ctypes.byref(desc), ctypes.byref(opts), _ptr(theta), _ptr(friction), _ptr(lengths), _ptr(x), _ptr(x_plus),
            c_int64(batch), _ptr(weights), c_double(scale), _ptr(out_loss), _ptr(grad), _ptr(force), _ptr(iters)]
    if dtype == np.float64:
        status = lib().hostsim_loss_f64(*args)
    else:
        status = lib().hostsim_loss_f32(*args, c_int(1 if mixed else 0))
    assert status == 0
    return {'loss': out_loss, 'grad': grad, 'force': force, 'iters': iters}


def step(desc: ModelDesc, theta, friction, lengths, x, dtype=np.float64, opts=None, mixed=True):
    dtype = np.dtype(dtype)
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=dtype))
    theta, friction, lengths, x = map(arr, (theta, friction, lengths, x))
    batch = x.shape[0]
    x_next = np.zeros_like(x)
    iters = np.zeros(batch, dtype=np.int32)
    opts = opts or default_opts(dtype)
    args = [ctypes.byref(desc), ctypes.byref(opts), _ptr(theta), _ptr(friction), _ptr(lengths), _ptr(x),
            c_int64(batch), _ptr(x_next), _ptr(iters)]
    if dtype == np.float64:
        status = lib().hostsim_step_f64(*args)
    else:
        status = lib().hostsim_step_f32(*args, c_int(1 if mixed else 0))
    assert status == 0
    return x_next, iters


def mesh(desc: ModelDesc, theta, friction, weights, pert, x, x_plus, dtype=np.float64, scale=None, opts=None,
         want_grad=True, want_step=False):
    """cube with a DeepSupportConvex geometry; weights = concat[Wh, Wd0, Wd1, wout] flat."""
    dtype = np.dtype(dtype)
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=dtype))
    theta, friction, weights, pert, x, x_plus = map(arr, (theta, friction, weights, pert, x, x_plus))
    batch = x.shape[0]
    scale = 1.0 / batch if scale is None else scale
    out_loss = np.zeros(batch, dtype=dtype)
    grad = np.zeros(12 + weights.size, dtype=np.float64) if want_grad else None
    x_next = np.zeros_like(x) if want_step else None
    opts = opts or default_opts(dtype)
    fn = lib().hostsim_mesh_f64 if dtype == np.float64 else lib().hostsim_mesh_f32
    status = fn(ctypes.byref(desc), ctypes.byref(opts), _ptr(theta), _ptr(friction), _ptr(weights), _ptr(pert), _ptr(x),
                _ptr(x_plus), c_int64(batch), c_double(scale), _ptr(out_loss), _ptr(grad), _ptr(x_next))
    assert status == 0
    return {'loss': out_loss, 'grad': grad, 'x_next': x_next}


def step_backward(desc: ModelDesc, theta, friction, lengths, x, xbar_next, opts=None, want_state=False):
    """d(sum xbar_next . x_next)/d[theta | friction | lengths] (float64); with want_state also d/dx (B, n_x)."""
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    theta, friction, lengths, x, xbar_next = map(arr, (theta, friction, lengths, x, xbar_next))
    n_b = desc.n_joints + 1
    grad = np.zeros(10 * n_b + 1 + (1 + geo_stride(desc)) * n_geom_slots(desc), dtype=np.float64)
    opts = opts or default_opts(np.float64)
    xbar = np.zeros_like(x) if want_state else None
    status = lib().hostsim_step_backward_f64(ctypes.byref(desc), ctypes.byref(opts), _ptr(theta), _ptr(friction),
                                             _ptr(lengths), _ptr(x), _ptr(xbar_next), c_int64(x.shape[0]), _ptr(grad), _ptr(xbar))
    assert status == 0
    return (grad, xbar) if want_state else grad


def pair_direction(kind_a: int, verts_a, kind_b: int, verts_b):
    """csrc/dpll_core.hpp pair_direction on two vertex sets (n, 3) in one frame -> unit direction from A to B"""
    va = np.ascontiguousarray(np.asarray(verts_a, dtype=np.float64))
    vb = np.ascontiguousarray(np.asarray(verts_b, dtype=np.float64))
    out = np.zeros(3)
    status = lib().hostsim_pair_direction(c_int(kind_a), _ptr(va), c_int(va.shape[0]), c_int(kind_b), _ptr(vb), c_int(vb.shape[0]), _ptr(out))
    assert status == 0
    return out
