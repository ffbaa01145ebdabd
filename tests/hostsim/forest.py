"""Host build of the forest build's per-item program (one lane per item) -- TEST INFRASTRUCTURE ONLY.

``dair_pll_amd`` never imports this; it exists so the arithmetic in ``csrc/dpll_forest.hpp`` can be checked against the oracle and
the reference-run fixtures in the CPU-only container.
"""
import ctypes
import os
import subprocess
from ctypes import c_double, c_int64, c_void_p

import numpy as np

from dair_pll_amd._capi import ForestDesc, SolverOpts

from . import default_opts

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))
_SRC = os.path.join(_HERE, 'forestsim.cpp')
_HEADERS = [os.path.join(_REPO, 'dair_pll_amd', 'csrc', name) for name in ('dpll_forest.hpp', 'dpll_core.hpp', 'dpll_terms.hpp', 'dpll_solver.hpp', 'dpll_contact.hpp', 'dpll_loss.hpp', 'dpll_step.hpp')]
_LIB = os.path.join(_HERE, 'libforestsim.so')
_lib = None
GEO_STRIDE = 24


def build(force: bool = False, sanitize: bool = False) -> str:
    out = _LIB if not sanitize else os.path.join(_HERE, 'libforestsim_asan.so')
    newest = max(os.path.getmtime(path) for path in [_SRC] + _HEADERS)
    if force or not os.path.exists(out) or os.path.getmtime(out) < newest:
        flags = ['-O1', '-fsanitize=address,undefined', '-fno-omit-frame-pointer'] if sanitize else ['-O2']
        subprocess.check_call(['g++', '-std=c++17', '-shared', '-fPIC', '-Wall', '-Wno-unknown-pragmas', '-Wno-unused-variable', *flags, '-o', out, _SRC])
    return out


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        assert _lib.forestsim_sizeof_desc() == ctypes.sizeof(ForestDesc)
    return _lib


_actuation = None


def set_actuation(u) -> None:
    """actuation inputs ``(B, n_u)`` of the next loss / step / terms / step_backward calls on an actuated system (``None``: none)"""
    global _actuation
    if u is None:
        _actuation = None
        lib().forestsim_set_actuation(None, c_int64(0))
    else:
        _actuation = np.ascontiguousarray(u, dtype=np.float64)
        lib().forestsim_set_actuation(_actuation.ctypes.data_as(c_void_p), c_int64(_actuation.shape[1]))


def params_of(system_spec):
    """(theta, friction, lengths) at their URDF values: friction (1 + n_geoms,), lengths (n_geoms, 24) -- a box's half lengths, a
    sphere's radius in column 0, a polygon's vertices row-major"""
    from dair_pll_amd.inertia import pi_cm_to_theta
    theta = np.stack([pi_cm_to_theta(np.array([b.mass] + [b.mass * c for c in b.com] + list(b.inertia_cm))) for b in system_spec.bodies])
    geoms = system_spec.geoms()
    friction = np.array(system_spec.friction_init(), dtype=np.float64)
    lengths = np.zeros((len(geoms), GEO_STRIDE))
    for g, (_, geom) in enumerate(geoms):
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


def loss(desc: ForestDesc, theta, friction, lengths, x, x_plus, dtype=np.float64, scale=None, weights=None, opts=None, want_grad=True):
    dtype = np.dtype(dtype)
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=dtype))
    theta, friction, lengths, x, x_plus = map(arr, (theta, friction, lengths, x, x_plus))
    batch, k = x.shape[0], desc.n_contacts
    assert x.shape[1] == desc.n_q + desc.n_v and friction.size == 1 + desc.n_geoms and lengths.size == GEO_STRIDE * desc.n_geoms
    scale = 1.0 / batch if scale is None else scale
    out_loss = np.zeros(batch, dtype=dtype)
    grad = np.zeros(lib().forestsim_param_count(ctypes.byref(desc)), dtype=np.float64) if want_grad else None
    force = np.zeros((batch, 3 * k), dtype=dtype)
    iters = np.zeros(batch, dtype=np.int32)
    weights = arr(weights) if weights is not None else None
    opts = opts or default_opts(dtype)
    fn = lib().forestsim_loss_f64 if dtype == np.float64 else lib().forestsim_loss_f32
    status = fn(ctypes.byref(desc), ctypes.byref(opts), _ptr(theta), _ptr(friction), _ptr(lengths), _ptr(x), _ptr(x_plus), c_int64(batch),
                _ptr(weights), c_double(scale), _ptr(out_loss), _ptr(grad), _ptr(force), _ptr(iters))
    assert status == 0
    return {'loss': out_loss, 'grad': grad, 'force': force, 'iters': iters}


def step(desc: ForestDesc, theta, friction, lengths, x, dtype=np.float64, opts=None):
    dtype = np.dtype(dtype)
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=dtype))
    theta, friction, lengths, x = map(arr, (theta, friction, lengths, x))
    x_next = np.zeros_like(x)
    iters = np.zeros(x.shape[0], dtype=np.int32)
    opts = opts or default_opts(dtype)
    fn = lib().forestsim_step_f64 if dtype == np.float64 else lib().forestsim_step_f32
    status = fn(ctypes.byref(desc), ctypes.byref(opts), _ptr(theta), _ptr(friction), _ptr(lengths), _ptr(x), c_int64(x.shape[0]), _ptr(x_next),
                _ptr(iters))
    assert status == 0
    return x_next, iters


def terms(desc: ForestDesc, theta, friction, lengths, x):
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    theta, friction, lengths, x = map(arr, (theta, friction, lengths, x))
    n, nv, k = x.shape[0], desc.n_v, desc.n_contacts
    M, a, phi, J = np.zeros((n, nv, nv)), np.zeros((n, nv)), np.zeros((n, k)), np.zeros((n, 3 * k, nv))
    status = lib().forestsim_terms_f64(ctypes.byref(desc), _ptr(theta), _ptr(friction), _ptr(lengths), _ptr(x), c_int64(n), _ptr(M), _ptr(a),
                                       _ptr(phi), _ptr(J))
    assert status == 0
    return M, a, phi, J


def step_backward(desc: ForestDesc, theta, friction, lengths, x, xbar_next, opts=None, want_state=False):
    """d(sum xbar_next . x_next)/d[theta | friction | lengths] (float64); with want_state also d/dx (B, n_x)."""
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    theta, friction, lengths, x, xbar_next = map(arr, (theta, friction, lengths, x, xbar_next))
    grad = np.zeros(lib().forestsim_param_count(ctypes.byref(desc)), dtype=np.float64)
    opts = opts or default_opts(np.float64)
    xbar = np.zeros_like(x) if want_state else None
    status = lib().forestsim_step_backward_f64(ctypes.byref(desc), ctypes.byref(opts), _ptr(theta), _ptr(friction), _ptr(lengths), _ptr(x),
                                               _ptr(xbar_next), c_int64(x.shape[0]), _ptr(grad), _ptr(xbar))
    assert status == 0
    return (grad, xbar) if want_state else grad
