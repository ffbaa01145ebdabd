"""csrc/dpll_gjk.hpp -- the direction between two learned convex shapes (DeepSupportConvex x DeepSupportConvex, the pair
the reference sends to fcl, geometry.py:585-643) -- compiled for the host (one lane) against the oracle's exact method:
the convex hull of the Minkowski difference of the two vertex sets (oracle.pair_direction_exact)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ASSET_DIR, GOLDEN_DIR
from oracle import dpll_oracle as O

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hostsim')
SRC, LIB = os.path.join(HERE, 'gjk_host.cpp'), os.path.join(HERE, 'libgjk_host.so')
HEADER = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'dair_pll_amd', 'csrc', 'dpll_gjk.hpp')


@pytest.fixture(scope='module')
def gjk():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HEADER)):
        subprocess.check_call(['g++', '-std=c++17', '-O2', '-shared', '-fPIC', '-Wall', '-Wno-unknown-pragmas', '-o', LIB, SRC])
    lib = ctypes.CDLL(LIB)

    def run(va, vb, R, p):
        va, vb, R, p = (np.ascontiguousarray(a, dtype=np.float64) for a in (va, vb, R, p))
        d, sep, info = np.zeros(3), ctypes.c_double(0.0), (ctypes.c_int * 3)()
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        assert lib.gjk_host_direction(ptr(va), va.shape[0], ptr(vb), vb.shape[0], ptr(R), ptr(p), ptr(d), ctypes.byref(sep), info) == 0
        return d, sep.value, list(info)
    return run


def random_rotation(rng):
    w, x, y, z = (lambda q: q / np.linalg.norm(q))(rng.normal(size=4))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def random_cloud(rng, n, scale):
    """points on a random ellipsoid, some pulled inside (not hull vertices), some repeated (the 296 support points of a
    network hold duplicates where several surface directions share a vertex)"""
    x = rng.normal(size=(n, 3))
    x = x / np.linalg.norm(x, axis=1, keepdims=True) * scale * rng.uniform(0.5, 1.5, size=3)
    x[rng.integers(0, n, size=n // 8)] *= 0.5
    x[rng.integers(0, n, size=n // 10)] = x[0]
    return x


def test_direction_of_random_convex_clouds_apart_and_overlapping(gjk):
    rng = np.random.default_rng(0)
    errors, seps, epa = [], [], []
    for _ in range(160):
        va, vb = random_cloud(rng, rng.integers(20, 297), 0.05), random_cloud(rng, rng.integers(20, 297), 0.04)
        R = random_rotation(rng)
        direction = rng.normal(size=3)
        p = direction / np.linalg.norm(direction) * rng.choice([0.02, 0.05, 0.07, 0.08, 0.09, 0.1, 0.12]) * rng.uniform(0.7, 1.3)
        d, sep, info = gjk(va, vb, R, p)
        vb_in_a = vb @ R.T + p
        ref = O.pair_direction_exact(va, vb_in_a)
        assert info[0] == 0
        assert abs(np.linalg.norm(d) - 1.0) < 1e-12
        assert abs(sep - ((vb_in_a @ ref).min() - (va @ ref).max())) < 1e-12  # the separation the direction achieves
        errors.append(np.linalg.norm(d - ref)); seps.append(sep); epa.append(info[2])
    assert max(errors) < 1e-10
    assert sum(s < 0 for s in seps) > 40 and sum(s > 0 for s in seps) > 40  # both branches (EPA / GJK)
    assert max(epa) < 90


def test_direction_between_the_fixture_networks(gjk, golden):
    """the two networks of clasp_mesh_literal: their 296 support points (duplicates kept, as the kernels keep them) at the
    fixture's states and at states with the tip pushed into the base"""
    g = golden('clasp_mesh_literal')
    prefix = 'param/multibody_terms.contact_terms.geometries.'
    sets = []
    for index in (1, 2):
        weights = {key: torch.tensor(g[f'{prefix}{index}.network.{key}']) for key in ('hidden_weights.0', 'input_weights.0', 'input_weights.1', 'output_weight')}
        sets.append(O.icnn_support_point(weights, O.surface_directions()).numpy())
    va, vb = sets
    spec = O.OracleSystem(os.path.join(ASSET_DIR, 'clasp_mesh.urdf'), float(g['dt'])).spec
    rng = np.random.default_rng(1)
    q = torch.tensor(g['x_plus'][:, :9]).clone()
    pushed = q.clone()
    pushed[:, 7:] += torch.tensor(0.3 * rng.normal(size=(q.shape[0], 2)))
    worst, overlapping = 0.0, 0
    for states in (q, pushed):
        R_WC, p_W, _ = O.geometry_kinematics(spec, states)
        R_AW = R_WC[:, 1].transpose(-1, -2)
        R_AB = (R_AW @ R_WC[:, 2]).numpy()
        p_AB = (R_AW @ (p_W[:, 2] - p_W[:, 1]).unsqueeze(-1)).squeeze(-1).numpy()
        for n in range(states.shape[0]):
            d, sep, info = gjk(va, vb, R_AB[n], p_AB[n])
            ref = O.pair_direction_exact(np.unique(va, axis=0), np.unique(vb, axis=0) @ R_AB[n].T + p_AB[n])
            assert info[0] == 0
            worst = max(worst, np.linalg.norm(d - ref))
            overlapping += sep < 0
    assert worst < 1e-10 and overlapping >= 5


def test_degenerate_inputs_do_not_hang(gjk):
    """coincident shapes, a shape inside the other, flat and single-point clouds: a unit direction comes back"""
    rng = np.random.default_rng(2)
    cube = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64) * 0.05
    eye = np.eye(3)
    for va, vb, p in ((cube, cube, np.zeros(3)), (cube, 0.2 * cube, np.array([0.01, 0.0, 0.0])), (cube, cube, np.array([0.1, 0.0, 0.0])),
                      (cube, cube[:1], np.array([0.2, 0.1, 0.0])), (cube, cube * np.array([1, 1, 0]), np.array([0.0, 0.0, 0.2])),
                      (random_cloud(rng, 50, 0.05), random_cloud(rng, 50, 0.05), np.zeros(3))):
        d, sep, info = gjk(va, vb, eye, p)
        assert np.isfinite(d).all() and abs(np.linalg.norm(d) - 1.0) < 1e-9
