"""Host-side inertial parameter conversions (numpy, float64).

Only what construction and reporting need; the per-step ``theta -> spatial inertia`` map and its
derivative run inside the HIP kernels (``csrc/dpll_core.hpp: theta_to_iota``).  Parameterisations
follow ``dair_pll/inertia.py``: ``pi_cm = [m, m c, I_cm(xx,yy,zz,xy,xz,yz)]``, ``pi_o`` the same
about the body origin (``:108-145, 305-360``), ``theta`` the log-Cholesky vector (``:206-302``).
"""
from __future__ import annotations

import numpy as np


def _skew(v: np.ndarray) -> np.ndarray:
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def _sym(vec6: np.ndarray) -> np.ndarray:
    xx, yy, zz, xy, xz, yz = vec6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


def _vec6(mat: np.ndarray) -> np.ndarray:
    return np.array([mat[0, 0], mat[1, 1], mat[2, 2], mat[0, 1], mat[0, 2], mat[1, 2]])


def pi_cm_to_pi_o(pi_cm: np.ndarray) -> np.ndarray:
    """Parallel-axis shift com -> origin (``dair_pll/inertia.py:334-360``)."""
    mass = pi_cm[0]
    s = _skew(pi_cm[1:4] / mass)
    return np.concatenate(([mass], pi_cm[1:4], _vec6(_sym(pi_cm[4:]) - mass * (s @ s))))


def pi_o_to_pi_cm(pi_o: np.ndarray) -> np.ndarray:
    """``dair_pll/inertia.py:305-331``."""
    mass = pi_o[0]
    s = _skew(pi_o[1:4] / mass)
    return np.concatenate(([mass], pi_o[1:4], _vec6(_sym(pi_o[4:]) + mass * (s @ s))))


def pi_o_to_theta(pi_o: np.ndarray) -> np.ndarray:
    """Inverse of the log-Cholesky map (``dair_pll/inertia.py:237-302``)."""
    m, h1, h2, h3, ixx, iyy, izz, ixy, ixz, iyz = pi_o
    a_e1 = np.sqrt(0.5 * (iyy + izz - ixx))
    a_s12 = -ixy / a_e1
    a_s13 = -ixz / a_e1
    a_e2 = np.sqrt(izz - a_e1**2 - a_s12**2)
    a_s23 = (-iyz - a_s12 * a_s13) / a_e2
    a_e3 = np.sqrt(iyy - a_e1**2 - a_s13**2 - a_s23**2)
    a_t1 = h1 / a_e1
    a_t2 = (h2 - a_t1 * a_s12) / a_e2
    a_t3 = (h3 - a_t1 * a_s13 - a_t2 * a_s23) / a_e3
    a = np.sqrt(m - a_t1**2 - a_t2**2 - a_t3**2)
    return np.array([np.log(a), np.log(a_e1 / a), np.log(a_e2 / a), np.log(a_e3 / a), a_s12 / a, a_s23 / a,
                     a_s13 / a, a_t1 / a, a_t2 / a, a_t3 / a])


def theta_to_pi_o(theta: np.ndarray) -> np.ndarray:
    """``dair_pll/inertia.py:206-234``."""
    alpha, d1, d2, d3, s12, s23, s13, t1, t2, t3 = theta
    e1, e2, e3 = np.exp(d1), np.exp(d2), np.exp(d3)
    rows = np.array([t1 * t1 + t2 * t2 + t3 * t3 + 1, t1 * e1, t1 * s12 + t2 * e2, t1 * s13 + t2 * s23 + t3 * e3,
                     s12**2 + s23**2 + s13**2 + e2**2 + e3**2, s13**2 + s23**2 + e1**2 + e3**2,
                     s12**2 + e1**2 + e2**2, -s12 * e1, -s13 * e1, -s12 * s13 - s23 * e2])
    return np.exp(2 * alpha) * rows


def pi_cm_to_theta(pi_cm: np.ndarray) -> np.ndarray:
    return pi_o_to_theta(pi_cm_to_pi_o(pi_cm))


def theta_to_pi_cm(theta: np.ndarray) -> np.ndarray:
    return pi_o_to_pi_cm(theta_to_pi_o(theta))
