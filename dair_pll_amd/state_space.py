"""State layout of a floating-base chain (host-side mirror of ``dair_pll/state_space.py``).

Only the pieces the drop-in boundary needs: the ``q / v / x`` splits (``state_space.py:171-192``),
and a torch implementation of the Lie-group Euler step (``:295-311, 466-486`` with
``quaternion.py:89-147, 276-309``) for callers that drive a generic :class:`Integrator` with their
own ``partial_step`` callback.  The fused kernels (``dpll_step`` / ``dpll_simulate``) do the same
update on chip and are what :class:`MultibodyLearnableSystem` uses.

Layout (``state_space.py:412-424``): ``q = [quat wxyz, p_world, joint angles]``,
``v = [omega_body, v_world, joint rates]``.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor


def quaternion_multiply(q: Tensor, r: Tensor) -> Tensor:
    qw, qv = q[..., :1], q[..., 1:]
    rw, rv = r[..., :1], r[..., 1:]
    return torch.cat((qw * rw - (qv * rv).sum(-1, keepdim=True), qw * rv + rw * qv + torch.cross(qv, rv, dim=-1)), -1)


def quaternion_exp(r: Tensor) -> Tensor:
    half = r.norm(dim=-1, keepdim=True) / 2
    safe = torch.where(half.abs() > 0, half, torch.ones_like(half))
    sinc = torch.where(half.abs() > 0, torch.sin(safe) / safe, torch.ones_like(half))
    return torch.cat((torch.cos(half), r * sinc / 2), -1)


class FloatingBaseSpace:
    """``ProductSpace([FixedBaseSpace(0), FloatingBaseSpace(n_joints)])`` of the reference
    (``drake_utils.py:309-335``): one free body plus ``n_joints`` revolute joints."""

    def __init__(self, n_joints: int) -> None:
        self.n_joints = n_joints
        self.n_q = 7 + n_joints
        self.n_v = 6 + n_joints
        self.n_x = self.n_q + self.n_v

    def q(self, x: Tensor) -> Tensor:
        assert x.shape[-1] == self.n_x
        return x[..., :self.n_q]

    def v(self, x: Tensor) -> Tensor:
        assert x.shape[-1] == self.n_x
        return x[..., self.n_q:]

    def q_v(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        return self.q(x), self.v(x)

    def x(self, q: Tensor, v: Tensor) -> Tensor:
        assert q.shape[-1] == self.n_q and v.shape[-1] == self.n_v
        return torch.cat((q, v), -1)

    def euler_step(self, q: Tensor, v: Tensor, dt: float) -> Tensor:
        """``q (+) v dt``; the quaternion is not re-normalised (as in the reference)."""
        dq = v * dt
        quat = quaternion_multiply(q[..., :4], quaternion_exp(dq[..., :3]))
        return torch.cat((quat, q[..., 4:] + dq[..., 3:]), -1)

    def zero_state(self) -> Tensor:
        x = torch.zeros(self.n_x)
        x[0] = 1.0
        return x


class FixedBaseSpace:
    """``FixedBaseSpace(n_joints)`` (``state_space.py:556-647``): a model whose root is welded to the world -- joint coordinates only."""

    def __init__(self, n_joints: int) -> None:
        self.n_joints = n_joints
        self.n_q = self.n_v = n_joints
        self.n_x = 2 * n_joints

    def euler_step(self, q: Tensor, v: Tensor, dt: float) -> Tensor:
        return q + v * dt

    def zero_state(self) -> Tensor:
        return torch.zeros(self.n_x)


class ProductSpace(FloatingBaseSpace):
    """``ProductSpace`` (``state_space.py:650-730``): the models of one system side by side -- coordinates of every factor
    concatenated into ``q``, velocities into ``v`` (what ``drake_utils.py:309-335`` builds from the plant's models)."""

    def __init__(self, spaces) -> None:  # pylint: disable=super-init-not-called
        self.spaces = list(spaces)
        self.n_joints = sum(space.n_joints for space in self.spaces)
        self.n_q = sum(space.n_q for space in self.spaces)
        self.n_v = sum(space.n_v for space in self.spaces)
        self.n_x = self.n_q + self.n_v

    def q_split(self, q: Tensor):
        assert q.shape[-1] == self.n_q
        return list(torch.split(q, [space.n_q for space in self.spaces], dim=-1))

    def v_split(self, v: Tensor):
        assert v.shape[-1] == self.n_v
        return list(torch.split(v, [space.n_v for space in self.spaces], dim=-1))

    def euler_step(self, q: Tensor, v: Tensor, dt: float) -> Tensor:
        return torch.cat([space.euler_step(qi, vi, dt) for space, qi, vi in zip(self.spaces, self.q_split(q), self.v_split(v))], -1)

    def zero_state(self) -> Tensor:
        q = torch.cat([space.zero_state()[:space.n_q] for space in self.spaces])
        return torch.cat((q, torch.zeros(self.n_v)))
