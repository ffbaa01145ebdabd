"""``MultibodyLearnableSystem``: the reference's system surface on top of ``libdpll_hip.so``.

Mirrors ``dair_pll/multibody_learnable_system.py:41-333`` and ``dair_pll/system.py:47-191`` for the
hot path: same constructor, same method names / argument meaning / shapes, same ``state_dict`` keys

    multibody_terms.lagrangian_terms.inertial_parameters        (n_bodies, 10)
    multibody_terms.contact_terms.friction_params               (n_geometries,)   ground first
    multibody_terms.contact_terms.geometries.{i}.length_params  (1, 3)            boxes, i >= 1

The arithmetic runs in hand-written gfx950 kernels; PyTorch only owns device memory, streams and the
autograd plumbing.  There is no CPU fallback: every compute method raises unless its tensors live
on a ROCm device and the HIP library is built.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor
from torch.nn import Module, ModuleList, Parameter

from . import _capi
from .inertia import pi_cm_to_theta
from .integrator import VelocityIntegrator
from .state_space import FixedBaseSpace, FloatingBaseSpace, ProductSpace
from .urdf import ModelSpec, build_system_spec, check_forest_supported, check_supported, parse_urdf

_DTYPES = {torch.float32: _capi.F32, torch.float64: _capi.F64}


def _differs_from_identity(rotation) -> bool:
    return any(abs(rotation[i][j] - (1.0 if i == j else 0.0)) > 1e-12 for i in range(3) for j in range(3))


def _ptr(tensor: Optional[Tensor]) -> Optional[int]:
    return None if tensor is None else tensor.data_ptr()


class Plane(Module):
    """Ground half-space z <= 0 (``dair_pll/geometry.py:94-129``); no parameters."""


class Sphere(Module):
    """``dair_pll/geometry.py:415-456``: ``length_param`` is the radius, a scalar (its absolute value is used); ONE
    witness point per sphere.  (The reference's constructor cannot be called -- ``assert radius.numel == 1`` compares a
    bound method with 1, ``:431`` -- so no reference object of this class exists outside a ``__new__`` bypass; the
    arithmetic below it is what is reproduced.)"""

    def __init__(self, radius: Tensor) -> None:
        super().__init__()
        self.length_param = Parameter(radius.reshape(()).clone(), requires_grad=True)

    def get_radius(self) -> Tensor:
        return torch.abs(self.length_param)


class Polygon(Module):
    """``dair_pll/geometry.py:220-252``: a convex polytope given by its ``vertices`` ``(N, 3)``, all of them learnable;
    a support query returns the 4 vertices furthest along the direction."""

    def __init__(self, vertices: Tensor) -> None:
        super().__init__()
        self.vertices = Parameter(vertices.reshape(-1, 3).clone(), requires_grad=True)

    def scalars(self) -> Dict[str, float]:
        """one scalar per vertex coordinate (``geometry.py:245-252``)"""
        out = {}
        for axis, values in zip('xyz', self.vertices.detach().t()):
            for index, value in enumerate(values):
                out[f'v{index}_{axis}'] = value.item()
        return out


class Box(Module):
    """``dair_pll/geometry.py:367-412``: ``length_params`` are the half lengths, shape ``(1, 3)``."""

    def __init__(self, half_lengths: Tensor) -> None:
        super().__init__()
        self.length_params = Parameter(half_lengths.reshape(1, 3).clone(), requires_grad=True)

    def get_half_lengths(self) -> Tensor:
        return torch.abs(self.length_params)


ICNN_WIDTH = 256
ICNN_DEPTH = 2
ICNN_SLOPE = 0.5


class HomogeneousICNN(Module):
    """Parameter container of ``dair_pll/deep_support_function.py:125-194`` (depth 2, width 256,
    LeakyReLU(0.5)); initial values follow the reference's distributions (``:147-183``).  Evaluation
    (support points = input Jacobian, ``:238-266``) runs in the HIP kernels."""

    def __init__(self, scale: float, dtype: torch.dtype, device: torch.device) -> None:
        super().__init__()
        width, slope = ICNN_WIDTH, ICNN_SLOPE
        scale_hidden = 2 * (2.0 / (1 + slope**2))**0.5 / width
        hidden = [Parameter((2 * (torch.rand((width, width)) - 0.5) * scale_hidden).to(device=device, dtype=dtype))
                  for _ in range(ICNN_DEPTH - 1)]
        inputs = []
        for layer in range(ICNN_DEPTH):
            weight = torch.empty((3, width))
            torch.nn.init.kaiming_uniform_(weight)
            if layer > 0:
                weight = weight * 2**(-0.5)
            inputs.append(Parameter(weight.to(device=device, dtype=dtype)))
        scale_out = scale * 2 * (2.0 / (width * (1 + slope**2)))**0.5
        self.hidden_weights = torch.nn.ParameterList(hidden)
        self.input_weights = torch.nn.ParameterList(inputs)
        self.output_weight = Parameter((2 * (torch.rand(width) - 0.5) * scale_out).to(device=device, dtype=dtype))


class DeepSupportConvex(Module):
    """``dair_pll/geometry.py:255-325``: convex shape given by its support function network.
    ``perturbations`` (``:306-307``) is a fixed buffer outside the ``state_dict``, as in the reference;
    ``train(False)`` does NOT build an fcl mesh (quirk Q7: body-body collision is out of scope)."""

    def __init__(self, vertices: Tensor, dtype: torch.dtype, device: torch.device, n_query: int = 4,
                 perturbation: float = 0.4) -> None:
        super().__init__()
        scale = float((vertices.max(dim=0).values - vertices.min(dim=0).values).norm() / 2)
        self.network = HomogeneousICNN(scale, dtype, device)
        pert = torch.cat((torch.zeros((1, 3)), perturbation * (torch.rand((n_query - 1, 3)) - 0.5)))
        self.register_buffer('perturbations', pert.to(device=device, dtype=dtype), persistent=False)


class FusedAdamState:
    """Adam moments and step count of a system's flat parameter buffer for :meth:`MultibodyLearnableSystem.contactnets_train_step`
    (the update of ``torch.optim.Adam`` without amsgrad, ``experiment.py:213-228``, applied by the finalize kernel)."""

    def __init__(self, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0) -> None:
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg: Optional[Tensor] = None
        self.exp_avg_sq: Optional[Tensor] = None
        self.state: Optional[Tensor] = None  # (3,) float64: [steps taken, beta1^steps, beta2^steps]

    @property
    def step(self) -> Tensor:
        return self.state[:1]

    def bind(self, flat: Tensor) -> None:
        if self.exp_avg is None or self.exp_avg.shape != flat.shape or self.exp_avg.device != flat.device or self.exp_avg.dtype != flat.dtype:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
            self.state = torch.tensor([0.0, 1.0, 1.0], dtype=torch.float64, device=flat.device)

    def struct(self, flat: Tensor) -> '_capi.AdamState':
        return _capi.AdamState(flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.state.data_ptr(), self.lr,
                               self.betas[0], self.betas[1], self.eps, self.weight_decay)

    def state_dict(self) -> Dict[str, Tensor]:
        return {'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'state': self.state}

    def load_state_dict(self, state: Dict[str, Tensor]) -> None:
        self.exp_avg, self.exp_avg_sq, self.state = state['exp_avg'].clone(), state['exp_avg_sq'].clone(), state['state'].clone()


class LagrangianTerms(Module):
    def __init__(self, theta: Tensor) -> None:
        super().__init__()
        self.inertial_parameters = Parameter(theta.clone(), requires_grad=True)


class ContactTerms(Module):
    def __init__(self, friction: Tensor, geometries: List[Module]) -> None:
        super().__init__()
        self.friction_params = Parameter(friction.clone(), requires_grad=True)
        self.geometries = ModuleList(geometries)

    def get_friction_coefficients(self) -> Tensor:
        return torch.abs(self.friction_params)


class MultibodyTerms(Module):
    """Container with the reference's parameter tree; ``forward`` is ``MultibodyTerms.forward``
    (``dair_pll/multibody_terms.py:584-609``) evaluated by ``dpll_terms``."""

    def __init__(self, spec: ModelSpec, dtype: torch.dtype, device: torch.device) -> None:
        super().__init__()
        # one row per Drake body (multibody_terms.py:161-207): the kernel bodies, or -- a model whose `fixed` joints weld links
        # that carry mass -- every link in document order, each with its OWN inertia (ModelSpec.inertia_rows)
        pi_cm = np.array([[r.mass] + [r.mass * c for c in r.com] + list(r.inertia_cm) for r in spec.inertia_rows()])
        theta = np.stack([pi_cm_to_theta(row) for row in pi_cm])  # multibody_terms.py:186-188
        as_t = lambda a: torch.tensor(np.asarray(a), dtype=dtype, device=device)
        self.lagrangian_terms = LagrangianTerms(as_t(theta))
        geometries: List[Module] = [Plane()]
        for _, geom in spec.geoms():  # body order, as friction_params (multibody_terms.py:314-317)
            if geom.kind == 'box':
                geometries.append(Box(as_t(geom.half_lengths)))
            elif geom.kind == 'sphere':
                geometries.append(Sphere(as_t(geom.radius)))
            elif geom.kind == 'polygon':
                geometries.append(Polygon(as_t(geom.vertices)))
            else:
                geometries.append(DeepSupportConvex(torch.tensor(geom.vertices), dtype, device))
        self.contact_terms = ContactTerms(as_t(spec.friction_init()), geometries)
        object.__setattr__(self, '_owner', None)  # plain attribute: the owner must not become a submodule

    def forward(self, q: Tensor, v: Tensor, u: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
        assert self._owner is not None
        with self._owner._actuated(self._owner._actuation(u, q.shape[:-1])):
            return self._owner._terms(q, v)


class _LossFunction(torch.autograd.Function):
    """Per-item loss with the hand-written adjoint as its backward."""

    @staticmethod
    def forward(ctx, system, x, x_plus, u, *params):  # pylint: disable=arguments-differ
        ctx.system = system
        ctx.u = u  # (actuation inputs of the call, (B, n_u) or None: data, no gradient)
        ctx.save_for_backward(x, x_plus)
        with system._actuated(u):
            return system._launch_loss(x, x_plus, weights=None, scale=1.0, want_grad=False)[0]

    @staticmethod
    def backward(ctx, grad_output):  # pylint: disable=arguments-differ
        x, x_plus = ctx.saved_tensors
        system = ctx.system
        # a buffer of its own: the shared one is what the parameters' .grad alias after a fused step, and autograd adds
        # what is returned here ON TOP of .grad
        own = torch.empty(1 + system._packed().numel(), dtype=system.dtype, device=x.device)
        with system._actuated(ctx.u):
            _, flat_grad, _ = system._launch_loss(x, x_plus, weights=grad_output.contiguous(), scale=1.0, want_grad=True,
                                                  grad_out=own)
        return (None, None, None, None) + tuple(system._split_flat(flat_grad))


class _StepFunction(torch.autograd.Function):
    """One ``VelocityIntegrator.step``; backward = ``dpll_step_backward``: gradient with respect to the
    parameters and -- when the input state carries a graph (multi-step rollouts) -- to the state, both by
    implicit differentiation of the cone solve."""

    @staticmethod
    def forward(ctx, system, x, u, *params):  # pylint: disable=arguments-differ
        ctx.system = system
        ctx.u = u
        ctx.save_for_backward(x)
        with system._actuated(u):
            return system._step(x)

    @staticmethod
    def backward(ctx, grad_x_next):  # pylint: disable=arguments-differ
        (x,) = ctx.saved_tensors
        with ctx.system._actuated(ctx.u):
            flat_grad, grad_x = ctx.system._step_backward(x, grad_x_next.contiguous(), want_state=ctx.needs_input_grad[1])
        return (None, grad_x, None) + tuple(ctx.system._split_flat(flat_grad))


class MultibodyLearnableSystem(Module):
    """Drop-in for ``dair_pll.multibody_learnable_system.MultibodyLearnableSystem``."""

    def __init__(self, init_urdfs: Dict[str, str], dt: float, output_urdfs_dir: Optional[str] = None,
                 inertia_mode: str = 'reference_literal', dtype: torch.dtype = torch.float32,
                 device: Optional[str] = None, mesh_representation: str = 'deep_support', build: str = 'auto') -> None:
        """``mesh_representation`` (an extension; see :func:`dair_pll_amd.urdf.parse_urdf`): ``'polygon'`` turns ``<mesh>``
        collision elements into ``Polygon`` geometries over the OBJ's vertices instead of ``DeepSupportConvex``.
        ``build``: ``'auto'`` picks the kernels by the system (below); ``'forest'`` runs a system the register-resident builds
        would take on the forest build (tests hold the builds against each other)."""
        if build not in ('auto', 'forest'):
            raise ValueError("build must be 'auto' or 'forest'")
        super().__init__()
        if len(init_urdfs) < 1:
            raise ValueError('init_urdfs names no model')
        if dtype not in _DTYPES:
            raise TypeError('dtype must be torch.float32 or torch.float64')
        self.urdfs = dict(init_urdfs)
        self.output_urdfs_dir = output_urdfs_dir
        models = {name: parse_urdf(path, mesh_representation) for name, path in init_urdfs.items()}
        # One model inside the limits of the register-resident builds (cube / elbow; the general build: 3 joints, 3 geometries,
        # 4 candidates) runs there; anything else -- several models in one system (multibody_learnable_system.py:51-54,
        # drake_utils.py:309-335), longer trees, more geometries or candidates -- on the forest build (csrc/dpll_forest.hip)
        self.forest = build == 'forest' or len(models) > 1 or next(iter(models.values())).fixed_base
        if not self.forest:
            try:
                check_supported(next(iter(models.values())))
            except NotImplementedError:
                self.forest = True
        if self.forest:
            self.spec = build_system_spec(models)
            check_forest_supported(self.spec)
            self.space = ProductSpace([(FixedBaseSpace if spec.fixed_base else FloatingBaseSpace)(spec.n_joints) for spec in self.spec.models])
        else:
            self.spec = next(iter(models.values()))
            self.space = FloatingBaseSpace(self.spec.n_joints)
        self.dt = dt
        self.inertia_mode = inertia_mode
        self.dtype = dtype
        dev = torch.device(device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu'))
        self.multibody_terms = MultibodyTerms(self.spec, dtype, dev)
        object.__setattr__(self.multibody_terms, '_owner', self)
        self.integrator = VelocityIntegrator(self.space, self.sim_step, dt)
        self.integrator.fused_simulate = self._fused_simulate
        self.max_batch_dim = 1  # multibody_learnable_system.py:80
        self.carry_callback = lambda: torch.tensor([False])  # :79
        self._desc = _capi.make_forest_desc(self.spec, dt, inertia_mode) if self.forest else _capi.make_desc(self.spec, dt, inertia_mode)
        self._handle: Optional[ctypes.c_void_p] = None
        self._flat: Optional[Tensor] = None
        self._grad_buf: Optional[Tensor] = None  # [loss_total | flat gradient]: one buffer = one all-reduce
        self._flat_grad: Optional[Tensor] = None
        self._workspace: Optional[Tensor] = None
        self._loss_total: Optional[Tensor] = None
        self.grad_world = 1      # set by distributed.GradientAllReduce
        self.global_batch = 0
        self._fused_ar: Optional[ctypes.c_void_p] = None  # peer all-reduce handle: the exchange rides in the loss launch
        self._u: Optional[Tensor] = None  # actuation inputs of the call in flight (see _actuated)
        # links welded on with inertial rows of their own (the reference's parameter tree, multibody_terms.py:161-207): the
        # kernels take the BODIES' composed inertial vectors, written into the theta block of the flat buffer before a launch
        # (dpll_weld_compose) and chained back to the rows after it (dpll_weld_compose_backward)
        self._weld: Optional[Dict[str, object]] = None
        if self.spec.has_welded_rows():
            rows = self.spec.inertia_rows()
            if len(rows) > _capi.MAX_WELD_ROWS:
                raise NotImplementedError(f'at most {_capi.MAX_WELD_ROWS} inertial rows per system')
            transforms = np.stack([_capi.weld_transform(row.rotation, row.origin) for row in rows])
            self._weld = {'n_rows': len(rows), 'mode': _capi.INERTIA_MODES[inertia_mode],
                          'host': torch.tensor([row.body for row in rows], dtype=torch.int32, device=dev),
                          'transforms': torch.tensor(transforms, dtype=torch.float64, device=dev).contiguous(), 'grad': None}
        self._grad_reduced = False  # the last contactnets_loss_and_grad already summed [loss | gradients] over the ranks
        # the default form of the float32 ICNN GEMMs needs weights inside fp16's range: checked now and after every load_state_dict
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._fp16_range_or_f32_mfma())
        self._fp16_range_or_f32_mfma()

    # ---- parameters ---------------------------------------------------------------------------
    def _geom_slots(self) -> int:
        """geometry slots of the build that serves this model: one per body (specialised builds) or always ``DPLL_GEN_SLOTS``
        (general: three geometries and, behind them, the slot of a body-body pair)"""
        if self.forest:
            return len(self.spec.geoms())  # (the forest build: exactly the model's geometries)
        return self.spec.n_joints + 1 if self.spec.is_fast() else _capi.GEN_SLOTS

    def _n_contact_slots(self) -> int:
        """contacts per item in the kernels' force / phi / J / D layouts: 4 per geometry slot, or -- the forest build -- the
        model's own contacts"""
        return self.spec.n_contacts if self.forest else 4 * self._geom_slots()

    def _geo_stride(self) -> int:
        """numbers per geometry in the lengths block: a box's 3 (specialised builds) or ``DPLL_GEOM_BLOCK`` (general)"""
        return 3 if self.spec.is_fast() else _capi.GEOM_BLOCK

    def _layout(self):
        """``[(parameter, offset in the flat buffer)]`` and the buffer's length.  Layout of ``dpll_param_count``:
        ``[theta (n_bodies, 10) | friction (1 + slots) | lengths (slots, stride)]`` (a box's length_params, a sphere's
        radius or a polygon's vertices at the start of the geometry's block; the rest, and slots the model does not use, are
        padding), then -- mesh systems, which have no lengths block -- the network weights."""
        terms = self.multibody_terms
        n_b, slots, stride = self.spec.n_bodies, self._geom_slots(), self._geo_stride()
        out = [(terms.lagrangian_terms.inertial_parameters, 0), (terms.contact_terms.friction_params, 10 * n_b)]
        if self._weld is not None:  # (the rows are a tensor of their own: the buffer's theta block holds what they compose to)
            out = out[1:]
        lengths0 = 10 * n_b + 1 + slots
        # (the general build always carries its lengths block; the specialised mesh builds have none)
        end = lengths0 if self.spec.is_fast() else lengths0 + stride * slots
        for g, geometry in enumerate(list(terms.contact_terms.geometries)[1:]):
            if isinstance(geometry, (Box, Sphere, Polygon)):
                param = geometry.length_params if isinstance(geometry, Box) else (
                    geometry.length_param if isinstance(geometry, Sphere) else geometry.vertices)
                out.append((param, lengths0 + stride * g))
                end = lengths0 + stride * slots
            elif isinstance(geometry, DeepSupportConvex):
                net = geometry.network
                for p in (net.hidden_weights[0], net.input_weights[0], net.input_weights[1], net.output_weight):
                    out.append((p, end))
                    end += p.numel()
        return out, end

    def _param_list(self) -> List[Parameter]:
        head = [] if self._weld is None else [self.multibody_terms.lagrangian_terms.inertial_parameters]
        return head + [p for p, _ in self._layout()[0]]

    def _weld_call(self, backward: bool, flat_block: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
        """``dpll_weld_compose`` (rows -> the bodies' inertial vectors, into ``flat_block``) or its backward (``flat_block`` = the
        theta block of a gradient row -> the rows' gradient)"""
        weld, lib = self._weld, _capi.library()
        rows = self.multibody_terms.lagrangian_terms.inertial_parameters
        if not (rows.is_cuda and rows.is_contiguous() and rows.dtype == self.dtype and flat_block.is_contiguous()):
            raise _capi.DpllError('inertial_parameters must be a contiguous device tensor of the system\'s dtype')
        if weld['host'].device != rows.device:
            weld['host'], weld['transforms'] = weld['host'].to(rows.device), weld['transforms'].to(rows.device)
        args = (_DTYPES[self.dtype], weld['mode'], weld['n_rows'], self.spec.n_bodies, _ptr(weld['host']), _ptr(weld['transforms']))
        if not backward:
            _capi.check(lib.dpll_weld_compose(*args, _ptr(rows.data), _ptr(flat_block), self._stream()))
            return flat_block
        if out is None:
            out = torch.empty_like(rows.data)
        _capi.check(lib.dpll_weld_compose_backward(*args, _ptr(rows.data), _ptr(flat_block), _ptr(out), int(accumulate), self._stream()))
        return out

    def _mesh(self) -> Optional['DeepSupportConvex']:
        for geometry in self.multibody_terms.contact_terms.geometries:
            if isinstance(geometry, DeepSupportConvex):
                return geometry
        return None

    def _meshes(self) -> List['DeepSupportConvex']:
        return [g for g in self.multibody_terms.contact_terms.geometries if isinstance(g, DeepSupportConvex)]

    def _mesh_struct(self, flat: Tensor):
        """``dpll_mesh_params_t[]``: the specialised builds take one entry per body (one network each), the general build
        one per geometry slot (entries of geometries that are not learned shapes stay null); network k's weights sit at
        ``head + k * 67,328`` of the flat buffer."""
        base, size, width = flat.data_ptr(), flat.element_size(), ICNN_WIDTH
        per_net = width * width + 7 * width
        geometries = list(self.multibody_terms.contact_terms.geometries)[1:]
        fast = self.spec.is_fast()
        head = self._layout()[1] - per_net * len(self._meshes())
        array = (_capi.MeshParams * (len(geometries) if fast else _capi.MAX_GEOMS))()
        self._pert_keepalive = []
        k = 0
        for g, mesh in enumerate(geometries):
            if not isinstance(mesh, DeepSupportConvex):
                continue
            off = head + k * per_net
            k += 1
            pert = mesh.perturbations
            if pert.dtype != flat.dtype or pert.device != flat.device or not pert.is_contiguous():
                mesh.perturbations = pert = pert.to(device=flat.device, dtype=flat.dtype).contiguous()
            self._pert_keepalive.append(pert)
            array[g] = _capi.MeshParams(base + off * size, base + (off + width * width) * size,
                                        base + (off + width * width + 3 * width) * size,
                                        base + (off + width * width + 6 * width) * size, pert.data_ptr())
        return array

    def _mesh_workspace(self, batch: int, device) -> Tensor:
        need = _capi.library().dpll_mesh_workspace_bytes(self._model(), batch, _DTYPES[self.dtype])
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != device:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
        return self._workspace

    def n_params(self) -> int:
        return sum(p.numel() for p in self._param_list())

    def _packed(self) -> Tensor:
        """All learnable parameters as views of ONE flat device buffer (:meth:`_layout`), so a kernel call needs no
        gather and an optimizer's in-place update is seen by the next call."""
        layout, total = self._layout()
        flat = self._flat
        first = layout[0][0]
        ok = flat is not None and flat.numel() == total and flat.dtype == first.dtype and flat.device == first.device
        if ok:
            for p, offset in layout:
                if p.data_ptr() != flat.data_ptr() + offset * flat.element_size() or not p.is_contiguous():
                    ok = False
                    break
        if not ok:
            flat = torch.zeros(total, dtype=first.dtype, device=first.device)
            n_b, slots = self.spec.n_bodies, self._geom_slots()
            flat[10 * n_b:10 * n_b + 1 + slots] = 1.0  # friction of geometry slots the model does not use
            for p, offset in layout:
                flat[offset:offset + p.numel()] = p.detach().reshape(-1)
            for p, offset in layout:
                p.data = flat[offset:offset + p.numel()].view(p.shape)
            self._flat = flat
            self._flat_grad = self._grad_buf = self._loss_total = None
        if self._weld is not None and flat.is_cuda:
            self._weld_call(False, flat[:10 * self.spec.n_bodies])
        return flat

    def _alloc_grad_buffer(self, n_params: int, device) -> None:
        self._grad_buf = torch.zeros(1 + n_params, dtype=self.dtype, device=device)
        self._loss_total = self._grad_buf[:1]
        self._flat_grad = self._grad_buf[1:]

    def grad_buffer(self) -> Tensor:
        """``[mean loss | gradient of every parameter]`` of the last :meth:`contactnets_loss_and_grad`
        call, one contiguous tensor whose slices ARE the parameters' ``.grad`` -- the object a
        data-parallel all-reduce acts on."""
        if self._grad_buf is None:
            self._alloc_grad_buffer(self._packed().numel(), self._packed().device)
        return self._grad_buf

    def _split_flat(self, flat: Tensor, weld_out: Optional[Tensor] = None) -> List[Tensor]:
        """the parameters' pieces of a flat gradient; welded links: the rows' gradient chained from the theta block (into
        ``weld_out`` when given, else a new tensor)"""
        head = [] if self._weld is None else [self._weld_call(True, flat[:10 * self.spec.n_bodies], weld_out)]
        return head + [flat[offset:offset + p.numel()].view(p.shape) for p, offset in self._layout()[0]]

    def after_grad_reduce(self) -> None:
        """after a data-parallel SUM of :meth:`grad_buffer`: the rows of welded links are not views of that buffer, so their
        gradient is chained again from its (now summed) theta block -- the chain is linear in it"""
        if self._weld is not None and self._weld['grad'] is not None and self._flat_grad is not None:
            self._weld_call(True, self._flat_grad[:10 * self.spec.n_bodies], self._weld['grad'])

    # ---- native handle ------------------------------------------------------------------------
    def _model(self) -> ctypes.c_void_p:
        if self._handle is None:
            handle = ctypes.c_void_p()
            create = _capi.library().dpll_forest_model_create if self.forest else _capi.library().dpll_model_create
            _capi.check(create(ctypes.byref(self._desc), ctypes.byref(handle)))
            self._handle = handle
        return self._handle

    def __del__(self):
        handle = getattr(self, '_handle', None)
        if handle is not None and _capi._lib is not None:
            _capi._lib.dpll_model_destroy(handle)
            self._handle = None

    def racing_copies(self, batch: int, rollout: bool = False) -> int:
        """Racing copies per item (``dpll_solver_opts_t.portfolio``) of a loss (or rollout) launch of ``batch`` items."""
        if self._mesh() is not None:  # (learned shapes: the loss launch of a single body races, rollouts do not)
            return 1 if rollout else int(_capi.library().dpll_racing_copies(self._model(), _DTYPES[self.dtype], batch, 4))
        return int(_capi.library().dpll_racing_copies(self._model(), _DTYPES[self.dtype], batch, 1 if rollout else 0))

    def set_solver(self, **kwargs) -> None:
        """Override ``max_iter / max_ls / tol / stall_tol / ls_tol`` for this system's dtype."""
        lib = _capi.library()
        opts = _capi.SolverOpts()
        code = _DTYPES[self.dtype]
        _capi.check(lib.dpll_model_get_solver(self._model(), code, ctypes.byref(opts)))
        for key, value in kwargs.items():
            if isinstance(value, (tuple, list)):  # (the racing schedules: three entries each)
                value = type(getattr(opts, key))(*value)
            setattr(opts, key, value)
        if kwargs.get('mesh_gemm') == 4:
            self._check_fp16_range()  # (raises before anything is changed)
        _capi.check(lib.dpll_model_set_solver(self._model(), code, ctypes.byref(opts)))

    def _fp16_range_or_f32_mfma(self) -> None:
        """weights outside fp16's range (:meth:`_check_fp16_range`): the float32 mesh pipeline goes to the f32 MFMA kernels, with a
        warning -- a slower form of the same HIP path, not a fallback off the device"""
        if self._mesh() is None or self.dtype != torch.float32 or not next(self.parameters()).is_cuda:
            return
        try:
            self._check_fp16_range()
        except _capi.DpllError as error:
            import warnings
            warnings.warn(f'{error} -- selected now')
            self.set_solver(mesh_gemm=0)

    def _check_fp16_range(self, *_unused) -> None:
        """The default form of the float32 ICNN GEMMs runs on two fp16 planes (csrc/dpll_mesh_bf16.hpp): fp16 ends at 65504.  The
        weights are checked on the host when the form is chosen, at construction and after ``load_state_dict`` (one sync each);
        should they grow past the bound in between, the prep kernel turns the offending entries into NaN -- no item then has a
        valid solve, never a finite wrong number."""
        if self.dtype != torch.float32:
            return
        for geometry in self._meshes():
            net = geometry.network
            for name, bound in (('hidden_weights', 16384.0), ('input_weights', 4096.0)):
                for weight in getattr(net, name):
                    if not bool((weight.detach().abs() < bound).all()):
                        raise _capi.DpllError(f'the fp16-plane form of the ICNN GEMMs (mesh_gemm = 4, the default) needs |{name}| < '
                                              f'{bound:g}: set_solver(mesh_gemm=0) selects the f32 MFMA kernels')
            if not bool((net.output_weight.detach().abs() < 16384.0).all()):
                raise _capi.DpllError('the fp16-plane form of the ICNN GEMMs (mesh_gemm = 4, the default) needs |output_weight| < 16384: '
                                      'set_solver(mesh_gemm=0) selects the f32 MFMA kernels')

    def _check_input(self, tensor: Tensor, width: int, what: str, keep_graph: bool = False) -> Tensor:
        if tensor.shape[-1] != width:
            raise AssertionError(f'{what}: last dimension {tensor.shape[-1]} != {width}')
        if not tensor.is_cuda:
            raise _capi.DpllError(f'{what} must live on a ROCm device: the HIP kernels are the only implementation')
        if not keep_graph:
            tensor = tensor.detach()
        if tensor.dtype != self.dtype:
            tensor = tensor.to(self.dtype)
        tensor = tensor.reshape(-1, width)
        return tensor if tensor.stride(-1) == 1 and tensor.stride(0) >= width else tensor.contiguous()

    def _params_struct(self, flat: Tensor) -> _capi.Params:
        n_b, slots = self.spec.n_bodies, self._geom_slots()
        base, size = flat.data_ptr(), flat.element_size()
        u = self._u  # the actuation inputs of the call in flight (see _actuated), or None
        return _capi.Params(base, base + 10 * n_b * size, base + (10 * n_b + slots + 1) * size,
                            u.data_ptr() if u is not None else None, u.stride(0) if u is not None else 0)

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    # ---- ContactNets loss -----------------------------------------------------------------------
    def _launch_loss(self, x: Tensor, x_plus: Tensor, weights: Optional[Tensor], scale: float, want_grad: bool,
                     force: Optional[Tensor] = None, iters: Optional[Tensor] = None,
                     loss: Optional[Tensor] = None, want_loss: bool = True, fused_ar=None,
                     grad_out: Optional[Tensor] = None):
        """``grad_out``: a ``(1 + n_params,)`` buffer ``[loss total | gradient]`` to write instead of the shared one
        (whose slices are the parameters' ``.grad`` after a fused step)."""
        lib = _capi.library()
        flat = self._packed()
        batch = x.shape[0]
        if loss is None and want_loss:
            loss = torch.empty(batch, dtype=self.dtype, device=x.device)
        grad = total = workspace = None
        ws_bytes = 0
        if want_grad:
            if grad_out is not None:
                total, grad = grad_out[:1], grad_out[1:]
            else:
                if self._flat_grad is None or self._flat_grad.device != x.device:
                    self._alloc_grad_buffer(flat.numel(), x.device)
                grad, total = self._flat_grad, self._loss_total
            ws_bytes = lib.dpll_workspace_bytes(self._model(), batch) if self._mesh() is None else 0
            if self._workspace is None or self._workspace.numel() < ws_bytes or self._workspace.device != x.device:
                self._workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            workspace = self._workspace
        params = self._params_struct(flat)
        if self._mesh() is not None:
            workspace = self._mesh_workspace(batch, x.device)
            mesh = self._mesh_struct(flat)
            _capi.check(lib.dpll_contactnets_loss_mesh(
                self._model(), _DTYPES[self.dtype], ctypes.byref(params), mesh, _ptr(x), x.stride(0),
                _ptr(x_plus), x_plus.stride(0), batch, _ptr(weights), float(scale), _ptr(loss), _ptr(grad), _ptr(total),
                _ptr(force), _ptr(iters), _ptr(workspace), workspace.numel(), self._stream()))
            return loss, grad, total
        if fused_ar is not None and want_grad and loss is None and force is None and iters is None:
            _capi.check(lib.dpll_contactnets_loss_allreduce(
                self._model(), _DTYPES[self.dtype], ctypes.byref(params), _ptr(x), x.stride(0), _ptr(x_plus),
                x_plus.stride(0), batch, _ptr(weights), float(scale), _ptr(grad), _ptr(total), _ptr(workspace), ws_bytes,
                fused_ar, self._stream()))
            self._grad_reduced = True
            return loss, grad, total
        _capi.check(lib.dpll_contactnets_loss(self._model(), _DTYPES[self.dtype], ctypes.byref(params), _ptr(x),
                                              x.stride(0), _ptr(x_plus), x_plus.stride(0), batch, _ptr(weights),
                                              float(scale), _ptr(loss), _ptr(grad), _ptr(total), _ptr(force),
                                              _ptr(iters), _ptr(workspace), ws_bytes, self._stream()))
        return loss, grad, total

    def _actuation(self, u: Optional[Tensor], batch_shape) -> Optional[Tensor]:
        """The actuation inputs of a call as the kernels take them -- ``(B, n_u)`` rows on the device in the system's dtype -- or
        ``None``.  The reference's ``lagrangian_forces(q, v, u, inertia)`` carries ``B u`` (``multibody_terms.py:142-146,
        235-236``): a model with ``<transmission>`` elements has ``n_u`` actuators (general build, ``urdf.ModelSpec.actuators``).
        An input of width 0 means "no actuation" for any model (what ``sim_step`` passes, ``multibody_learnable_system.py:311``);
        a non-empty input for a model WITHOUT actuators, or of the wrong width, is refused rather than dropped."""
        n_u = self.spec.n_u
        if u is None or u.shape[-1] == 0:
            return None
        if u.shape[-1] != n_u:
            raise _capi.DpllError(f'actuation inputs of width {u.shape[-1]} for a model with n_u = {n_u} actuators'
                                  + (' (the model has no <transmission>: a silent drop of B u would be wrong dynamics)' if n_u == 0 else ''))
        if tuple(u.shape[:-1]) != tuple(batch_shape):
            raise _capi.DpllError(f'actuation inputs of batch shape {tuple(u.shape[:-1])} for states of batch shape {tuple(batch_shape)}')
        if not u.is_cuda:
            raise _capi.DpllError('actuation inputs must live on the ROCm device like the states')
        return u.detach().reshape(-1, n_u).to(self.dtype).contiguous()

    @contextlib.contextmanager
    def _actuated(self, u: Optional[Tensor]):
        """the actuation inputs of the launches made inside the block (``_params_struct`` hands them to the library)"""
        previous, self._u = self._u, u
        try:
            yield
        finally:
            self._u = previous

    def contactnets_loss(self, x: Tensor, u: Tensor, x_plus: Tensor, loss_pool=None) -> Tensor:
        """``(*, n_x), (*, ?), (*, n_x) -> (*,)`` ContactNets loss
        (``multibody_learnable_system.py:104-197``); differentiable with respect to the module's
        parameters (the backward pass re-runs the fused kernel with ``grad_output`` as weights)."""
        del loss_pool  # (pools are accepted and ignored)
        batch_shape = x.shape[:-1]
        uf = self._actuation(u, batch_shape)
        xf = self._check_input(x, self.space.n_x, 'x')
        xpf = self._check_input(x_plus, self.space.n_x, 'x_plus')
        self._packed()
        loss = _LossFunction.apply(self, xf, xpf, uf, *self._param_list())
        return loss.reshape(batch_shape)

    def contactnets_loss_and_grad(self, x: Tensor, x_plus: Tensor, accumulate: bool = False, u: Optional[Tensor] = None) -> Tensor:
        """Fused training step: mean loss over the batch AND its parameter gradients in one pass,
        i.e. what ``loss = system.contactnets_loss(x, u, x_plus).mean(); loss.backward()`` produces
        (``drake_experiment.py:202-224`` + ``experiment.py:355-359``).  Returns the mean loss as a
        one-element device tensor (no host sync) and writes ``.grad`` of every parameter.  ``u``: actuation inputs of an
        actuated model (:meth:`_actuation`)."""
        with self._actuated(self._actuation(u, x.shape[:-1])):
            return self._loss_and_grad(x, x_plus, accumulate)

    def _loss_and_grad(self, x: Tensor, x_plus: Tensor, accumulate: bool) -> Tensor:
        xf = self._check_input(x, self.space.n_x, 'x')
        xpf = self._check_input(x_plus, self.space.n_x, 'x_plus')
        denom = self.global_batch if self.global_batch > 0 else xf.shape[0] * self.grad_world
        if denom <= 0:
            raise _capi.DpllError('contactnets_loss_and_grad: empty batch (an empty SHARD of a data-parallel batch is fine '
                                  'once global_batch is set)')
        self._grad_reduced = False
        # accumulate: the old gradients may BE views of the buffer the launch overwrites -- take them out first
        old = [None if (not accumulate or p.grad is None) else p.grad.clone() for p in self._param_list()]
        _, grad, total = self._launch_loss(xf, xpf, None, 1.0 / denom, True, want_loss=False, fused_ar=self._fused_ar)
        if self._weld is not None and self._weld['grad'] is None:
            self._weld['grad'] = torch.empty_like(self.multibody_terms.lagrangian_terms.inertial_parameters.data)
        for param, piece, before in zip(self._param_list(), self._split_flat(grad, None if self._weld is None else self._weld['grad']), old):
            if before is not None:
                param.grad = before + piece
            elif param.grad is None or param.grad.data_ptr() != piece.data_ptr():
                param.grad = piece
        return total

    def contactnets_train_step(self, x: Tensor, x_plus: Tensor, adam: 'FusedAdamState') -> Tensor:
        """One whole training step -- the mean loss of the batch, its gradients, the gradient exchange when a process group
        is bound (peer transport) and the Adam update of every parameter, in place -- in the loss launch and ONE more
        kernel (``dpll_contactnets_train_step``; ``experiment.py:332-363``).  Returns the mean loss (one-element device
        tensor); ``.grad`` of every parameter holds the gradient the update used.  Every build takes it: the specialised box builds
        (with the gradient exchange of a process group inside the same kernel), the general and the forest build (Adam in the
        kernel that chains the folded rows; single process) and the specialised mesh builds (Adam where the network's weight
        gradients are reduced); not the general build with learned shapes."""
        if self._weld is not None:
            raise NotImplementedError('the fused training step updates the flat parameter buffer in the kernel; the inertial rows of '
                                      'welded links are not part of it: use contactnets_loss_and_grad + an optimizer')
        if self._mesh() is not None:
            return self._mesh_train_step(x, x_plus, adam)
        if not self.spec.is_fast() and self.grad_world > 1:
            raise NotImplementedError('the fused training step of the general / forest build is single process: use '
                                      'contactnets_loss_and_grad + GradientAllReduce + an optimizer')
        lib = _capi.library()
        xf = self._check_input(x, self.space.n_x, 'x')
        xpf = self._check_input(x_plus, self.space.n_x, 'x_plus')
        denom = self.global_batch if self.global_batch > 0 else xf.shape[0] * self.grad_world
        flat = self._packed()
        if self._flat_grad is None or self._flat_grad.device != xf.device:
            self._alloc_grad_buffer(flat.numel(), xf.device)
        ws_bytes = lib.dpll_workspace_bytes(self._model(), xf.shape[0])
        if self._workspace is None or self._workspace.numel() < ws_bytes or self._workspace.device != xf.device:
            self._workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=xf.device)
        adam.bind(flat)
        params = self._params_struct(flat)
        state = adam.struct(flat)
        _capi.check(lib.dpll_contactnets_train_step(
            self._model(), _DTYPES[self.dtype], ctypes.byref(params), _ptr(xf), xf.stride(0), _ptr(xpf), xpf.stride(0), xf.shape[0],
            None, 1.0 / denom, _ptr(self._flat_grad), _ptr(self._loss_total), _ptr(self._workspace), ws_bytes, self._fused_ar,
            ctypes.byref(state), self._stream()))
        self._grad_reduced = self._fused_ar is not None
        for param, piece in zip(self._param_list(), self._split_flat(self._flat_grad)):
            if param.grad is None or param.grad.data_ptr() != piece.data_ptr():
                param.grad = piece
        return self._loss_total

    def _mesh_train_step(self, x: Tensor, x_plus: Tensor, adam: 'FusedAdamState') -> Tensor:
        """:meth:`contactnets_train_step` of a system with learned shapes (the cube / elbow with a ``DeepSupportConvex`` per
        body, or a general tree with learned shapes among its geometries): ``dpll_contactnets_train_step_mesh``"""
        if self.grad_world > 1:
            raise NotImplementedError('the fused training step of a mesh system is single process')
        lib = _capi.library()
        xf = self._check_input(x, self.space.n_x, 'x')
        xpf = self._check_input(x_plus, self.space.n_x, 'x_plus')
        flat = self._packed()
        if self._flat_grad is None or self._flat_grad.device != xf.device:
            self._alloc_grad_buffer(flat.numel(), xf.device)
        workspace = self._mesh_workspace(xf.shape[0], xf.device)
        adam.bind(flat)
        params, mesh, state = self._params_struct(flat), self._mesh_struct(flat), adam.struct(flat)
        _capi.check(lib.dpll_contactnets_train_step_mesh(
            self._model(), _DTYPES[self.dtype], ctypes.byref(params), mesh, _ptr(xf), xf.stride(0), _ptr(xpf), xpf.stride(0), xf.shape[0],
            None, 1.0 / xf.shape[0], _ptr(self._flat_grad), _ptr(self._loss_total), _ptr(workspace), workspace.numel(), ctypes.byref(state),
            self._stream()))
        for param, piece in zip(self._param_list(), self._split_flat(self._flat_grad)):
            if param.grad is None or param.grad.data_ptr() != piece.data_ptr():
                param.grad = piece
        return self._loss_total

    def support_points(self, x: Tensor) -> Tensor:
        """``DeepSupportConvex.get_vertices`` for the ground-contact direction of every state:
        ``(*, n_x) -> (*, 4 n_bodies, 3)`` support points in the frames of their bodies (mesh systems only)."""
        if self._mesh() is None:
            raise TypeError('support_points needs a mesh (DeepSupportConvex) geometry')
        lib = _capi.library()
        batch_shape = x.shape[:-1]
        xf = self._check_input(x, self.space.n_x, 'x')
        flat = self._packed()
        workspace = self._mesh_workspace(xf.shape[0], xf.device)
        mesh = self._mesh_struct(flat)
        k = 4 * len(self._meshes()) if self.spec.is_fast() else 4 * _capi.GEN_SLOTS  # (general build: every contact slot)
        points = torch.zeros((xf.shape[0], k, 3), dtype=self.dtype, device=xf.device)
        _capi.check(lib.dpll_mesh_support_points(self._model(), _DTYPES[self.dtype], mesh, _ptr(xf),
                                                 xf.stride(0), xf.shape[0], _ptr(points), _ptr(workspace),
                                                 workspace.numel(), self._stream()))
        if not self.spec.is_fast():  # keep the four ground queries of every learned shape, in geometry order
            rows = [4 * g + s for g, (_, geom) in enumerate(self.spec.geoms()) if geom.kind == 'mesh' for s in range(4)]
            points = points[:, rows]
            k = len(rows)
        return points.reshape(batch_shape + (k, 3))

    def profile_loss_kernels(self, x: Tensor, x_plus: Tensor, reps: int = 100) -> Tuple[float, float]:
        """Average duration in ms of (loss kernel, finalize kernel) measured with HIP events on the
        launch stream (``dpll_profile_contactnets_loss``).  Synchronises."""
        lib = _capi.library()
        xf = self._check_input(x, self.space.n_x, 'x')
        xpf = self._check_input(x_plus, self.space.n_x, 'x_plus')
        self._launch_loss(xf, xpf, None, 1.0 / xf.shape[0], True, want_loss=False)  # allocates grad / workspace
        flat = self._packed()
        params = self._params_struct(flat)
        ms_loss, ms_fin = ctypes.c_float(0.0), ctypes.c_float(0.0)
        _capi.check(lib.dpll_profile_contactnets_loss(
            self._model(), _DTYPES[self.dtype], ctypes.byref(params), _ptr(xf), xf.stride(0), _ptr(xpf), xpf.stride(0),
            xf.shape[0], 1.0 / xf.shape[0], _ptr(self._flat_grad), _ptr(self._workspace), self._workspace.numel(),
            self._stream(), reps, ctypes.byref(ms_loss), ctypes.byref(ms_fin)))
        return ms_loss.value, ms_fin.value

    MESH_KERNELS = ('icnn_prep', 'icnn_fwd1', 'icnn_fwd2', 'loss_kernel', 'icnn_bwd1', 'icnn_bwd2', 'icnn_reduce')

    def profile_mesh_kernels(self, x: Tensor, x_plus: Tensor, reps: int = 50) -> Dict[str, float]:
        """Average duration in ms of each kernel of the mesh-geometry loss pipeline, HIP events on the launch
        stream (``dpll_profile_contactnets_loss_mesh``).  Synchronises."""
        lib = _capi.library()
        xf = self._check_input(x, self.space.n_x, 'x')
        xpf = self._check_input(x_plus, self.space.n_x, 'x_plus')
        self._launch_loss(xf, xpf, None, 1.0 / xf.shape[0], True, want_loss=False)  # allocates grad buffers
        flat = self._packed()
        params, mesh = self._params_struct(flat), self._mesh_struct(flat)
        workspace = self._mesh_workspace(xf.shape[0], xf.device)
        out = (ctypes.c_float * len(self.MESH_KERNELS))()
        _capi.check(lib.dpll_profile_contactnets_loss_mesh(
            self._model(), _DTYPES[self.dtype], ctypes.byref(params), mesh, _ptr(xf), xf.stride(0), _ptr(xpf),
            xpf.stride(0), xf.shape[0], 1.0 / xf.shape[0], _ptr(self._flat_grad), _ptr(workspace), workspace.numel(),
            self._stream(), reps, out))
        return dict(zip(self.MESH_KERNELS, (float(v) for v in out)))

    def contact_forces(self, x: Tensor, x_plus: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """Loss, the detached contact impulses ``(*, 3k)`` (``[normals | (t_x, t_y) per contact]``)
        and solver iteration counts; diagnostic twin of :meth:`contactnets_loss`."""
        batch_shape = x.shape[:-1]
        xf = self._check_input(x, self.space.n_x, 'x')
        xpf = self._check_input(x_plus, self.space.n_x, 'x_plus')
        force = torch.empty((xf.shape[0], 3 * self._n_contact_slots()), dtype=self.dtype, device=xf.device)
        iters = torch.empty(xf.shape[0], dtype=torch.int32, device=xf.device)
        loss, _, _ = self._launch_loss(xf, xpf, None, 1.0, False, force=force, iters=iters)
        force = force[:, self._contact_rows(force.device)]  # the kernels' contact slots -> the model's contacts
        return loss.reshape(batch_shape), force.reshape(batch_shape + (-1,)), iters.reshape(batch_shape)

    def _contact_rows(self, device) -> Tensor:
        """rows of the kernels' ``[normals (K) | (t_x, t_y) per contact (2 K)]`` slot layout that belong to the model's
        contacts (a sphere uses one of its geometry's four slots, unused geometry slots none), in the reference's
        order (multibody_terms.py:415-426)"""
        slots = self.spec.contact_slots()
        k_slots = self._n_contact_slots()
        rows = slots + [k_slots + 2 * s + t for s in slots for t in (0, 1)]
        return torch.tensor(rows, dtype=torch.long, device=device)

    # ---- dynamics -------------------------------------------------------------------------------
    def _step(self, x: Tensor) -> Tensor:
        lib = _capi.library()
        flat = self._packed()
        x_next = torch.empty((x.shape[0], self.space.n_x), dtype=self.dtype, device=x.device)
        params = self._params_struct(flat)
        if self._mesh() is not None:
            workspace = self._mesh_workspace(x.shape[0], x.device)
            mesh = self._mesh_struct(flat)
            _capi.check(lib.dpll_step_mesh(self._model(), _DTYPES[self.dtype], ctypes.byref(params), mesh,
                                           _ptr(x), x.stride(0), x.shape[0], _ptr(x_next), x_next.stride(0),
                                           _ptr(workspace), workspace.numel(), self._stream()))
            return x_next
        _capi.check(lib.dpll_step(self._model(), _DTYPES[self.dtype], ctypes.byref(params), _ptr(x), x.stride(0),
                                  x.shape[0], _ptr(x_next), x_next.stride(0), None, self._stream()))
        return x_next

    def _step_backward(self, x: Tensor, grad_x_next: Tensor, want_state: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
        lib = _capi.library()
        flat = self._packed()
        grad = torch.empty(flat.numel(), dtype=self.dtype, device=x.device)
        params = self._params_struct(flat)
        gx = grad_x_next.to(self.dtype)
        grad_x = torch.empty_like(x, memory_format=torch.contiguous_format) if want_state else None
        ld_gx = grad_x.stride(0) if want_state else 0
        if self._mesh() is not None:
            workspace = self._mesh_workspace(x.shape[0], x.device)
            mesh = self._mesh_struct(flat)
            _capi.check(lib.dpll_step_backward_mesh(
                self._model(), _DTYPES[self.dtype], ctypes.byref(params), mesh, _ptr(x), x.stride(0), _ptr(gx),
                gx.stride(0), x.shape[0], _ptr(grad), _ptr(grad_x), ld_gx, _ptr(workspace), workspace.numel(), self._stream()))
            return grad, grad_x
        ws_bytes = lib.dpll_workspace_bytes(self._model(), x.shape[0])
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        _capi.check(lib.dpll_step_backward(self._model(), _DTYPES[self.dtype], ctypes.byref(params), _ptr(x), x.stride(0),
                                           _ptr(gx), gx.stride(0), x.shape[0], _ptr(grad), _ptr(grad_x), ld_gx,
                                           _ptr(workspace), ws_bytes, self._stream()))
        return grad, grad_x

    def _wants_graph(self, x: Tensor) -> bool:
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._param_list()))

    def _differentiable_step(self, x: Tensor, u: Optional[Tensor] = None) -> Tensor:
        self._packed()
        if self._wants_graph(x):
            return _StepFunction.apply(self, x, u, *self._param_list())
        with self._actuated(u):
            return self._step(x.detach())

    def forward_dynamics(self, q: Tensor, v: Tensor, u: Tensor, dynamics_pool=None) -> Tensor:
        """``(*, n_q), (*, n_v), (*, ?) -> (*, n_v)`` next velocity by Anitescu's convex contact
        model (``multibody_learnable_system.py:199-304``).  Differentiable with respect to the module's
        parameters and to ``q, v`` (implicit differentiation of the cone solve, ``dpll_step_backward``)."""
        del dynamics_pool
        batch_shape = q.shape[:-1]
        uf = self._actuation(u, batch_shape)
        x = self._check_input(torch.cat((q, v), -1), self.space.n_x, 'state', keep_graph=True)
        return self._differentiable_step(x, uf)[:, self.space.n_q:].reshape(batch_shape + (self.space.n_v,))

    def sim_step(self, x: Tensor, carry: Tensor) -> Tuple[Tensor, Tensor]:
        """``Integrator.partial_step`` callback (``multibody_learnable_system.py:306-313``)."""
        q, v = self.space.q_v(x)
        return self.forward_dynamics(q, v, torch.zeros(q.shape[:-1] + (0,), device=q.device)), carry

    def step(self, x: Tensor) -> Tensor:
        """One fused ``VelocityIntegrator.step``: ``(*, n_x) -> (*, n_x)``."""
        batch_shape = x.shape[:-1]
        return self._differentiable_step(self._check_input(x, self.space.n_x, 'x', keep_graph=True)).reshape(
            batch_shape + (self.space.n_x,))

    def _fused_simulate(self, x_0: Tensor, steps: int) -> Tensor:
        lib = _capi.library()
        batch_shape = x_0.shape[:-1]
        x = self._check_input(x_0, self.space.n_x, 'x_0', keep_graph=True)
        flat = self._packed()
        if steps >= 1 and self._wants_graph(x):
            # prediction losses (experiment.py:292-320): one autograd node per step, parameter gradient and state
            # adjoint from dpll_step_backward (back-propagation through time); the fused rollout kernel below is
            # the inference path
            states = [x]
            for _ in range(steps):
                states.append(self._differentiable_step(states[-1]))
            return torch.stack(states, dim=1).reshape(batch_shape + (steps + 1, self.space.n_x))
        x = x.detach()
        traj = torch.empty((x.shape[0], steps + 1, self.space.n_x), dtype=self.dtype, device=x.device)
        if self._mesh() is not None:  # support points depend on the state: the networks' forward kernels run every step
            if steps == 0:
                traj[:, 0] = x
            else:
                params = self._params_struct(flat)
                workspace = self._mesh_workspace(x.shape[0], x.device)
                mesh = self._mesh_struct(flat)
                _capi.check(lib.dpll_simulate_mesh(self._model(), _DTYPES[self.dtype], ctypes.byref(params), mesh, _ptr(x),
                                                   x.stride(0), x.shape[0], steps, _ptr(traj), _ptr(workspace),
                                                   workspace.numel(), self._stream()))
            return traj.reshape(batch_shape + (steps + 1, self.space.n_x))
        params = self._params_struct(flat)
        _capi.check(lib.dpll_simulate(self._model(), _DTYPES[self.dtype], ctypes.byref(params), _ptr(x), x.stride(0),
                                      x.shape[0], steps, _ptr(traj), self._stream()))
        return traj.reshape(batch_shape + (steps + 1, self.space.n_x))

    def preprocess_initial_condition(self, x_0: Tensor, carry_0: Tensor) -> Tuple[Tensor, Tensor]:
        """``dair_pll/system.py:147-173``: keep the last state of the initial sequence."""
        assert len(x_0.shape) >= 2
        assert x_0.shape[-1] == self.space.n_x
        return x_0[..., -1, :], carry_0

    def simulate(self, x_0: Tensor, carry_0: Tensor, steps: int = 1) -> Tuple[Tensor, Tensor]:
        """``(*, T_0, n_x) -> (*, steps + 1, n_x)`` (``dair_pll/system.py:97-129``).  Extra batch
        dimensions are flattened into the kernel's batch instead of being iterated in Python."""
        x, carry = self.preprocess_initial_condition(x_0, carry_0)
        return self.integrator.simulate(x, carry, steps)

    # ---- reporting (host side, off the hot path) -------------------------------------------------
    def scalars(self) -> Dict[str, float]:
        """Scalar description of the learned parameters with the reference's key scheme
        (``dair_pll/multibody_terms.py:536-582``: ``{body}_{m, com_*, I_*}`` from ``pi_cm``
        (``inertia.py:444-458``), ``{body}_len_*`` full box lengths (``geometry.py:405-411``),
        ``{body}_mu``); mesh extraction of learned shapes is out of scope."""
        from .inertia import theta_to_pi_cm
        theta = self.multibody_terms.lagrangian_terms.inertial_parameters.detach().double().cpu().numpy()
        friction = self.multibody_terms.contact_terms.get_friction_coefficients().detach().cpu()
        out: Dict[str, float] = {}
        labels = self._body_labels()
        for index, label in enumerate(self._row_labels()):  # one set per Drake body: a row of inertial_parameters
            pi_cm = theta_to_pi_cm(theta[index])
            out[f'{label}_m'] = float(pi_cm[0])
            for axis, value in zip('xyz', pi_cm[1:4] / pi_cm[0]):
                out[f'{label}_com_{axis}'] = float(value)
            for name, value in zip(('I_xx', 'I_yy', 'I_zz', 'I_xy', 'I_xz', 'I_yz'), pi_cm[4:]):
                out[f'{label}_{name}'] = float(value)
        for index, body in enumerate(self.spec.bodies):
            label = labels[index]
            mine = [g for g, (b, _) in enumerate(self.spec.geoms()) if b == index]
            for count, g in enumerate(mine):
                # one geometry per body is all the reference's key scheme distinguishes; further ones get a suffix
                prefix = label if count == 0 else f'{label}_g{count}'
                geometry = self.multibody_terms.contact_terms.geometries[g + 1]
                if isinstance(geometry, Box):
                    for axis, value in zip('xyz', geometry.get_half_lengths().detach().cpu().reshape(-1)):
                        out[f'{prefix}_len_{axis}'] = 2 * float(value)
                elif isinstance(geometry, Sphere):
                    out[f'{prefix}_radius'] = float(geometry.get_radius().detach())
                elif isinstance(geometry, Polygon):
                    out.update({f'{prefix}_{key}': value for key, value in geometry.scalars().items()})
                out[f'{prefix}_mu'] = float(friction[g + 1])
        return out

    def _body_labels(self) -> List[str]:
        """a body's name in the scalar summaries: the link name, prefixed by its model's name when the system has several models
        (the reference's unique_body_identifier, ``drake_utils.py:123-126``: ``{model instance}_{body}``)"""
        if not self.forest or len(self.spec.models) == 1:
            return [body.name for body in self.spec.bodies]
        return [f'{self.spec.names[m]}_{body.name}' for m, body in zip(self.spec.body_model(), self.spec.bodies)]

    def _row_labels(self) -> List[str]:
        """the names of the rows of ``inertial_parameters`` (:meth:`_body_labels` unless links are welded on)"""
        if self._weld is None:
            return self._body_labels()
        rows = self.spec.inertia_rows()
        if not self.forest or len(self.spec.models) == 1:
            return [row.name for row in rows]
        model = self.spec.body_model()
        return [f'{self.spec.names[model[row.body]]}_{row.name}' for row in rows]

    def _pi_cm(self) -> np.ndarray:
        from .inertia import theta_to_pi_cm
        theta = self.multibody_terms.lagrangian_terms.inertial_parameters.detach().double().cpu().numpy()
        return np.stack([theta_to_pi_cm(row) for row in theta])

    def extract_meshes(self) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
        """``{body name: (vertices (V, 3), faces (F, 3))}`` for every body whose geometry is a learned convex
        shape (``multibody_terms.py:561-563`` with ``deep_support_function.py:93-123``)."""
        from . import export
        meshes = {}
        for g, (index, _) in enumerate(self.spec.geoms()):
            body = self.spec.bodies[index]
            geometry = self.multibody_terms.contact_terms.geometries[g + 1]
            if isinstance(geometry, DeepSupportConvex):
                # a geometry turned in its body (rpy on the collision <origin>): the kernels query the network along
                # -(row 2 of R_WB R_BG), so the BODY-frame direction that makes the query d is R_BG d; the support points come
                # back in the geometry frame, which is the frame the exported mesh lives in (the URDF keeps the origin)
                r_bg = torch.tensor(self.spec.geoms()[g][1].rotation, dtype=torch.float64)
                k_mesh = sum(1 for _, other in self.spec.geoms()[:g] if other.kind == 'mesh')  # rows of support_points

                def support(directions: np.ndarray, geometry=geometry, k_mesh=k_mesh, r_bg=r_bg) -> np.ndarray:
                    # the HIP kernels evaluate the network (dpll_mesh_support_points): a state whose rotation takes
                    # the direction to -e_z makes the kernel's first query (perturbation row 0 = 0) that direction
                    param = geometry.network.output_weight
                    d = torch.tensor(directions, dtype=torch.float64) @ r_bg.t()
                    w = 1.0 - d[:, 2]                                    # 1 + d . (-e_z)
                    axis = torch.stack((-d[:, 1], d[:, 0], torch.zeros_like(w)), -1)  # d x (-e_z)
                    flip = w < 1e-12                                     # d = +e_z: half turn about x
                    quat = torch.cat((w[:, None], axis), -1)
                    quat[flip] = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64)
                    quat = quat / quat.norm(dim=-1, keepdim=True)
                    x = torch.zeros((d.shape[0], self.space.n_x), dtype=torch.float64)
                    x[:, :4] = quat
                    with torch.no_grad():  # joint angles zero: every body has the base's rotation
                        points = self.support_points(x.to(device=param.device, dtype=param.dtype))
                    return points[:, 4 * k_mesh, :].double().cpu().numpy()
                meshes[body.name] = export.extract_mesh(support)
        return meshes

    def scalars_and_meshes(self) -> Tuple[Dict[str, float], Dict[str, Tuple[np.ndarray, np.ndarray]]]:
        """``MultibodyTerms.scalars_and_meshes`` (``multibody_terms.py:536-582``): :meth:`scalars` plus, for a
        learned convex shape, its mesh and the ``{body}_diameter_*`` / ``{body}_center_*`` of the vertices."""
        scalars = self.scalars()
        meshes = self.extract_meshes()
        for name, (vertices, _) in meshes.items():
            low, high = vertices.min(axis=0), vertices.max(axis=0)
            for axis, diameter, centre in zip('xyz', high - low, low + (high - low) / 2):
                scalars[f'{name}_diameter_{axis}'] = float(diameter)
                scalars[f'{name}_center_{axis}'] = float(centre)
        return scalars, meshes

    def generate_updated_urdfs(self) -> Dict[str, str]:
        """Writes the current parameters as URDFs with the original base names into ``output_urdfs_dir`` and
        returns ``{urdf name: path}`` (``multibody_learnable_system.py:82-102``, ``urdf_utils.py:317-384``);
        a learned convex shape is written next to them as ``test.obj`` (``urdf_utils.py:244-252``)."""
        from . import export
        assert self.output_urdfs_dir is not None
        pi_cm = self._pi_cm()
        friction = self.multibody_terms.contact_terms.get_friction_coefficients().detach().double().cpu().numpy()
        meshes = self.extract_meshes()
        bodies = []
        body_model = self.spec.body_model() if self.forest else [0] * self.spec.n_bodies
        # one entry per row of inertial_parameters = per URDF link that carries mass (a link welded on by a `fixed` joint keeps
        # its own <inertial> and its own <collision> elements: GeomSpec.link)
        rows = self.spec.inertia_rows()
        row_model = [body_model[row.body] for row in rows]
        for index, row in enumerate(rows):
            body = self.spec.bodies[row.body]
            shapes = []  # this link's <collision> elements in order
            for g, (b, geom) in enumerate(self.spec.geoms()):
                if b != row.body or (geom.link or body.name) != row.name:
                    continue
                geometry = self.multibody_terms.contact_terms.geometries[g + 1]
                if isinstance(geometry, Box):
                    half = geometry.get_half_lengths().detach().double().cpu().numpy().reshape(-1)
                    shape = ('box', {'size': ' '.join(repr(2.0 * float(h)) for h in half)})
                elif isinstance(geometry, Sphere):
                    shape = ('sphere', {'radius': repr(float(geometry.get_radius().detach()))})
                elif isinstance(geometry, Polygon):
                    raise NotImplementedError('Polygon URDF representation not yet implemented')  # urdf_utils.py:224-228
                else:
                    # the reference writes every learned shape to 'test.obj' (urdf_utils.py:244-252): with one mesh per
                    # body of a multi-body system that would overwrite, so those get a file per body
                    obj_name = export.MESH_FILE if len(meshes) == 1 else f'{body.name}.obj'
                    export.save_string(os.path.join(self.output_urdfs_dir, obj_name), export.mesh_to_obj(*meshes[body.name]))
                    shape = ('mesh', {'filename': obj_name})
                shapes.append((shape, float(friction[g + 1])))
            bodies.append((row.name, pi_cm[index], shapes))
        new_urdfs = {}
        for model, (name, source) in enumerate(self.urdfs.items()):
            # (several models may come from ONE file -- two cubes -- so a model's file carries its name when there are several)
            base = os.path.basename(source) if len(self.urdfs) == 1 else f'{name}.urdf'
            target = os.path.join(self.output_urdfs_dir, base)
            mine = [entry for entry, owner in zip(bodies, row_model) if owner == model]
            export.save_string(target, export.render_urdf(source, mine))
            new_urdfs[name] = target
        return new_urdfs

    # ---- terms ----------------------------------------------------------------------------------
    def _terms(self, q: Tensor, v: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
        lib = _capi.library()
        batch_shape = q.shape[:-1]
        x = self._check_input(torch.cat((q, v), -1), self.space.n_x, 'state')
        flat = self._packed()
        n, n_v, k = x.shape[0], self.space.n_v, self._n_contact_slots()
        new = lambda *shape: torch.empty((n,) + shape, dtype=self.dtype, device=x.device)
        delassus, mass, jac, phi, acc = new(3 * k, 3 * k), new(n_v, n_v), new(3 * k, n_v), new(k), new(n_v)
        params = self._params_struct(flat)
        if self._mesh() is not None:
            workspace = self._mesh_workspace(n, x.device)
            mesh = self._mesh_struct(flat)
            _capi.check(lib.dpll_terms_mesh(self._model(), _DTYPES[self.dtype], ctypes.byref(params), mesh,
                                            _ptr(x), x.stride(0), n, _ptr(delassus), _ptr(mass), _ptr(jac), _ptr(phi),
                                            _ptr(acc), _ptr(workspace), workspace.numel(), self._stream()))
        else:
            _capi.check(lib.dpll_terms(self._model(), _DTYPES[self.dtype], ctypes.byref(params), _ptr(x), x.stride(0), n,
                                       _ptr(delassus), _ptr(mass), _ptr(jac), _ptr(phi), _ptr(acc), self._stream()))
        if k != self.spec.n_contacts:  # keep the model's contacts of the kernels' slots
            rows = self._contact_rows(x.device)
            delassus = delassus[:, rows][:, :, rows]
            jac = jac[:, rows]
            phi = phi[:, torch.tensor(self.spec.contact_slots(), dtype=torch.long, device=x.device)]
        shape = lambda t: t.reshape(batch_shape + t.shape[1:])
        return shape(delassus), shape(mass), shape(jac), shape(phi), shape(acc)
