"""dair_pll_amd -- MI355X-native contact-dynamics hot path behind dair_pll's System / Integrator API.

Host side (this package): URDF -> model description, parameter plumbing, the reference's method
surface.  Device side: ``csrc/libdpll_hip.so`` (hand-written gfx950 kernels behind the C ABI of
``include/dpll.h``).  Nothing here computes on the CPU and nothing imports ``oracle/``.
"""
from .integrator import Integrator, VelocityIntegrator
from .state_space import FloatingBaseSpace
from .system import Box, DeepSupportConvex, HomogeneousICNN, MultibodyLearnableSystem, MultibodyTerms, Plane
from .urdf import ModelSpec, parse_urdf

__all__ = ['MultibodyLearnableSystem', 'MultibodyTerms', 'Integrator', 'VelocityIntegrator', 'FloatingBaseSpace',
           'Box', 'Plane', 'DeepSupportConvex', 'HomogeneousICNN', 'ModelSpec', 'parse_urdf']
