"""Flow library for the hot path's configurations ("next" rows F1/F2 of SURVEY.md 8(f)):
Taylor-Green vortex (2-D/3-D), flow around an obstacle, doubly periodic shear layer (2-D as in
the reference, plus the 3-D variant BASELINE.json's cfg5 asks for).

API of lettuce/ext/_flows/_ext_flow.py:8-42, taylorgreen.py:16-122, obstacle.py:54-151,
doublyshear.py:19-80.  The other demo flows (Couette, Poiseuille, cavity, decaying turbulence)
are out of scope.
"""
import warnings
from abc import ABC, abstractmethod
from typing import List, Optional, Union

import numpy as np
import torch

from .._flow import Flow
from .._stencil import D1Q3, D2Q9, D3Q19
from .._unit import UnitConversion
from ..util import append_axes
from ._boundary import AntiBounceBackOutlet, BounceBackBoundary, EquilibriumBoundaryPU
from ._equilibrium import QuadraticEquilibrium

__all__ = ["ExtFlow", "TaylorGreenVortex", "TaylorGreenVortex2D", "TaylorGreenVortex3D",
           "Obstacle", "DoublyPeriodicShear2D", "DoublyPeriodicShear3D", "flow_by_name"]


class ExtFlow(Flow, ABC):
    """Common constructor: resolution / units are made by the subclass, the stencil defaults by
    dimension (D1Q3, D2Q9, D3Q19), the equilibrium to QuadraticEquilibrium."""

    def __init__(self, context: "Context", resolution: Union[int, List[int]], reynolds_number,
                 mach_number, stencil: Optional["Stencil"] = None,
                 equilibrium: Optional["Equilibrium"] = None):
        resolution = self.make_resolution(resolution, stencil)
        assert len(resolution) in [1, 2, 3], \
            f"flow supports dimensions 1, 2 and 3 but {len(resolution)} dimensions where requested."
        if not stencil:
            stencil = (D1Q3, D2Q9, D3Q19)[len(resolution) - 1]
        if callable(stencil):
            stencil = stencil()
        Flow.__init__(self, context, resolution,
                      self.make_units(reynolds_number, mach_number, resolution), stencil,
                      equilibrium or QuadraticEquilibrium())

    @abstractmethod
    def make_resolution(self, resolution: Union[int, List[int]],
                        stencil: Optional["Stencil"] = None) -> List[int]:
        ...

    @abstractmethod
    def make_units(self, reynolds_number, mach_number, resolution: List[int]) -> "UnitConversion":
        ...


def _periodic_axes(resolution, length, context, slab=None):
    """x_i = length * i / n, built with linspace in the context dtype exactly as the reference
    does (taylorgreen.py:52-61, doublyshear.py:69-77).

    With ``slab`` (a ``ZSlab``) the last axis holds this rank's planes of the GLOBAL z axis
    (plus the slab's halo planes, periodic), so a flow built on ``slab.extended_resolution``
    gets, plane for plane, the values of the global initial condition."""
    if slab is None:
        return [torch.linspace(0, length * (1 - 1 / n), steps=n, device=context.device,
                               dtype=context.dtype) for n in resolution]
    assert list(resolution) == slab.extended_resolution, "flow resolution != extended slab"
    axes = [torch.linspace(0, length * (1 - 1 / n), steps=n, device=context.device,
                           dtype=context.dtype) for n in slab.global_resolution]
    axes[2] = axes[2][slab.z_indices(device=context.device)]
    return axes


class TaylorGreenVortex(ExtFlow):
    def __init__(self, context: "Context", resolution: Union[int, List[int]], reynolds_number,
                 mach_number, stencil: Optional["Stencil"] = None,
                 equilibrium: Optional["Equilibrium"] = None, initialize_fneq: bool = True,
                 slab: Optional["ZSlab"] = None):
        self.initialize_fneq = initialize_fneq
        self.slab = slab          # multi-GPU extension, not in the reference (see _slab.py)
        if stencil is None and not isinstance(resolution, list):
            warnings.warn("Requiring information about dimensionality! Either via stencil or "
                          "resolution. Setting dimension to 2.", UserWarning)
            self.stencil = D2Q9()
        else:
            self.stencil = stencil() if callable(stencil) else stencil
        ExtFlow.__init__(self, context, resolution, reynolds_number, mach_number, stencil,
                         equilibrium)

    def make_resolution(self, resolution, stencil=None) -> List[int]:
        if isinstance(resolution, int):
            return [resolution] * self.stencil.d
        assert len(resolution) in [2, 3], \
            "the resolution of a taylor-green-vortex must be 2- or 3-dimensional!"
        return resolution

    def make_units(self, reynolds_number, mach_number, resolution) -> "UnitConversion":
        return UnitConversion(reynolds_number=reynolds_number, mach_number=mach_number,
                              characteristic_length_lu=resolution[0],
                              characteristic_length_pu=2 * torch.pi,
                              characteristic_velocity_pu=1)

    @property
    def grid(self):
        return torch.meshgrid(*_periodic_axes(self.resolution, 2 * torch.pi, self.context,
                                              getattr(self, "slab", None)), indexing="ij")

    def initial_pu(self) -> (torch.Tensor, torch.Tensor):
        return self.analytic_solution(t=0)

    def analytic_solution(self, t: float) -> (torch.Tensor, torch.Tensor):
        if t > 0 and self.stencil.d > 2:
            warnings.warn("The analytic solution is only true for the 2D TGV!")
        # The reference evaluates sin / cos on the meshgrid (taylorgreen.py:69-95), i.e. on whole fields;
        # the functions are elementwise, so evaluating them on the 1-D axes and broadcasting the products
        # gives the same values with d + 1 field-sized results instead of a dozen temporaries
        # (a 512 x 512 x 70 slab is initialised per rank)
        axes = _periodic_axes(self.resolution, 2 * torch.pi, self.context, getattr(self, "slab", None))
        d = len(axes)
        g = [a.reshape([-1 if k == i else 1 for k in range(d)]) for i, a in enumerate(axes)]
        nu = self.context.convert_to_tensor(self.units.viscosity_pu)
        if len(self.resolution) == 2:
            decay_u, decay_p = torch.exp(-2 * nu * t), torch.exp(-4 * nu * t)
            u = torch.stack([torch.cos(g[0]) * torch.sin(g[1]) * decay_u,
                             -torch.sin(g[0]) * torch.cos(g[1]) * decay_u])
            p = -torch.stack([0.25 * (torch.cos(2 * g[0]) + torch.cos(2 * g[1])) * decay_p])
        else:
            u = torch.stack([torch.sin(g[0]) * torch.cos(g[1]) * torch.cos(g[2]),
                             -torch.cos(g[0]) * torch.sin(g[1]) * torch.cos(g[2]),
                             torch.zeros(self.resolution, dtype=self.context.dtype, device=self.context.device)])
            p = torch.stack([1 / 16. * (torch.cos(2 * g[0]) + torch.cos(2 * g[1]))
                             * (torch.cos(2 * g[2]) + 2)])
        return p, u

    @property
    def boundaries(self) -> List["Boundary"]:
        return []


def _deprecated_tgv(name):
    def factory(context, resolution, reynolds_number, mach_number, stencil=None, equilibrium=None):
        warnings.warn(f"{name} is deprecated. Use TaylorGreenVortex instead", DeprecationWarning)
        return TaylorGreenVortex(context=context, resolution=resolution,
                                 reynolds_number=reynolds_number, mach_number=mach_number,
                                 stencil=stencil, equilibrium=equilibrium)
    factory.__name__ = name
    return factory


TaylorGreenVortex2D = _deprecated_tgv("TaylorGreenVortex2D")
TaylorGreenVortex3D = _deprecated_tgv("TaylorGreenVortex3D")


class Obstacle(ExtFlow):
    """Flow in +x around a solid ``mask``: equilibrium inlet at x = 0, anti-bounce-back outlet
    at x = L, bounce-back on the mask (lettuce/ext/_flows/obstacle.py:16-125).

    >>> flow = Obstacle(context, [101, 51], 100, 0.1, domain_length_x=10.1, stencil=D2Q9)
    >>> x, y = flow.grid
    >>> flow.mask = ((x - 2.5) ** 2 + (y - 2.5) ** 2) < 1.
    """

    def __init__(self, context: "Context", resolution: Union[int, List[int]], reynolds_number,
                 mach_number, domain_length_x, char_length=1, char_velocity=1,
                 stencil: Optional["Stencil"] = None, equilibrium: Optional["Equilibrium"] = None,
                 slab: Optional["ZSlab"] = None):
        self.slab = slab          # multi-GPU extension, not in the reference (see _slab.py)
        self.char_length_lu = resolution[0] / domain_length_x * char_length
        self.char_length = char_length
        self.char_velocity = char_velocity
        self.resolution = self.make_resolution(resolution, stencil)
        self._mask = torch.zeros(self.resolution, dtype=torch.bool)
        ExtFlow.__init__(self, context, resolution, reynolds_number, mach_number, stencil,
                         equilibrium)

    def make_units(self, reynolds_number, mach_number, resolution: List[int]) -> "UnitConversion":
        return UnitConversion(reynolds_number=reynolds_number, mach_number=mach_number,
                              characteristic_length_lu=self.char_length_lu,
                              characteristic_length_pu=self.char_length,
                              characteristic_velocity_pu=self.char_velocity)

    def make_resolution(self, resolution, stencil=None) -> List[int]:
        if isinstance(resolution, int):
            return [resolution] * (stencil.d or self.stencil.d)
        return resolution

    @property
    def mask(self):
        return self._mask

    @mask.setter
    def mask(self, m):
        assert isinstance(m, (np.ndarray, torch.Tensor)) and all(
            m.shape[dim] == self.resolution[dim] for dim in range(self.stencil.d))
        self._mask = self.context.convert_to_tensor(m, dtype=torch.bool)

    def initial_pu(self) -> (float, Union[np.array, torch.Tensor]):
        p = np.zeros_like(self.grid[0], dtype=float)[None, ...]
        u_char = append_axes(self.units.characteristic_velocity_pu * self._unit_vector(),
                             self.stencil.d)
        return p, ~self.mask.to(u_char.device) * u_char

    @property
    def grid(self):
        index = [torch.arange(n) for n in self.resolution]
        if getattr(self, "slab", None) is not None:      # global z index of this rank's planes
            index[2] = self.slab.z_indices()
        axes = [self.units.convert_length_to_pu(i) for i in index]
        return torch.meshgrid(*axes, indexing="ij")

    @property
    def boundaries(self):
        x = self.grid[0]
        inlet_velocity = self.units.characteristic_velocity_pu * self._unit_vector()
        return [EquilibriumBoundaryPU(context=self.context, mask=torch.abs(x) < 1e-6,
                                      velocity=inlet_velocity),
                AntiBounceBackOutlet(self._unit_vector().tolist(), self),
                BounceBackBoundary(self.mask)]

    def _unit_vector(self, i=0):
        return torch.eye(self.stencil.d)[i]


def _deprecated_obstacle(name):
    def factory(context, resolution, reynolds_number, mach_number, stencil, char_length_lu):
        warnings.warn(f"{name} is deprecated. Use Obstacle instead", DeprecationWarning)
        nx = resolution[0] if isinstance(resolution, list) else resolution
        return Obstacle(context=context, resolution=resolution, reynolds_number=reynolds_number,
                        mach_number=mach_number, domain_length_x=nx / char_length_lu, stencil=stencil)
    factory.__name__ = name
    return factory


# present in the reference's module but not exported (obstacle.py:13,128-151)
Obstacle2D = _deprecated_obstacle("Obstacle2D")
Obstacle3D = _deprecated_obstacle("Obstacle3D")


def _shear_axes(flow):
    return torch.meshgrid(*_periodic_axes(flow.resolution, 1, flow.context,
                                          getattr(flow, "slab", None)), indexing="ij")


class DoublyPeriodicShear2D(ExtFlow):
    """Doubly periodic shear layer, 2-D (lettuce/ext/_flows/doublyshear.py:19-80).

    The reference's ``initial_pu`` has its two ``torch.where`` branches the wrong way round, so
    that tanh(80 * (+-0.25..0.75)) saturates and u_x == 1 everywhere.  For
    drop-in parity this class reproduces the reference's expression as it is written;
    ``DoublyPeriodicShear3D`` below uses the textbook profile."""

    def __init__(self, context: "Context", resolution: Union[int, List[int]], reynolds_number,
                 mach_number, stencil: Optional["Stencil"] = None,
                 equilibrium: Optional["Equilibrium"] = None, shear_layer_width=80,
                 initial_perturbation_magnitude=0.05, initialize_fneq: bool = True):
        self.initialize_fneq = initialize_fneq
        self.initial_perturbation_magnitude = initial_perturbation_magnitude
        self.shear_layer_width = shear_layer_width
        self.stencil = D2Q9() if stencil is None else (stencil() if callable(stencil) else stencil)
        super().__init__(context, resolution, reynolds_number, mach_number, self.stencil,
                         equilibrium)

    def make_resolution(self, resolution, stencil=None) -> List[int]:
        if isinstance(resolution, int):
            return [resolution] * self.stencil.d
        assert len(resolution) == 2, "expected 2-dimensional resolution"
        return resolution

    def make_units(self, reynolds_number, mach_number, resolution) -> "UnitConversion":
        return UnitConversion(reynolds_number=reynolds_number, mach_number=mach_number,
                              characteristic_length_lu=resolution[0], characteristic_length_pu=1,
                              characteristic_velocity_pu=1)

    def analytic_solution(self, t=0):
        raise NotImplementedError

    def initial_pu(self):
        w, delta = self.shear_layer_width, self.initial_perturbation_magnitude
        x, y = self.grid
        ux = self.context.convert_to_tensor(torch.where(y > 0.5, torch.tanh(w * (y - 0.25)),
                                                        torch.tanh(w * (0.75 - y))))
        uy = delta * torch.sin(2 * torch.pi * (x + 0.25))
        return torch.zeros_like(ux)[None, ...], torch.stack([ux, uy])

    @property
    def grid(self):
        return _shear_axes(self)

    @property
    def boundaries(self):
        return []


class DoublyPeriodicShear3D(ExtFlow):
    """Periodic shear layer extruded along z -- BASELINE.json cfg5.  The reference has no 3-D
    shear flow; this is the textbook profile:
    u_x = tanh(w (y - 1/4)) for y <= 1/2, tanh(w (3/4 - y)) otherwise,
    u_y = delta sin(2 pi (x + 1/4)), u_z = 0, p = 0 on the grid i / n.  Step parity of this
    flow is pinned by running the same initial field through the reference Simulation
    (tests/golden/shear3d_*)."""

    def __init__(self, context: "Context", resolution: Union[int, List[int]], reynolds_number,
                 mach_number, stencil: Optional["Stencil"] = None,
                 equilibrium: Optional["Equilibrium"] = None, shear_layer_width=80,
                 initial_perturbation_magnitude=0.05, initialize_fneq: bool = False,
                 slab: Optional["ZSlab"] = None):
        self.initialize_fneq = initialize_fneq
        self.slab = slab
        self.initial_perturbation_magnitude = initial_perturbation_magnitude
        self.shear_layer_width = shear_layer_width
        ExtFlow.__init__(self, context, resolution, reynolds_number, mach_number,
                         stencil or D3Q19, equilibrium)

    def make_resolution(self, resolution, stencil=None) -> List[int]:
        if isinstance(resolution, int):
            return [resolution] * 3
        assert len(resolution) == 3, "expected 3-dimensional resolution"
        return resolution

    def make_units(self, reynolds_number, mach_number, resolution) -> "UnitConversion":
        return UnitConversion(reynolds_number=reynolds_number, mach_number=mach_number,
                              characteristic_length_lu=resolution[0], characteristic_length_pu=1,
                              characteristic_velocity_pu=1)

    def initial_pu(self):
        w, delta = self.shear_layer_width, self.initial_perturbation_magnitude
        x, y, _ = self.grid
        ux = torch.where(y <= 0.5, torch.tanh(w * (y - 0.25)), torch.tanh(w * (0.75 - y)))
        uy = delta * torch.sin(2 * torch.pi * (x + 0.25))
        return torch.zeros_like(ux)[None, ...], torch.stack([ux, uy, torch.zeros_like(ux)])

    @property
    def grid(self):
        return _shear_axes(self)

    @property
    def boundaries(self):
        return []


# name -> flow class, as used by `lettuce benchmark` (lettuce/ext/_flows/_flow_by_name.py:10-16)
flow_by_name = {
    "taylor2D": [TaylorGreenVortex, D2Q9],
    "taylor3D": [TaylorGreenVortex, D3Q19],
    "shear2D": [DoublyPeriodicShear2D, D2Q9],
    "shear3D": [DoublyPeriodicShear3D, D3Q19],
}
