"""Boundary conditions on the hot path: bounce-back, equilibrium inlet (pu), anti-bounce-back
outlet -- the three ``Obstacle.boundaries`` returns (lettuce/ext/_flows/obstacle.py:108-122).

All three live in this one module; ``Simulation`` orders boundaries by ``str(boundary)``,
i.e. by class path, and the class names alone reproduce the reference's order
(AntiBounceBackOutlet < BounceBackBoundary < EquilibriumBoundaryPU).
``EquilibriumOutletP`` is out of scope (commented out in Obstacle, its test is skipped).
"""
from typing import List, Optional

import numpy as np
import torch

from ..util import LettuceException

from .._flow import Boundary
from ..native_desc import NativeBoundary
from ._collision import BGKCollision

__all__ = ["BounceBackBoundary", "EquilibriumBoundaryPU", "AntiBounceBackOutlet"]


class BounceBackBoundary(Boundary):
    """Full-way bounce-back: on the masked (solid, un-collided) nodes f_q <- f_opposite(q),
    then ordinary streaming (lettuce/ext/_boundary/bounce_back_boundary.py:10-32)."""

    def __init__(self, mask: torch.Tensor):
        self._mask = mask

    def __call__(self, flow: "Flow"):
        return flow.f[flow.stencil.opposite]

    def make_no_streaming_mask(self, shape: List[int], context: "Context") -> Optional[torch.Tensor]:
        return None

    def make_no_collision_mask(self, shape: List[int], context: "Context") -> Optional[torch.Tensor]:
        return self._mask

    def native_available(self) -> bool:
        return True

    def native_generator(self, index: int) -> "NativeBoundary":
        return NativeBoundary("bounce_back", index)


class EquilibriumBoundaryPU(Boundary):
    """f <- feq(rho(p_pu), u_lu(v_pu)) on the masked nodes
    (lettuce/ext/_boundary/equilibrium_boundary_pu.py:13-46)."""

    def __init__(self, context: "Context", mask, velocity, pressure=0):
        velocity = velocity if hasattr(velocity, "__len__") else [velocity]
        self.velocity = context.convert_to_tensor(velocity)
        self.pressure = context.convert_to_tensor(pressure)
        self._mask = mask
        self._cache = None

    def _feq(self, flow):
        rho = flow.units.convert_pressure_pu_to_density_lu(self.pressure)
        u = flow.units.convert_velocity_to_lu(self.velocity)
        return flow.equilibrium(flow, rho, u)

    def __call__(self, flow: "Flow"):
        feq = self._feq(flow)
        return flow.einsum("q,q->q", [feq, torch.ones_like(flow.f)])

    def make_no_collision_mask(self, shape: List[int], context: "Context") -> Optional[torch.Tensor]:
        return self._mask

    def make_no_streaming_mask(self, shape: List[int], context: "Context") -> Optional[torch.Tensor]:
        return None

    def native_available(self) -> bool:
        return True

    def _engine_params(self, flow):
        # the reference's kernel re-reads velocity/pressure every step
        # (lettuce/cuda_native/ext/_boundary/equilibrium_pu.py:40-48); re-evaluate when they change
        key = (id(self.velocity), self.velocity._version, id(self.pressure), self.pressure._version,
               flow.units.characteristic_velocity_lu, flow.units.characteristic_pressure_lu)
        if self._cache is None or self._cache[0] != key:
            feq = self._feq(flow)
            if feq.dim() == 1:
                entry = {"feq": [float(v) for v in feq.detach().cpu().double()]}
            else:
                entry = {"field": self(flow).contiguous()}
            self._cache = (key, entry)
        return self._cache[1]

    def native_generator(self, index: int) -> "NativeBoundary":
        return NativeBoundary("equilibrium", index, params=self._engine_params)


class AntiBounceBackOutlet(Boundary):
    """Outlet on one face of the domain after Krueger et al. (2016), p. 195
    (lettuce/ext/_boundary/anti_bounce_back_outlet.py:13-109).

    ``direction`` is a one-hot list with +1 / -1, e.g. ``[1, 0, 0]`` for the +x face."""

    def __init__(self, direction: List[int], flow: "Flow", collision: "Collision" = None):
        self.collision = (BGKCollision(tau=flow.units.relaxation_parameter_lu)
                          if collision is None else collision)
        direction = list(direction)
        assert len(direction) in [1, 2, 3], \
            f"Invalid direction parameter. Expected direction of of length 1, 2 or 3 but got {len(direction)}."
        assert (direction.count(0) == (len(direction) - 1)) and ((1 in direction) ^ (-1 in direction)), \
            f"Invalid direction parameter. Expected direction with all entries 0 except one 1 or -1 but got {direction}."
        self.direction = direction
        self.stencil = flow.torch_stencil
        e = np.array(flow.stencil.e)
        # populations leaving through the face: e_q . direction == 1
        self.velocities = np.nonzero(e @ np.array(direction) > 1 - 1e-6)[0]
        self.axis = [k for k, c in enumerate(direction) if c != 0][0]
        self.side = direction[self.axis]
        self.index = [slice(None)] * len(direction)
        self.neighbor = [slice(None)] * len(direction)
        self.index[self.axis] = -1 if self.side > 0 else 0
        self.neighbor[self.axis] = -2 if self.side > 0 else 1
        # z-slab decomposition (multi-GPU extension, lettuce_amd/_slab.py): an outlet along the decomposed
        # axis lives on the rank that holds the first / last plane of the GLOBAL grid; `flow` is that
        # rank's extended slab, so the plane's index is shifted, and the other ranks have no outlet
        self.present = True
        slab = getattr(flow, "slab", None)
        if slab is not None and self.axis == 2:
            plane = (slab.global_resolution[2] - 1 if self.side > 0 else 0) - slab.z_begin
            self.present = 0 <= plane < slab.nz_local
            if self.present and not 0 <= plane - self.side < slab.nz_local:
                raise LettuceException("an outlet along z needs its plane and the plane next to it on one rank "
                                       "(at least two planes per rank)")
            self.index[2] = slab.halo + plane
            self.neighbor[2] = slab.halo + plane - self.side
        w = flow.torch_stencil.w[self.velocities]
        self.w = w.reshape([-1] + [1] * (len(direction) - 1))

    def _opposite_of_velocities(self, context):
        return context.convert_to_ndarray(self.stencil.opposite)[self.velocities]

    def __call__(self, flow: "Flow"):
        if not self.present:
            return flow.f
        st = flow.torch_stencil
        u = flow.u()
        here = tuple([slice(None)] + self.index)
        u_w = u[here] + 0.5 * (u[here] - u[tuple([slice(None)] + self.neighbor)])
        if u_w.is_cuda:         # no BLAS on device tensors (see _flow.local_contract)
            from .._flow import local_contract
            e_dot_u = local_contract(st.e[self.velocities], u_w)
        else:
            e_dot_u = torch.einsum("cd,d...->c...", st.e[self.velocities], u_w)
        f = flow.f
        f[tuple([self._opposite_of_velocities(flow.context)] + self.index)] = (
            - flow.f[tuple([self.velocities] + self.index)]
            + self.w * flow.rho()[here]
            * (2 + e_dot_u ** 2 / st.cs ** 4 - (torch.norm(u_w, dim=0) / st.cs) ** 2))
        return f

    def make_no_streaming_mask(self, f_shape, context: "Context"):
        mask = torch.zeros(size=list(f_shape), dtype=torch.bool, device=context.device)
        if self.present:
            mask[tuple([self._opposite_of_velocities(context)] + self.index)] = 1
        return mask

    def make_no_collision_mask(self, shape: List[int], context: "Context"):
        mask = context.zero_tensor(shape, dtype=bool)
        if self.present:
            mask[tuple(self.index)] = 1
        return mask

    def native_available(self) -> bool:
        return True

    def native_generator(self, index: int) -> "NativeBoundary":
        return NativeBoundary("abb_outlet", index,
                              params=lambda flow: {"axis": self.axis, "side": self.side, "present": self.present})
