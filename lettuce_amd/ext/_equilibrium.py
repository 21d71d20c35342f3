"""Equilibrium distributions on the hot path.

``QuadraticEquilibrium`` is the default of every flow (lettuce/ext/_flows/_ext_flow.py:30)
and the only equilibrium the configs use; the reference's LessMemory / incompressible
variants are out of scope.
"""
import torch

from .._flow import Equilibrium
from ..native_desc import NativeEquilibrium

__all__ = ["QuadraticEquilibrium"]


class QuadraticEquilibrium(Equilibrium):
    """feq_q = w_q rho ((2 e_q.u - u.u) / (2 cs^2) + (e_q.u / cs^2)^2 / 2 + 1)
    (lettuce/ext/_equilibrium/quadratic_equilibrium.py:11-25)."""

    def __call__(self, flow: "Flow", rho=None, u=None):
        plan = flow._engine_plan(flow.f) if (rho is None and u is None) else None
        if plan is not None:
            rho_, u_ = plan.macroscopic(flow.f)
            return plan.equilibrium(rho_, u_)
        rho = flow.rho() if rho is None else rho
        u = flow.u() if u is None else u
        st = flow.torch_stencil
        grid = list(flow.resolution)
        if (flow.context.use_native and torch.is_tensor(rho) and list(rho.shape) == [1] + grid
                and list(u.shape) == [st.d] + grid and flow._engine_plan(flow.f) is not None):
            # whole-field feq(rho, u) on a native context: the engine's equilibrium kernel
            return flow._engine_plan(flow.f).equilibrium(rho.to(flow.f.dtype), u.contiguous())
        if u.is_cuda and u.dim() > 1:
            from .._flow import local_contract
            e_dot_u = local_contract(st.e, u)
        else:
            e_dot_u = torch.tensordot(st.e, u, dims=1)
        u_sq = (u * u).sum(dim=0) if u.is_cuda else flow.einsum("d,d->", [u, u])
        bracket = (2 * e_dot_u - u_sq) / (2 * st.cs ** 2) + 0.5 * (e_dot_u / (st.cs ** 2)) ** 2 + 1
        return flow.einsum("q,q->q", [st.w, rho * bracket])

    def native_available(self) -> bool:
        return True

    def native_generator(self) -> "NativeEquilibrium":
        return NativeEquilibrium("quadratic")
