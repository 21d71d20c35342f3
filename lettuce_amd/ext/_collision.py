"""Collision operators on the hot path: BGK, KBC (D2Q9 / D3Q27) and NoCollision.

Each ``__call__`` is a pure whole-field function ``flow -> tensor`` usable outside a
``Simulation`` (the reference's tests call ``collision(flow)`` directly).  On a native context
and for the flow's grid-shaped state it is one launch of the engine's collide kernel;
otherwise the reference's torch expressions are evaluated.  TRT / MRT / regularised /
Smagorinsky collisions and forcing schemes are out of scope.
"""
import warnings
from typing import AnyStr, Optional

import torch

from .._simulation import Collision
from ..native_desc import NativeCollision
from ..util import LettuceException

__all__ = ["BGKCollision", "KBCCollision", "KBCCollision2D", "KBCCollision3D", "NoCollision"]


def _engine_collide(flow, kind, tau):
    """C(flow.f) through the HIP engine, or None when flow.f is not engine-shaped."""
    if flow._engine_plan(flow.f) is None:
        return None
    plans = flow.__dict__.setdefault("_collision_plans", {})
    if kind not in plans:
        from .._native import Plan
        plans[kind] = Plan(type(flow.stencil).__name__, flow.context.dtype, kind, flow.resolution,
                           device=flow.f.device)
    return plans[kind].collide(flow.f, torch.empty_like(flow.f), tau)


class BGKCollision(Collision):
    """f - (f - feq(rho, u)) / tau (lettuce/ext/_collision/bgk_collision.py:12-35).

    ``arithmetic`` (an attribute, not part of the reference's signature): "exact" (default) -- the HIP engine reproduces
    the reference's floating-point operations one for one; "fast" -- its shorter collision, equal to rounding level
    (periodic 3-D flows; ``Simulation`` raises where the engine has no such kernel)."""
    arithmetic = "exact"

    def __init__(self, tau, force: Optional["Force"] = None):
        self.tau = tau
        self.force = force

    def __call__(self, flow: "Flow") -> torch.Tensor:
        if self.force is None:
            out = _engine_collide(flow, "bgk", self.tau)
            if out is not None:
                return out
            u = flow.u() + 0
            feq = flow.equilibrium(flow, u=u)
            return flow.f - 1.0 / self.tau * (flow.f - feq) + 0
        u = flow.u() + self.force.u_eq(flow)
        feq = flow.equilibrium(flow, u=u)
        return flow.f - 1.0 / self.tau * (flow.f - feq) + self.force.source_term(u)

    def name(self) -> AnyStr:
        if self.force is not None:
            return f"{type(self).__name__}_{type(self.force).__name__}"
        return type(self).__name__

    def native_available(self) -> bool:
        return self.force is None

    def native_generator(self) -> "NativeCollision":
        return NativeCollision("bgk", tau=lambda flow: self.tau, arithmetic=getattr(self, "arithmetic", "exact"))


class KBCCollision(Collision):
    """Entropic multi-relaxation model of Karlin, Boesch, Chikatamarla
    (lettuce/ext/_collision/kbc_collision.py:11-166).

    As in the reference the constructor's ``tau`` is not used: on the first call tau is taken
    from ``flow.units.relaxation_parameter_lu`` (kbc_collision.py:97-99)."""

    def __init__(self, tau: float = None):
        self.tau = tau
        self.beta = None
        self._ready = False

    def _prepare(self, flow):
        if self._ready:
            return
        name = type(flow.stencil).__name__
        if flow.stencil.d == 3:
            assert name == "D3Q27", "KBC Collision is only implemented for D3Q27!"
        elif flow.stencil.d == 2:
            assert name == "D2Q9", "KBC Collision is only implemented for D2Q9!"
        else:
            raise NotImplementedError("KBC Collision is only implemented for 2d and 3d!")
        self.tau = flow.units.relaxation_parameter_lu
        self.beta = 1. / (2 * self.tau)
        self._ready = True

    # second moments of a population set, normalised by its own density
    @staticmethod
    def _moments(flow, g):
        e = flow.torch_stencil.e
        rho = torch.sum(g, dim=0)

        def m(a, b):
            coeff = e[:, a] * e[:, b]
            if g.is_cuda:       # no BLAS on device tensors (see _flow.local_contract)
                from .._flow import local_contract
                return local_contract(coeff[None, :], g)[0] / rho
            return torch.einsum("q,q...->...", coeff, g) / rho

        return rho, m

    def _shear_vector(self, flow, g):
        """s_i of kbc_collision.py:44-94; the corner populations of D3Q27 get zero."""
        rho, m = self._moments(flow, g)
        s = torch.zeros_like(g)
        if flow.stencil.d == 3:
            xx, yy, zz = m(0, 0), m(1, 1), m(2, 2)
            trace, n_xz, n_yz = xx + yy + zz, xx - zz, yy - zz
            s[0] = rho * -trace
            s[1] = s[2] = 1. / 6. * rho * (2 * n_xz - n_yz + trace)
            s[3] = s[4] = 1. / 6. * rho * (2 * n_yz - n_xz + trace)
            s[5] = s[6] = 1. / 6. * rho * (-n_xz - n_yz + trace)
            for first, (a, b) in ((7, (1, 2)), (11, (0, 2)), (15, (0, 1))):
                p = 1. / 4 * rho * m(a, b)
                s[first] = s[first + 1] = p
                s[first + 2] = s[first + 3] = -p
        else:
            xx, yy = m(0, 0), m(1, 1)
            trace, n = xx + yy, xx - yy
            s[0] = rho * -trace
            s[1] = s[3] = 1. / 2. * rho * (0.5 * (trace + n))
            s[2] = s[4] = 1. / 2. * rho * (0.5 * (trace - n))
            p = 1. / 4. * rho * m(0, 1)
            s[5] = s[7] = p
            s[6] = s[8] = -p
        return s

    def __call__(self, flow: "Flow") -> torch.Tensor:
        self._prepare(flow)
        out = _engine_collide(flow, "kbc", self.tau)
        if out is not None:
            return out
        feq = flow.equilibrium(flow)
        delta_s = self._shear_vector(flow, flow.f) - self._shear_vector(flow, feq)
        delta_h = flow.f - feq - delta_s
        sum_s = flow.rho(delta_s * delta_h / feq)
        sum_h = flow.rho(delta_h * delta_h / feq)
        gamma = 1. / self.beta - (2 - 1. / self.beta) * sum_s / sum_h
        gamma[gamma < 1E-15] = 2.0
        gamma[torch.isnan(gamma)] = 2.0
        return flow.f - self.beta * (2 * delta_s + gamma * delta_h)

    def native_available(self) -> bool:
        return True

    def native_generator(self) -> "NativeCollision":
        def tau(flow):
            self._prepare(flow)
            return self.tau
        return NativeCollision("kbc", tau=tau)


class KBCCollision2D(KBCCollision):
    def __init__(self, tau: float = None):
        warnings.warn("KBCCollision2D is is deprecated! Use KBCCollision instead!")
        super().__init__()


class KBCCollision3D(KBCCollision):
    def __init__(self, tau: float = None):
        warnings.warn("KBCCollision3D is is deprecated! Use KBCCollision instead!")
        super().__init__()


class NoCollision(Collision):
    """Identity (lettuce/ext/_collision/no_collision.py:9-17); used by streaming tests."""

    def __call__(self, flow: "Flow") -> torch.Tensor:
        return flow.f

    def native_available(self) -> bool:
        return True

    def native_generator(self) -> "NativeCollision":
        return NativeCollision("none")
