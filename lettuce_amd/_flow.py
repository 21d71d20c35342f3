"""State of a simulation and its macroscopic moments; the operator ABCs.

Drop-in for lettuce/_flow.py:16-236,309-336 (``Equilibrium``, ``Boundary``, ``Flow``,
``initialize_f_neq``).  ``flow.f`` is a plain ``[q, *resolution]`` tensor attribute in the
reference's C-contiguous layout; the HIP engine works on that memory directly.

On a native context (``Context.use_native``) the moments of the grid-shaped state
(``rho``, ``u`` -- SURVEY.md 8(a) row A5) are evaluated by the engine's macroscopic kernel;
for any other argument (a field of another shape, lattices without kernels) they are the
reference's whole-field torch expressions.
"""
import pickle
from abc import ABC, abstractmethod
from typing import List, Optional, Union

import numpy as np
import torch

from ._stencil import TorchStencil
from .util import torch_gradient, LettuceException

__all__ = ["Equilibrium", "Flow", "Boundary", "initialize_f_neq"]


class Equilibrium(ABC):
    @abstractmethod
    def __call__(self, flow: "Flow", rho=None, u=None) -> torch.Tensor:
        ...

    @abstractmethod
    def native_available(self) -> bool:
        ...

    @abstractmethod
    def native_generator(self) -> "NativeEquilibrium":
        ...


class Boundary(ABC):
    @abstractmethod
    def __call__(self, flow: "Flow"):
        ...

    @abstractmethod
    def make_no_collision_mask(self, shape: List[int], context: "Context") -> Optional[torch.Tensor]:
        ...

    @abstractmethod
    def make_no_streaming_mask(self, shape: List[int], context: "Context") -> Optional[torch.Tensor]:
        ...

    @abstractmethod
    def native_available(self) -> bool:
        ...

    @abstractmethod
    def native_generator(self, index: int) -> "NativeBoundary":
        ...


class Flow(ABC):
    """Physical configuration and state (lettuce/_flow.py:54-236)."""

    initialize_pressure: bool = False
    initialize_fneq: bool = False

    def __init__(self, context: "Context", resolution: List[int], units: "UnitConversion",
                 stencil: "Stencil", equilibrium: "Equilibrium"):
        self.context = context
        self.resolution = resolution
        self.units = units
        self.stencil = stencil
        self.torch_stencil = TorchStencil(stencil, context)
        self.equilibrium = equilibrium
        self.i = 0
        self.f = context.empty_tensor([stencil.q, *resolution])
        self._f_next = None
        self._moment_plan = None
        self.initialize()

    # ---- to be provided by concrete flows ---------------------------------------------------
    @property
    @abstractmethod
    def boundaries(self) -> List["Boundary"]:
        return []

    @abstractmethod
    def initial_pu(self) -> (float, Union[np.array, torch.Tensor]):
        """initial pressure and velocity in physical units"""
        ...

    # ---- initialisation (lettuce/_flow.py:106-122) -------------------------------------------
    def initialize(self):
        p0, u0 = self.initial_pu()
        rho0 = self.context.convert_to_tensor(self.units.convert_pressure_pu_to_density_lu(p0))
        u0 = self.context.convert_to_tensor(self.units.convert_velocity_to_lu(u0))
        if self.initialize_pressure:
            raise LettuceException("pressure-Poisson initialisation is outside this engine's "
                                   "scope (SURVEY.md section 2); set initialize_pressure=False")
        self.f = self.equilibrium(self, rho=rho0, u=u0)
        if self.initialize_fneq:
            self.f = initialize_f_neq(self)

    # ---- the populations (lettuce/_flow.py:90) ---------------------------------------------------
    # A plain attribute in the reference.  Here the engine may leave a batch of steps one streaming pass short
    # (the post-collision populations f* of the last step are what the next batch continues from): the pass is
    # done when somebody looks -- reading ``flow.f`` (or ``flow.f_next``) completes it, assigning drops it.
    _pending = None      # callable that finishes the batch and returns (f, f_next), or None
    _f = None

    @property
    def f(self) -> torch.Tensor:
        if self._pending is not None:
            self._finish_pending()
        return self._f

    @f.setter
    def f(self, value: torch.Tensor):
        if self._pending is not None:
            # the batch that was one pass short is abandoned: let its owner release the two population buffers
            # it holds for that pass (the engine's stepper would otherwise keep them until its next batch)
            drop = getattr(self._pending, "drop", None)
            self._pending = None
            if drop is not None:
                drop()
        self._f = value

    def _finish_pending(self):
        finish, self._pending = self._pending, None
        self._f, self._f_next = finish()

    # ---- double buffer used by the engine (lettuce/_flow.py:124-134) ------------------------
    @property
    def f_next(self) -> torch.Tensor:
        if self._pending is not None:
            self._finish_pending()
        if self._f_next is None:
            self._f_next = self.context.empty_tensor([self.stencil.q, *self.resolution])
        return self._f_next

    @f_next.setter
    def f_next(self, value: torch.Tensor):
        if self._pending is not None:
            self._finish_pending()
        self._f_next = value

    # ---- engine access for the moments -------------------------------------------------------
    def _engine_plan(self, f: torch.Tensor):
        """The engine plan for moments of ``f`` or None when ``f`` is not this flow's
        grid-shaped device state (then the torch expressions apply)."""
        if not self.context.use_native:
            return None
        from ._native import Plan, STENCIL_IDS
        name = type(self.stencil).__name__
        if name not in STENCIL_IDS or self.context.dtype not in (torch.float32, torch.float64):
            return None
        if (f.device.type != "cuda" or f.dtype != self.context.dtype or not f.is_contiguous()
                or list(f.shape) != [self.stencil.q, *self.resolution]):
            return None
        if self._moment_plan is None:
            self._moment_plan = Plan(name, self.context.dtype, "none", self.resolution,
                                     device=f.device)
        return self._moment_plan

    # ---- moments (lettuce/_flow.py:136-181) ---------------------------------------------------
    def rho(self, f: Optional[torch.Tensor] = None) -> torch.Tensor:
        """density, shape [1, *resolution]"""
        f = self.f if f is None else f
        plan = self._engine_plan(f)
        if plan is not None:
            return plan.macroscopic(f, want_u=False)[0][None, ...]
        return torch.sum(f, dim=0)[None, ...]

    @property
    def rho_pu(self) -> torch.Tensor:
        return self.units.convert_density_to_pu(self.rho())

    @property
    def p_pu(self) -> torch.Tensor:
        return self.units.convert_density_lu_to_pressure_pu(self.rho())

    @property
    def u_pu(self):
        return self.units.convert_velocity_to_pu(self.u())

    def j(self, f: Optional[torch.Tensor] = None) -> torch.Tensor:
        """momentum, shape [d, *resolution]"""
        f = self.f if f is None else f
        if f.is_cuda:
            return local_contract(self.torch_stencil.e.t(), f)      # no BLAS on device, see there
        return self.einsum("qd,q->d", [self.torch_stencil.e, f])

    def u(self, f: Optional[torch.Tensor] = None, rho=None, acceleration=None) -> torch.Tensor:
        """velocity; ``acceleration`` adds the half-force correction of a forcing scheme"""
        src = self.f if f is None else f
        if rho is None and acceleration is None:
            plan = self._engine_plan(src)
            if plan is not None:
                return plan.macroscopic(src, want_rho=False)[1]
        rho = self.rho(f=f) if rho is None else rho
        v = self.j(f=f) / rho
        if acceleration is None:
            return v + 0.0
        if len(acceleration.shape) == 1:
            acceleration = acceleration[(Ellipsis,) + (None,) * self.stencil.d]
        return v + acceleration / (2 * rho)

    @property
    def velocity(self):
        return self.j() / self.rho()

    def incompressible_energy(self, f: Optional[torch.Tensor] = None) -> torch.Tensor:
        """0.5 u.u per node"""
        u = self.u(f)
        if u.is_cuda:
            return 0.5 * (u * u).sum(dim=0)
        return 0.5 * self.einsum("d,d->", [u, u])

    def entropy(self) -> torch.Tensor:
        f_log = -torch.log(self.einsum("q,q->q", [self.f, 1 / self.torch_stencil.w]))
        return self.einsum("q,q->", [self.f, f_log])

    def pseudo_entropy_global(self) -> torch.Tensor:
        f_w = self.einsum("q,q->q", [self.f, 1 / self.torch_stencil.w])
        return self.rho() - self.einsum("q,q->", [self.f, f_w])

    def pseudo_entropy_local(self, f: Optional[torch.Tensor] = None) -> torch.Tensor:
        f = self.f if f is None else f
        f_feq = f / self.equilibrium(self)
        return self.rho(f) - self.einsum("q,q->", [f, f_feq])

    def shear_tensor(self, f: Optional[torch.Tensor] = None) -> torch.Tensor:
        ee = self.einsum("qa,qb->qab", [self.torch_stencil.e, self.torch_stencil.e])
        return self.einsum("q,qab->ab", [self.f if f is None else f, ee])

    def einsum(self, equation, fields, *args) -> torch.Tensor:
        """Einstein summation over the local (non-grid) indices: operands that carry the
        grid axes get an ellipsis appended (lettuce/_flow.py:210-224)."""
        lhs, out = equation.split("->")
        terms = lhs.split(",")
        for k, term in enumerate(terms):
            extra = len(fields[k].shape) - len(term)
            if extra == 0:
                continue
            assert extra == self.stencil.d, "Bad dimension."
            terms[k] = term + "..."
            if not out.endswith("..."):
                out += "..."
        return torch.einsum(",".join(terms) + "->" + out, fields, *args)

    # ---- checkpointing (lettuce/_flow.py:226-236) ---------------------------------------------
    def dump(self, filename):
        with open(filename, "wb") as fh:
            pickle.dump(self.context.convert_to_ndarray(self.f), fh)

    def load(self, filename):
        with open(filename, "rb") as fh:
            self.f = self.context.convert_to_tensor(pickle.load(fh), dtype=self.context.dtype)
        if self.context.use_native:
            self._f_next = self.context.empty_tensor(self.f.shape)


def local_contract(matrix: torch.Tensor, field: torch.Tensor) -> torch.Tensor:
    """out[i, x] = sum_k matrix[i, k] * field[k, x] as plain elementwise multiply-adds.

    The reference writes these per-node contractions as einsum/tensordot, which torch lowers to
    a GEMM with one dimension of a few entries and the other of 10^7..10^8 nodes.  On ROCm such
    a call faulted inside the BLAS kernel for a 512 x 512 x 70 slab (observed on MI355X,
    torch 2.10 / ROCm 7.2), so device tensors never go through BLAS here; the K <= 27 sum is
    evaluated term by term (differs from the GEMM result by rounding only)."""
    out = None
    for k in range(matrix.shape[1]):
        col = matrix[:, k].reshape([-1] + [1] * (field.dim() - 1))
        term = col * field[k][None, ...]
        out = term if out is None else out.add_(term)
    return out


def initialize_f_neq(flow: "Flow"):
    """f = feq - f(1), the non-equilibrium part estimated from 6th-order finite differences
    of u (lettuce/_flow.py:309-336; Krueger et al. 2017)."""
    d = flow.stencil.d
    plan = flow._engine_plan(flow.f) if d >= 2 else None
    if plan is not None:
        # one launch of the engine (lt_init_fneq): the moments of the equilibrium populations, then
        # gradients, Pi1:Q and feq per node -- no [d, d, *res] / [q, *res] temporaries, no BLAS.
        # The reference's identity is built in torch's default dtype: an fp32-rounded cs^2.
        rho, u = plan.macroscopic(flow.f)
        eye_cs2 = float(torch.tensor(flow.stencil.cs ** 2, dtype=torch.get_default_dtype()))
        return plan.init_fneq(rho, u, flow.units.relaxation_parameter_lu, eye_cs2)
    rho = flow.rho()
    u = flow.u()
    grad_u = torch.cat([torch_gradient(u[a], dx=1, order=6)[None, ...] for a in range(d)])
    pi_1 = 1.0 * flow.units.relaxation_parameter_lu * rho * grad_u / flow.torch_stencil.cs ** 2
    e = flow.torch_stencil.e
    # the identity is built in torch's default dtype, as in the reference (fp32-rounded cs^2)
    q_tensor = (torch.einsum("ia,ib->iab", [e, e])
                - torch.eye(d, device=e.device) * flow.stencil.cs ** 2)
    if pi_1.is_cuda:
        pi_1_q = local_contract(q_tensor.reshape(q_tensor.shape[0], d * d).to(pi_1.dtype),
                                pi_1.reshape([d * d] + list(pi_1.shape[2:])))
    else:
        pi_1_q = flow.einsum("ab,iab->i", [pi_1, q_tensor])
    f_neq = flow.einsum("i,i->i", [flow.torch_stencil.w, pi_1_q])
    return flow.equilibrium(flow, rho, u) - f_neq
