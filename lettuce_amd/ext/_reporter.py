"""Observables read between steps ("next" row F3) and the reporter that logs them.

API of lettuce/ext/_reporter/observable_reporter.py:17-42,140-199.  On a native context the
kinetic energy, the maximum velocity, the enstrophy and the Mass observable are device-side
wavefront/LDS reductions in fp64 (``lt_kinetic_energy``, ``lt_max_velocity``, ``lt_enstrophy``,
``lt_mass_interior``); the enstrophy uses one [d, *res] scratch field for u, the others none.
VTK / HDF5 / image writers, the energy spectrum and the error reporter are host-side
post-processing and out of scope.
"""
import sys
from abc import ABC, abstractmethod
from typing import Optional

import torch

from .._simulation import Reporter
from ..util import torch_gradient

__all__ = ["Observable", "ObservableReporter", "MaximumVelocity", "IncompressibleKineticEnergy",
           "Enstrophy", "Mass", "ErrorReporter"]


class Observable(ABC):
    def __init__(self, flow: "Flow"):
        self.context = flow.context
        self.flow = flow

    @abstractmethod
    def __call__(self, f: Optional[torch.Tensor] = None):
        ...


class MaximumVelocity(Observable):
    """max |u| in physical units (observable_reporter.py:27-31); a device max-reduction on a
    native context."""

    def __call__(self, f: Optional[torch.Tensor] = None):
        plan = self.flow._engine_plan(self.flow.f)
        if plan is not None:
            return self.flow.units.convert_velocity_to_pu(plan.max_velocity_lu(self.flow.f))
        return torch.norm(self.flow.u_pu, dim=0).max()


class IncompressibleKineticEnergy(Observable):
    """E_pu = sum_x 0.5 u_lu.u_lu * (u_char_pu / u_char_lu)^2 * dx_pu^d  -- the parity metric of
    the north star (lettuce/ext/_reporter/observable_reporter.py:34-42)."""

    def __call__(self, f: Optional[torch.Tensor] = None):
        flow = self.flow
        dx = flow.units.convert_length_to_pu(1.0)
        plan = flow._engine_plan(flow.f)
        if plan is not None:
            total = plan.kinetic_energy_lu(flow.f)      # fp64 device scalar
        else:
            total = torch.sum(flow.incompressible_energy())
        kin = flow.units.convert_incompressible_energy_to_pu(total)
        kin *= dx ** flow.stencil.d
        return kin


class Enstrophy(Observable):
    """Integral of the squared vorticity; periodic domains only
    (observable_reporter.py:45-68)."""

    def __call__(self, f: Optional[torch.Tensor] = None):
        flow = self.flow
        dx = flow.units.convert_length_to_pu(1.0)
        plan = flow._engine_plan(flow.f) if flow.stencil.d >= 2 else None
        if plan is not None:
            # device reduction (lt_enstrophy): u into one scratch field, then the vorticity stencil
            total = plan.enstrophy_sum(flow.f, flow.units.convert_velocity_to_pu(1.0), 1.0 / dx)
            return (total * dx ** flow.stencil.d).to(flow.f.dtype)
        u = flow.units.convert_velocity_to_pu(flow.u())
        grad = [torch_gradient(u[a], dx=dx, order=6) for a in range(flow.stencil.d)]
        w_z = grad[0][1] - grad[1][0]
        total = torch.sum(w_z * w_z)
        if flow.stencil.d == 3:
            w_x = grad[2][1] - grad[1][2]
            w_y = grad[0][2] - grad[2][0]
            total += torch.sum(w_x * w_x + w_y * w_y)
        return total * dx ** flow.stencil.d


class Mass(Observable):
    """Total mass in lattice units (observable_reporter.py:140-158)."""

    def __init__(self, flow: "Flow", no_mass_mask=None):
        super().__init__(flow)
        self.mask = no_mass_mask

    def __call__(self, f: Optional[torch.Tensor] = None):
        plan = self.flow._engine_plan(f) if (f is not None and self.flow.stencil.d >= 2) else None
        if plan is not None and (self.mask is None or list(self.mask.shape) == list(f.shape[1:])):
            return plan.mass_interior(f, self.mask).to(f.dtype)      # device reduction (lt_mass_interior)
        mass = f[..., 1:-1, 1:-1].sum()
        if self.mask is not None:
            mass -= (f * self.mask.to(dtype=torch.float)).sum()
        return mass


class ObservableReporter(Reporter):
    """Evaluates ``observable`` every ``interval`` steps and prints ``i t_pu value...`` or
    appends it to ``out`` when that is a list (observable_reporter.py:161-199)."""
    batchable = True          # does nothing unless flow.i % interval == 0: steps in between may be fused

    def __init__(self, observable, interval=1, out=sys.stdout):
        super().__init__(interval)
        self.observable = observable
        self.out = [] if out is None else out
        self._parameter_name = type(observable).__name__
        print("steps    ", "time    ", self._parameter_name)

    def __call__(self, simulation: "Simulation"):
        if simulation.flow.i % self.interval != 0:
            return
        observed = self.observable.context.convert_to_ndarray(self.observable(simulation.flow.f))
        assert len(observed.shape) < 2
        values = [observed.item()] if observed.ndim == 0 else observed.tolist()
        entry = [simulation.flow.i, simulation.units.convert_time_to_pu(simulation.flow.i)] + values
        if isinstance(self.out, list):
            self.out.append(entry)
        else:
            print(*entry, file=self.out)


class ErrorReporter(Reporter):
    """L2 errors of u and p (physical units) against an analytic solution, e.g.
    ``flow.analytic_solution`` of the 2-D Taylor-Green vortex; used by the convergence check
    ("next" row F4; lettuce/ext/_reporter/error_reporter.py:9-48, lettuce/cli.py:128-180)."""
    batchable = True

    def __init__(self, analytical_solution, interval=1, out=sys.stdout):
        Reporter.__init__(self, interval)
        self.analytical_solution = analytical_solution
        self.out = [] if out is None else out
        if not isinstance(self.out, list):
            print("#error_u         error_p", file=self.out)

    def __call__(self, simulation: "Simulation"):
        flow = simulation.flow
        if flow.i % self.interval != 0:
            return
        p_ref, u_ref = self.analytical_solution(t=simulation.units.convert_time_to_pu(flow.i))
        p_ref = flow.context.convert_to_tensor(p_ref)
        u_ref = flow.context.convert_to_tensor(u_ref)
        p, u = flow.p_pu, flow.u_pu
        d = flow.stencil.d
        nodes = 1
        for n in p.size():
            nodes *= n
        scale = (nodes ** (1 / d)) ** (d / 2)
        err_u = (torch.norm(u - u_ref) / scale).item()
        err_p = (torch.norm(p - p_ref) / scale).item()
        if isinstance(self.out, list):
            self.out.append([err_u, err_p])
        else:
            print(err_u, err_p, file=self.out)
