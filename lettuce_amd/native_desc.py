"""Descriptors handed from the Python operators to the HIP engine.

The reference's components return *code emitters* from ``native_generator()``
(lettuce/_flow.py:21-27,45-51, lettuce/_simulation.py:21-27,119-127) because its kernel is
generated and JIT-compiled per configuration.  The engine here is prebuilt, so the same
protocol returns plain descriptors: an enum-like ``kind`` plus the scalar parameters the
kernels take.
"""
from dataclasses import dataclass, field
from typing import Callable, Optional

__all__ = ["NativeEquilibrium", "NativeCollision", "NativeBoundary"]


@dataclass
class NativeEquilibrium:
    kind: str = "quadratic"


@dataclass
class NativeCollision:
    kind: str                                  # 'none' | 'bgk' | 'kbc'
    # relaxation time used for the next batch of steps; evaluated per call because the
    # reference re-reads collision.tau on every invocation
    # (lettuce/cuda_native/ext/_collision/bgk_collision.py:30)
    tau: Callable[["Flow"], float] = field(default=lambda flow: 1.0)
    # "exact": the reference's floating-point operations one for one (the default; bit-identical periodic BGK flows);
    # "fast": the engine's shorter BGK collision, equal to rounding level (lt_plan_set_arithmetic) -- opt-in through
    # ``collision.arithmetic = "fast"``, never chosen by the engine
    arithmetic: str = "exact"


@dataclass
class NativeBoundary:
    kind: str                                  # 'bounce_back' | 'equilibrium' | 'abb_outlet'
    index: int
    # engine parameters of this boundary for a given flow (dict for lettuce_amd._native.Plan)
    params: Optional[Callable[["Flow"], dict]] = None

    def plan_entry(self, flow) -> dict:
        entry = {"kind": self.kind}
        if self.params is not None:
            entry.update(self.params(flow))
        return entry
