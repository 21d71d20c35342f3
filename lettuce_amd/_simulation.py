"""The time-step driver.

Drop-in for lettuce/_simulation.py:16-207: ``Collision`` / ``Reporter`` ABCs and
``Simulation(flow, collision, reporter)`` with ``simulation(num_steps) -> MLUPS``, the public
attributes ``flow, context, collision, reporter, boundaries, no_collision_mask,
no_streaming_mask`` and the swap point ``_collide_and_stream``.

With ``context.use_native`` the per-step hot loop (collide, boundaries, stream) runs in the
HIP engine (lettuce_amd/csrc via lettuce_amd/_native.py).  There is no downgrade path: a
component the engine has no kernel for raises ``LettuceException``.  The whole-field torch
implementation (``_collide`` / ``_stream``) is the reference's non-native mode and is only used
when the context says ``use_native=False``.
"""
import warnings
from abc import ABC, abstractmethod
from timeit import default_timer as timer
from typing import List, Optional

import numpy as np
import torch

from .util import LettuceException, NativeEngineError

__all__ = ["Collision", "Reporter", "Simulation"]


class Collision(ABC):
    @abstractmethod
    def __call__(self, flow: "Flow"):
        ...

    @abstractmethod
    def native_available(self) -> bool:
        ...

    @abstractmethod
    def native_generator(self) -> "NativeCollision":
        ...


class Reporter(ABC):
    interval: int
    # The reference calls every reporter after every step (lettuce/_simulation.py:203-205).  The engine
    # fuses the steps between two calls into one batch only for reporters that declare that they do
    # nothing unless ``flow.i % interval == 0`` (the library's reporters do); any other Reporter
    # subclass is called after every single step, exactly as in the reference.
    batchable: bool = False

    def __init__(self, interval: int):
        self.interval = interval

    @abstractmethod
    def __call__(self, simulation: "Simulation"):
        ...


def build_masks(flow, boundaries, context):
    """no_collision_mask (uint8 [*res], value = boundary index, later boundaries overwrite
    earlier ones) and no_streaming_mask (uint8 [q, *res], OR of all) for ``boundaries`` =
    ``[None] + sorted(flow.boundaries, key=str)``; (None, None) without boundaries
    (lettuce/_simulation.py:63-86)."""
    if len(boundaries) <= 1:
        return None, None
    grid = [n for n in flow.f.shape[1:]]
    ncm = context.zero_tensor(flow.resolution, dtype=torch.uint8)
    nsm = context.zero_tensor([flow.stencil.q, *flow.resolution], dtype=torch.uint8)
    for index, boundary in enumerate(boundaries[1:], start=1):
        mask = boundary.make_no_collision_mask(grid, context=context)
        if mask is not None:
            ncm[mask] = index            # last writer wins
        mask = boundary.make_no_streaming_mask([flow.stencil.q] + grid, context=context)
        if mask is not None:
            nsm |= mask
    return ncm, nsm


def _version(t: Optional[torch.Tensor]):
    return None if t is None else (id(t), t.data_ptr(), t._version, tuple(t.shape))


class _NativeStepper:
    """Owns the engine plan of one Simulation and advances it in batches.

    A batch of k steps is ``lt_run``: one collide launch, k-1 fused stream-collide launches
    and one stream launch, so that ``flow.f`` again holds post-streaming populations (the
    reference's state convention) whenever Python can look at it.  If nothing touched
    ``flow.f`` between two batches the engine carries on from the post-collision buffer
    (``lt_continue``) and saves the collide pass.
    """

    def __init__(self, sim: "Simulation"):
        from ._native import Plan, STENCIL_IDS
        self.sim = sim
        flow, ctx = sim.flow, sim.context
        problems = []
        lattice = type(flow.stencil).__name__
        if lattice not in STENCIL_IDS:
            problems.append(f"stencil '{lattice}' (kernels exist for {sorted(STENCIL_IDS)})")
        if ctx.dtype not in (torch.float32, torch.float64):
            problems.append(f"dtype {ctx.dtype} (float32/float64 only)")
        if flow.equilibrium is not None and not flow.equilibrium.native_available():
            problems.append(f"equilibrium '{type(flow.equilibrium).__name__}'")
        if not sim.collision.native_available():
            problems.append(f"collision '{type(sim.collision).__name__}'")
        for b in sim.boundaries[1:]:
            if not b.native_available():
                problems.append(f"boundary '{type(b).__name__}'")
        if problems:
            raise NativeEngineError(
                "the HIP engine was requested (Context.use_native) but has no kernel for: "
                + "; ".join(problems) + ". Use Context(use_native=False) for the torch path.")
        self.collision = sim.collision.native_generator()
        self.boundaries = [b.native_generator(i) for i, b in enumerate(sim.boundaries[1:], start=1)]
        if self.collision.kind == "kbc" and lattice not in ("D2Q9", "D3Q27"):
            raise NativeEngineError("KBC Collision is only implemented for D2Q9 and D3Q27!")
        self._entries = [b.plan_entry(flow) for b in self.boundaries]
        self.plan = Plan(lattice, ctx.dtype, self.collision.kind, flow.resolution, self._entries,
                         device=ctx.device)
        if getattr(self.collision, "arithmetic", "exact") != "exact":
            self.plan.set_arithmetic(self.collision.arithmetic)      # raises where the engine has no such kernel
        self._mask_state = None
        self._carry = None      # state that allows lt_continue
        self._lazy = None       # (f*, scratch, versions) while flow.f is one streaming pass short
        # optional (start, end) torch.cuda.Event pair: when set, lt_run / lt_continue record them on
        # the launch stream around the fused launches of a batch (lt_plan_set_fused_events;
        # bench.py times the dominant kernel live with them, plan.last_run_info() says what they bracket)
        self.fused_events = None

    # ---- masks -------------------------------------------------------------------------------
    def _streaming_mask_full(self):
        """The reference indexes ``no_streaming_mask[i]`` and broadcasts the result against
        ``f[i]`` (lettuce/_simulation.py:171-174); tests assign grid-shaped masks after
        construction.  Expand whatever is there to uint8 [q, *res] with those semantics."""
        nsm = self.sim.no_streaming_mask
        if nsm is None:
            return None
        flow = self.sim.flow
        q, res = flow.stencil.q, list(flow.resolution)
        if list(nsm.shape) == [q] + res:
            return nsm.to(torch.uint8)
        rows = [torch.zeros(res, dtype=torch.bool, device=nsm.device)]
        for i in range(1, q):
            rows.append(torch.broadcast_to(torch.eq(nsm[i], 1), res))
        return torch.stack(rows).to(torch.uint8)

    def _sync_masks(self):
        sim = self.sim
        state = (_version(sim.no_collision_mask), _version(sim.no_streaming_mask))
        if state == self._mask_state:
            return
        if sim.no_collision_mask is None and sim.no_streaming_mask is None:
            if self.boundaries:
                raise NativeEngineError("boundaries are present but no_collision_mask is None")
            self.plan.set_masks(None, None)
        else:
            self.plan.set_masks(sim.no_collision_mask, self._streaming_mask_full())
        self._mask_state = state
        self._carry = None

    def _sync_boundaries(self):
        entries = [b.plan_entry(self.sim.flow) for b in self.boundaries]
        for i, (new, old) in enumerate(zip(entries, self._entries)):
            changed = any(k not in old or (isinstance(v, torch.Tensor) and v is not old[k])
                          or (not isinstance(v, torch.Tensor) and v != old[k]) for k, v in new.items())
            if changed:
                self.plan.update_boundary(i, new)
                self._carry = None
        self._entries = entries

    # ---- stepping ----------------------------------------------------------------------------
    def _state_buffers(self):
        flow = self.sim.flow
        shape = [flow.stencil.q, *flow.resolution]
        f = flow.f
        if list(f.shape) != shape:
            raise NativeEngineError(f"flow.f has shape {list(f.shape)}, the engine steps {shape}")
        if f.device.type != "cuda" or f.dtype != self.sim.context.dtype or not f.is_contiguous():
            flow.f = f = f.to(device=self.sim.context.device, dtype=self.sim.context.dtype).contiguous()
        nxt = flow.f_next
        if (list(nxt.shape) != shape or nxt.dtype != f.dtype or nxt.device != f.device
                or not nxt.is_contiguous() or nxt.data_ptr() == f.data_ptr()):
            flow.f_next = nxt = torch.empty_like(f)
        return f, nxt

    def batch(self, k: int):
        """Advance ``k`` whole steps.  The engine stops one streaming pass short (``lt_plan_set_deferred_stream``):
        the post-collision populations of the last step stay in one buffer, and ``flow.f`` completes the pass
        when it is read (Flow.f) -- a caller that steps in batches without looking in between (reporters every
        k steps, benchmark loops) pays k fused launches per batch and nothing else."""
        flow = self.sim.flow
        self._sync_masks()
        self._sync_boundaries()
        tau = float(self.collision.tau(flow))
        if self.plan.resident_enabled()[0]:
            try:
                return self._batch_resident(k, tau)
            except NativeEngineError as exc:
                # the engine's own padded buffers could not be allocated (two more population fields, raw
                # hipMalloc; torch's cache was emptied and the allocation tried again).  "Automatic" then carries on
                # with the caller's dense tensors; a plan on which resident populations were REQUESTED keeps the error
                if getattr(exc, "code", None) != 4 or self.plan.resident_mode == 1:
                    raise
                warnings.warn(f"resident populations unavailable ({exc}); stepping on flow.f / flow.f_next", stacklevel=3)
                self.plan.set_resident(0)
                self._carry = None
        changed = self._carry is None                         # first batch, or masks / boundaries were replaced
        pending = flow._pending is not None and self._lazy is not None
        if pending:
            fstar, scratch, token = self._lazy
            pending = token == (_version(fstar), _version(scratch), tau) and self._carry == "lazy"
        if pending:
            a, b, from_fstar = fstar, scratch, True
        else:
            f, nxt = self._state_buffers()                    # reading flow.f completes a pending batch
            carry = not changed and self._carry == (_version(f), _version(nxt), tau)
            a, b, from_fstar = (nxt, f, True) if carry else (f, nxt, False)
        if self.fused_events is not None:
            self.plan.set_fused_events(*self.fused_events)    # recorded by lt_run around its fused launches
        self.plan.set_deferred_stream(True)
        try:
            fstar, scratch = self.plan.run(a, b, tau, k, from_fstar=from_fstar)
        finally:
            self.plan.set_deferred_stream(False)
            if self.fused_events is not None:
                self.plan.set_fused_events(None, None)
        self._lazy = (fstar, scratch, (_version(fstar), _version(scratch), tau))
        self._carry = "lazy"
        flow._f, flow._f_next = None, None                    # nobody may see the buffers until the pass is done

        def finish():
            result = self.plan.stream(fstar, scratch)
            self._lazy = None
            self._carry = (_version(result), _version(fstar), tau)
            return result, fstar

        def drop():                                           # flow.f was assigned while the pass was pending
            self._lazy = None
            self._carry = None
        finish.drop = drop
        flow._pending = finish

    def _batch_resident(self, k: int, tau: float):
        """The same batch with the post-collision populations in the ENGINE's padded ping-pong buffers
        (``lt_resident_*``): ``flow.f`` / ``flow.f_next`` stay the reference's plain ``[q, *res]`` tensors
        (lettuce/_flow.py:90,124-134) and are written only by the pass that presents the populations when somebody
        looks.  k steps from ``flow.f`` are load (collide) + k - 1 fused steps; a batch that carries on -- nobody
        looked, or what was shown has not been touched since -- is k fused steps."""
        flow, plan = self.sim.flow, self.plan
        if self.fused_events is not None:
            plan.set_fused_events(*self.fused_events)
        try:
            if flow._pending is not None and self._carry == ("resident-pending", tau):
                plan.resident_advance(tau, k)
            else:
                f, nxt = self._state_buffers()                # reading flow.f completes a pending batch
                if self._carry == ("resident", _version(f), tau):
                    plan.resident_advance(tau, k)             # what was shown is still what the engine holds
                else:
                    plan.resident_load(f, tau)
                    plan.resident_advance(tau, k - 1)
                self._lazy = (f, nxt, None)                   # the two dense buffers of the flow
        finally:
            if self.fused_events is not None:
                plan.set_fused_events(None, None)
        self._carry = ("resident-pending", tau)
        flow._f, flow._f_next = None, None                    # nobody may see the buffers until the pass is done

        def finish():
            f, nxt, _ = self._lazy
            result = plan.resident_store(nxt)                 # the buffer shown last keeps its content
            self._lazy = None
            self._carry = ("resident", _version(result), tau)
            return result, f

        def drop():                                           # flow.f was assigned while the pass was pending
            self._lazy = None
            self._carry = None
            plan.resident_free()                              # the state they held is void: give the two fields back
        finish.drop = drop
        flow._pending = finish

    def single_step(self, *_, **__):
        self.batch(1)


class Simulation:
    flow: "Flow"
    context: "Context"
    collision: "Collision"
    boundaries: List["Boundary"]
    no_collision_mask: Optional[torch.Tensor]
    no_streaming_mask: Optional[torch.Tensor]
    reporter: List["Reporter"]

    def __init__(self, flow: "Flow", collision: "Collision", reporter: List["Reporter"]):
        self.flow = flow
        self.flow.collision = collision
        self.context = flow.context
        self.collision = collision
        self.reporter = reporter
        # index in no_collision_mask = position in this list; str() of the default repr
        # sorts by class path, as in the reference (lettuce/_simulation.py:57-58)
        self.boundaries = [None] + sorted(flow.boundaries, key=lambda b: str(b))

        self.no_collision_mask, self.no_streaming_mask = build_masks(flow, self.boundaries,
                                                                     self.context)

        def collide_and_stream(*_, **__):
            self._collide()
            self._stream()

        self._collide_and_stream = collide_and_stream
        self._native = None
        if self.context.use_native:
            self._native = _NativeStepper(self)       # raises when something has no kernel
            self._collide_and_stream = self._native.single_step

    def step(self, num_steps: int):
        warnings.warn("lt.Simulation.step() is deprecated and will be removed in a future "
                      "version. Instead, call simulation directly: simulation(num_steps)",
                      DeprecationWarning)
        return self(num_steps)

    @property
    def units(self):
        return self.flow.units

    # ---- non-native mode: the reference's whole-field expressions ------------------------------
    def _stream(self):
        """f_i(x) <- f_i(x - e_i), periodic; masked destinations keep their value
        (lettuce/_simulation.py:160-175)."""
        flow = self.flow
        axes = tuple(range(flow.stencil.d))
        for i in range(1, flow.stencil.q):
            moved = torch.roll(flow.f[i], shifts=tuple(flow.stencil.e[i]), dims=axes)
            if self.no_streaming_mask is not None:
                moved = torch.where(torch.eq(self.no_streaming_mask[i], 1), flow.f[i], moved)
            flow.f[i] = moved
        return flow.f

    def _collide(self):
        """collision where no_collision_mask == 0, then each boundary where the mask holds
        its index (lettuce/_simulation.py:177-189)."""
        flow, ncm = self.flow, self.no_collision_mask
        if ncm is None:
            flow.f = self.collision(flow)
            for boundary in self.boundaries[1:]:
                flow.f = boundary(flow)
            return flow.f
        torch.where(torch.eq(ncm, 0), self.collision(flow), flow.f, out=flow.f)
        for index, boundary in enumerate(self.boundaries[1:], start=1):
            torch.where(torch.eq(ncm, index), boundary(flow), flow.f, out=flow.f)
        return flow.f

    def _report(self):
        for reporter in self.reporter:
            reporter(self)

    # ---- the loop -----------------------------------------------------------------------------
    def _steps_to_next_report(self, limit: int) -> int:
        """Reporters are called after every step but library reporters act only when
        ``flow.i % interval == 0`` (lettuce/ext/_reporter/observable_reporter.py:185).  Steps
        in between can be fused into one engine batch -- for reporters that opt in with
        ``batchable = True`` (ObservableReporter, ErrorReporter).  Any other reporter, one without
        an integer ``interval`` or one with ``every_step = True`` is honoured after each step, and
        relaxation time and boundary parameters are then re-read every step, as in the reference."""
        k = limit
        for r in self.reporter:
            interval = getattr(r, "interval", None)
            if (not getattr(r, "batchable", False) or getattr(r, "every_step", False)
                    or not isinstance(interval, (int, np.integer)) or interval < 1):
                return 1
            k = min(k, interval - self.flow.i % interval)
        return max(1, k)

    def __call__(self, num_steps):
        beg = timer()
        if self.flow.i == 0:
            self._report()
        native = self._native
        if native is not None and self._collide_and_stream == native.single_step:
            remaining = int(num_steps)
            while remaining > 0:
                k = self._steps_to_next_report(remaining)
                native.batch(k)
                self.flow.i += k
                remaining -= k
                self._report()
            torch.cuda.synchronize(self.context.device)   # MLUPS must include the device work
        else:
            for _ in range(num_steps):
                self._collide_and_stream(self)
                self.flow.i += 1
                self._report()
        end = timer()
        return num_steps * int(np.prod(self.flow.resolution)) / 1e6 / (end - beg)
