"""Multi-GPU: 1-D slab decomposition along z, one process per GPU.

The reference has no distributed code at all; this module is the
build's addition, designed for MI355X nodes: RCCL point-to-point over xGMI, every rank talking
to its two z-neighbours only.

Layout.  A rank stores its slab as ``[q][nz_local + 2][ny][nx]`` (x fastest, z slowest --
``LT_LAYOUT_SLAB`` of include/lettuce_hip.h) with one ghost plane below and above.  With z
slowest a ghost plane of one population is a single contiguous ``ny*nx`` block; the crossing
populations of a boundary plane are gathered into one message per direction by the same launch
that computes the plane.

Transports (``transport=``):
  "rccl"    batched isend/irecv through the process group (RCCL point-to-point over xGMI; gloo in
            the CPU tests).  The default.
  "copy"    the halo messages travel by device-to-device copies that use no compute unit (SDMA engines over xGMI)
            into receive windows the neighbours mapped through HIP IPC; arrival is a counter word written in stream
            order behind the copy and polled by one wave on the receiver (``_CopyWindow``).  Nothing of it needs
            LDS or a workgroup slot beside the sweep.
  "window"  one-sided: every rank exposes a receive window (torch symmetric memory: peer-mapped
            device memory + signal pads); the boundary-plane launch stores the crossing
            populations straight into the neighbours' windows over xGMI, a signal follows, the
            receiver waits for the signal and unpacks.  No copy kernel competes with the interior
            launch for HBM.  Two window parities make the reuse safe without a barrier (see
            ``_PeerWindow``).

Schedule of one fused step (pull scheme, state = post-collision populations f*):
  communication stream (high priority)          compute stream
  1. stream-collide the two boundary planes     3. stream-collide the interior planes
     (1 and nz_local): what the neighbours need    (interior nodes never read a ghost plane)
  2. pack the crossing populations (e_z = -1 of
     plane 1, e_z = +1 of plane nz_local: 5 + 5
     of 19 for D3Q19) into one buffer each, send
     them to the lower / upper neighbour, receive
     theirs, unpack into the ghost planes
  4. both streams join before the next step.
There is no collective on the step path; only observables use an all-reduce.
"""
import os
from timeit import default_timer as timer
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .util import LettuceException

__all__ = ["ZSlab", "SlabSimulation", "TwoStepSlabSimulation", "SlabKineticEnergy", "SlabEnstrophy", "SlabMass"]


class ZSlab:
    """Which z-planes of the global grid this rank owns.

    ``halo`` extra planes on both sides are carried only while the initial condition is built
    (the 6th-order finite differences of the TGV f_neq initialisation reach 3 planes), see
    ``extended_resolution`` and the ``slab=`` argument of the flows."""

    def __init__(self, global_resolution: List[int], rank: Optional[int] = None,
                 world_size: Optional[int] = None, halo: int = 3):
        if len(global_resolution) != 3:
            raise LettuceException("z-slab decomposition needs a 3-D grid")
        if rank is None or world_size is None:
            if dist.is_available() and dist.is_initialized():
                rank, world_size = dist.get_rank(), dist.get_world_size()
            else:
                rank, world_size = 0, 1
        nx, ny, nz = (int(n) for n in global_resolution)
        if nz % world_size != 0:
            raise LettuceException(f"nz = {nz} is not divisible by the {world_size} ranks")
        self.global_resolution = [nx, ny, nz]
        self.rank, self.world_size, self.halo = rank, world_size, halo
        self.nz_local = nz // world_size
        self.z_begin = rank * self.nz_local
        self.prev = (rank - 1) % world_size
        self.next = (rank + 1) % world_size

    @property
    def local_resolution(self):
        nx, ny, _ = self.global_resolution
        return [nx, ny, self.nz_local]

    @property
    def extended_resolution(self):
        nx, ny, _ = self.global_resolution
        return [nx, ny, self.nz_local + 2 * self.halo]

    def z_indices(self, device=None):
        """global z index of every plane of the extended slab (periodic)"""
        nz = self.global_resolution[2]
        return (torch.arange(-self.halo, self.nz_local + self.halo, device=device)
                + self.z_begin) % nz


_COMM_STREAMS = {}


def _comm_stream(device, priority):
    """One communication stream per device and priority for the whole process.  torch hands out its
    pooled streams round-robin and HIP maps them onto a few hardware queues; a fresh stream per
    simulation made some instances 30-45 % slower than others in the same process (the boundary /
    exchange work no longer overlapped the interior launch), see tools/slab_history_probe.py."""
    priority = int(os.environ.get("LT_SLAB_COMM_PRIORITY", priority))
    key = (str(device), int(priority))
    if key not in _COMM_STREAMS:
        _COMM_STREAMS[key] = torch.cuda.Stream(device=device, priority=priority)
    return _COMM_STREAMS[key]


def _copy_streams(device, n):
    """the streams the copy transport issues its two directions on (one pair per device for the whole process, like
    the communication stream); n = 1: both directions on one stream"""
    if n == 1:
        return [None, None]                      # the stream that is current when send() is called
    out = []
    for k in (0, 1):
        key = (str(device), "copy", k)
        if key not in _COMM_STREAMS:
            _COMM_STREAMS[key] = torch.cuda.Stream(device=device, priority=-1)
        out.append(_COMM_STREAMS[key])
    return out


def _crossing_sets(stencil):
    e = np.array(stencil.e)
    up = [int(q) for q in np.nonzero(e[:, 2] == 1)[0]]     # move to +z: fill the lower ghost
    down = [int(q) for q in np.nonzero(e[:, 2] == -1)[0]]  # move to -z: fill the upper ghost
    return up, down


class _PeerWindow:
    """Receive windows for the one-sided ghost-plane transfer.

    ``local[p, 0]`` receives the e_z = +1 populations from the lower neighbour (they fill my lower
    ghost plane), ``local[p, 1]`` the e_z = -1 populations from the upper neighbour; p is the
    parity of the exchange counter.  A neighbour can only write parity p again two exchanges
    later, i.e. after it has waited for my signal of the exchange in between, which I raise after
    (in stream order) my unpack of parity p: no barrier is needed.  Signals: channel 0 travels
    upwards with the +z data, channel 1 downwards with the -z data.
    """

    WAIT_MS = 20000       # a lost signal traps instead of spinning for ever

    def __init__(self, shape, dtype, device, slab: ZSlab, group):
        import torch.distributed._symmetric_memory as symm
        self.slab = slab
        full = (2, 2) + tuple(shape)
        with torch.cuda.device(device):
            self.buf = symm.empty(full, dtype=dtype, device=device)
            self.buf.zero_()
            self.handle = symm.rendezvous(self.buf, group if group is not None else dist.group.WORLD)
            self.local = self.buf
            self.at_prev = self.handle.get_buffer(slab.prev, full, dtype, 0)
            self.at_next = self.handle.get_buffer(slab.next, full, dtype, 0)
            torch.cuda.synchronize(device)
            self.handle.barrier(2, self.WAIT_MS)
        self.count = 0

    def targets(self):
        """(where my -z message goes, where my +z message goes) for the coming exchange"""
        p = self.count & 1
        return self.at_prev[p, 1], self.at_next[p, 0]

    def signal(self):
        """after the stores of this exchange (same stream): tell both neighbours"""
        s, h = self.slab, self.handle
        h.put_signal(s.prev, 1, self.WAIT_MS)
        h.put_signal(s.next, 0, self.WAIT_MS)

    def wait(self):
        """wait for both neighbours' signals; returns (message from above, message from below)"""
        p, s, h = self.count & 1, self.slab, self.handle
        h.wait_signal(s.next, 1, self.WAIT_MS)
        h.wait_signal(s.prev, 0, self.WAIT_MS)
        self.count += 1
        return self.local[p, 1], self.local[p, 0]


class _DevicePointer:
    """a raw device pointer as something ``torch.as_tensor`` can wrap without copying"""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


class _CopyWindow:
    """Receive windows for the halo transport that needs no compute unit (``transport="copy"``).

    Every rank allocates one block of device memory the other processes of the node can map (``lt_ipc_alloc``):
    2 parities x 2 directions of halo messages, and -- in fine-grained memory -- one 64-bit arrival counter per
    direction and a time-out word.  The ranks exchange the 64-byte handles through the process group and open those
    of their two z-neighbours.  An
    exchange is then, on the sender's side: two device-to-device copies WITHOUT compute units (``lt_halo_copy``: an
    SDMA engine, over xGMI) from the buffers the edge launch wrote into the neighbours' windows, each followed by
    the write of the exchange counter into the neighbour's arrival word (``lt_flag_write``: a stream memory
    operation); and on the receiver's: one polling wave per direction (``lt_flag_wait``; no LDS, gives up after a
    second and says so).  RCCL's copy kernel -- 64 workgroups with 20 KB of LDS each, which cannot share a compute
    unit with a 150 KB sweep workgroup -- is not involved.

    ``slot(p, 0)`` receives what the lower neighbour sends upwards, ``slot(p, 1)`` what the upper one sends downwards;
    p is the parity of the exchange counter.  A neighbour writes parity p again two exchanges later, after it has
    seen my counter of the exchange in between, which I write (in stream order) after the launch that read parity p
    has finished: no barrier is needed (the argument of ``_PeerWindow``).

    ``engine``: 0 = the runtime's device-to-device copy (default): to a PEER's window that is an SDMA engine over xGMI;
    when source and destination are the same device -- the one-GPU rehearsal -- a blit kernel without LDS that shares
    the compute units with the sweep (20 MB in ~20 us); 1 = ask for a copy without compute units everywhere
    (hipMemcpyDeviceToDeviceNoCU).  Measured in the rehearsal (one MI355X, 512 x 512 x 64, ms per step, same box;
    profiles/r04c_slab_copy_transport.json): RCCL to self 0.3212, engine 0 0.3064, engine 1 0.322-0.327 with a
    scatter of 4 % between batches -- an SDMA engine moves a 20 MB message within one device at ~57 GB/s (0.35 ms),
    two of them beside a sweep that saturates HBM disturb it more than RCCL's kernel does."""

    def __init__(self, shape, dtype, device, slab: ZSlab, group, engine: int = 0, flag_how: int = 1,
                 streams: Optional[int] = None):
        import ctypes
        from ._native import load_library
        self.lib = load_library()
        self.slab, self.device, self.dtype = slab, torch.device(device), dtype
        self.shape = tuple(int(v) for v in shape)
        self.engine, self.flag_how = int(engine), int(flag_how)
        esize = torch.empty((), dtype=dtype).element_size()
        self.msg_bytes = int(np.prod(self.shape)) * esize
        self.msg_stride = -(-self.msg_bytes // 256) * 256
        total = 4 * self.msg_stride
        self._opened = []

        def allocate(n_bytes, fine):
            base, handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
            with torch.cuda.device(self.device):
                code = self.lib.lt_ipc_alloc(n_bytes, int(fine), ctypes.byref(base), handle)
                if code != 0 and fine:                    # no fine-grained memory to be had: ordinary memory
                    code = self.lib.lt_ipc_alloc(n_bytes, 0, ctypes.byref(base), handle)
                self._check(code)
            return int(base.value), bytes(handle.raw)
        # the messages, and -- in memory that stays coherent with a writer outside the running kernel -- the arrival
        # counters (2 x 8 bytes) and the time-out word
        self.base, data_handle = allocate(total, False)
        self.flags, flag_handle = allocate(256, True)
        if slab.world_size == 1:
            self.at_prev = self.at_next = self.base          # my own neighbour: no mapping needed
            self.flags_prev = self.flags_next = self.flags
        else:
            # (nobody writes into a window before its owner has zeroed it: lt_ipc_alloc does so, synchronously, before
            # the handles are published)
            handles = [None] * slab.world_size
            dist.all_gather_object(handles, (data_handle, flag_handle), group=group)

            def mapped(handle):
                out = ctypes.c_void_p()
                with torch.cuda.device(self.device):
                    self._check(self.lib.lt_ipc_open(handle, ctypes.byref(out)))
                self._opened.append(int(out.value))
                return int(out.value)
            self.at_prev, self.flags_prev = mapped(handles[slab.prev][0]), mapped(handles[slab.prev][1])
            if slab.next == slab.prev:
                self.at_next, self.flags_next = self.at_prev, self.flags_prev
            else:
                self.at_next, self.flags_next = mapped(handles[slab.next][0]), mapped(handles[slab.next][1])
        typestr = {torch.float32: "<f4", torch.float64: "<f8"}[dtype]
        self._local = [[torch.as_tensor(_DevicePointer(self.base + (2 * p + d) * self.msg_stride, self.shape, typestr),
                                        device=self.device) for d in (0, 1)] for p in (0, 1)]
        self._timed_out = torch.as_tensor(_DevicePointer(self.flags + 64, (1,), "<i4"), device=self.device)
        self.count = 0
        self.engines_used = set()
        # Both directions on the communication stream by default.  A stream of their own each lets two copy engines
        # work side by side (an SDMA engine moves 20 MB in ~0.35 ms), but every further active hardware queue costs
        # the sweep: in the one-GPU rehearsal two extra high-priority streams made the copy candidate 0.37 instead
        # of 0.30 ms per step whenever RCCL's streams existed in the process too (tools/slab_order_probe.py,
        # profiles/r04g_slab_stream_count.jsonl; GPU_MAX_HW_QUEUES = 8 / 16 changed nothing, 2 gave 0.295).
        # LT_SLAB_COPY_STREAMS=2 to A/B on real links.
        # ``streams=2`` (the driver's ``copy_streams``) is the same choice per simulation: bench.py --gpus N offers
        # both as candidates, because across real links two messages through ONE stream travel one after the other.
        two = (int(streams) == 2) if streams is not None else os.environ.get("LT_SLAB_COPY_STREAMS", "1") == "2"
        self._streams = _copy_streams(self.device, 2 if two else 1)
        self.n_streams = 2 if two else 1

    def _check(self, code):
        if code != 0:
            raise LettuceException(f"copy transport: {self.lib.lt_last_error().decode('utf-8', 'replace')}")

    def send(self, send_down: torch.Tensor, send_up: torch.Tensor):
        """behind what the current stream holds so far: my downward message into the lower neighbour's slot (p, 1), my
        upward one into the upper neighbour's slot (p, 0), each followed by the counter of this exchange -- on the
        current stream, or (LT_SLAB_COPY_STREAMS=2) on a stream of its own per direction, after which the current
        stream continues when both are through (the launch that overwrites the send buffers two double steps later
        is ordered behind it)."""
        import ctypes
        p, cur = self.count & 1, torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)
        used = ctypes.c_int32(0)
        for side, (base, flags, d, msg) in zip(self._streams, ((self.at_prev, self.flags_prev, 1, send_down),
                                                               (self.at_next, self.flags_next, 0, send_up))):
            if side is None:
                side = cur
            else:
                side.wait_event(ready)
            stream = ctypes.c_void_p(side.cuda_stream)
            self._check(self.lib.lt_halo_copy(ctypes.c_void_p(base + (2 * p + d) * self.msg_stride),
                                              ctypes.c_void_p(msg.data_ptr()), self.msg_bytes, self.engine, stream,
                                              ctypes.byref(used)))
            self.engines_used.add("copy engine (no compute units)" if used.value else "runtime's device-to-device copy")
            self._check(self.lib.lt_flag_write(ctypes.c_void_p(flags + 8 * d), self.count + 1, self.flag_how, stream))
            if side is not cur:
                cur.wait_stream(side)

    def wait(self):
        """on the current stream: wait for both neighbours' messages of this exchange; returns (message from above,
        message from below)"""
        import ctypes
        p, stream = self.count & 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for d in (1, 0):
            self._check(self.lib.lt_flag_wait(ctypes.c_void_p(self.flags + 8 * d), self.count + 1,
                                              ctypes.c_void_p(self._timed_out.data_ptr()), stream))
        self.count += 1
        return self._local[p][1], self._local[p][0]

    def timed_out(self) -> bool:
        """did a polling wave give up since the last call?  (synchronises)"""
        flag = bool(int(self._timed_out.item()))
        if flag:
            self._timed_out.zero_()
        return flag

    def close(self):
        if getattr(self, "base", None):
            try:
                torch.cuda.synchronize(self.device)
                with torch.cuda.device(self.device):
                    for ptr in self._opened:
                        self.lib.lt_ipc_close(ptr)
                    self.lib.lt_ipc_free(self.base)
                    self.lib.lt_ipc_free(self.flags)
            finally:
                self.base, self.flags, self._opened = 0, 0, []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlabKineticEnergy:
    """IncompressibleKineticEnergy of the WHOLE domain for ``ObservableReporter`` on a slab driver: a rank
    holds only its slab (``flow.f`` is None there), so the library's observables, which read ``flow.f``,
    cannot be used; this one asks the driver (device reduction on the slab + all-reduce)."""
    slab_aware = True

    def __init__(self, flow):
        self.context = flow.context
        self.flow = flow
        self.simulation = None              # bound by the slab driver

    def __call__(self, f=None):
        if self.simulation is None:
            raise LettuceException("SlabKineticEnergy is evaluated through a slab driver's reporter")
        return torch.tensor(self.simulation.kinetic_energy_pu(), dtype=torch.float64)


class SlabEnstrophy(SlabKineticEnergy):
    """Enstrophy of the WHOLE domain (observable_reporter.py:45-68) through a slab driver: every rank computes the
    velocity of its planes, swaps three planes with either neighbour (the 6th-order differences reach that far),
    reduces its own nodes on the device and the ranks' sums are all-reduced -- no gather."""

    def __call__(self, f=None):
        if self.simulation is None:
            raise LettuceException("SlabEnstrophy is evaluated through a slab driver's reporter")
        return torch.tensor(self.simulation.enstrophy_pu(), dtype=torch.float64)


class SlabMass(SlabKineticEnergy):
    """The reference's Mass observable (observable_reporter.py:140-158) of the WHOLE domain through a slab driver.
    ``no_mass_mask``: boolean mask on the flow's extended slab (as the flow's own masks), or None."""

    def __init__(self, flow, no_mass_mask=None):
        super().__init__(flow)
        self.mask = no_mass_mask

    def __call__(self, f=None):
        if self.simulation is None:
            raise LettuceException("SlabMass is evaluated through a slab driver's reporter")
        return torch.tensor(self.simulation.mass_interior(self.mask), dtype=torch.float64)


class SlabSimulation:
    """Time-step driver of one rank's slab.

    ``flow`` is built on ``slab.extended_resolution`` with ``slab=slab`` (so that its initial
    condition and its boundary masks equal the global ones on this rank's planes); ``collision``
    is a BGK / KBC / NoCollision object.  Boundaries may be bounce-back, equilibrium (uniform or with
    per-node velocity / pressure given on the extended slab) and an anti-bounce-back outlet along any axis
    (along z it lives on the rank that holds the first / last plane of the global grid, together with the
    plane next to it).  ``engine`` defaults to the HIP engine; tests inject a
    CPU stand-in with the same three ``*_planes`` methods to exercise the decomposition and the
    exchange with the gloo backend.
    """

    GHOST = 1            # ghost planes per side
    ONE_STREAM_WINDOWS = False
    # elements between the end of one population and the start of the next in the slab tensors (HIP engine only).
    # The one-step kernels stream best from dense populations (tools/pad_sweep_probe.py: padding costs them 2-4 %),
    # the two-step kernels from padded ones: TwoStepSlabSimulation overrides this.
    POPULATION_PAD = 0

    def __init__(self, flow, collision, slab: ZSlab, reporter=None, engine=None, group=None,
                 overlap: bool = True, comm_priority: int = -1, transport: str = "rccl",
                 copy_streams: Optional[int] = None):
        if list(flow.resolution) != slab.extended_resolution:
            raise LettuceException(f"flow resolution {flow.resolution} != extended slab "
                                   f"{slab.extended_resolution}")
        if slab.halo < self.GHOST:
            raise LettuceException(f"ZSlab.halo = {slab.halo} planes, the driver needs {self.GHOST} ghost planes")
        self.flow, self.collision, self.slab = flow, collision, slab
        self.context = flow.context
        self.reporter = reporter if reporter is not None else []
        for r in self.reporter:
            # a rank holds its slab only: reporters whose observable reads flow.f cannot work here
            obs = getattr(r, "observable", None)
            if not getattr(r, "slab_aware", False) and not getattr(obs, "slab_aware", False):
                raise LettuceException(
                    f"reporter {type(r).__name__} reads flow.f, which a slab rank does not hold; use an "
                    f"observable that asks the driver (e.g. SlabKineticEnergy) or mark the reporter slab_aware")
            if obs is not None and getattr(obs, "slab_aware", False):
                obs.simulation = self
        self.group = group
        self.i = 0
        flow.i = 0
        self.overlap = overlap and self.context.device.type == "cuda"
        nx, ny, _ = slab.global_resolution
        nzl, h, g = slab.nz_local, slab.halo, self.GHOST
        self.nzl = nzl
        self.lo, self.hi = g, g + nzl               # interior planes [lo, hi) of the slab tensor
        self.up, self.down = _crossing_sets(flow.stencil)
        desc = collision.native_generator()
        self._tau = desc.tau
        # boundaries: same ordering and masks as Simulation (built on the extended slab, then cut
        # to this rank's planes + one ghost plane per side and laid out z-slowest)
        from ._simulation import build_masks
        self.boundaries = [None] + sorted(flow.boundaries, key=lambda b: str(b))
        ncm, nsm = build_masks(flow, self.boundaries, self.context)
        entries = []
        for i, b in enumerate(self.boundaries[1:], start=1):
            if not b.native_available():
                raise LettuceException(f"boundary '{type(b).__name__}' has no engine kernel")
            entry = b.native_generator(i).plan_entry(flow)
            if "field" in entry:
                # per-node boundary arguments (equilibrium_boundary_pu.py:21-40) were given on the extended
                # slab like everything else of the flow: cut this rank's planes + ghost planes, z slowest
                entry = dict(entry)
                entry["field"] = entry["field"][..., h - g:h + nzl + g].permute(0, 3, 2, 1).contiguous()
            entries.append(entry)
        if ncm is not None:
            ncm = ncm[..., h - g:h + nzl + g].permute(2, 1, 0).contiguous()
            nsm = nsm[..., h - g:h + nzl + g].permute(0, 3, 2, 1).contiguous()
        self.no_collision_mask, self.no_streaming_mask = ncm, nsm
        if engine is None:
            from ._native import Plan, LAYOUT_SLAB
            engine = Plan(type(flow.stencil).__name__, self.context.dtype, desc.kind,
                          slab.local_resolution, entries, layout=LAYOUT_SLAB, ghost_planes=g,
                          device=self.context.device)
            if ncm is not None:
                engine.set_masks(ncm, nsm)
        elif entries:
            if hasattr(engine, "ghosts"):
                engine.ghosts = g
            engine.set_boundaries(entries, ncm, nsm, flow.units)      # test stand-ins
        self.engine = engine
        # [q, nx, ny, nzl + 2g] incl. the ghost planes -> [q, nzl + 2g, ny, nx]
        core = flow.f[..., h - g:h + nzl + g]
        # The slab tensors are this driver's own (the reference never sees them), so the populations need not be
        # dense: with the HIP engine they are ``pad`` elements further apart than their size (one allocation,
        # strided views), which keeps the q streams of a node out of each other's memory channels
        # (lt_plan_set_population_stride; LT_SLAB_PAD overrides, 0 = dense)
        pad = int(os.environ.get("LT_SLAB_PAD", str(self.POPULATION_PAD)))
        if pad > 0 and hasattr(engine, "set_population_stride") and core.is_cuda:
            nodes = (nzl + 2 * g) * ny * nx
            unit = 256 // core.element_size()
            engine.set_population_stride(-(-(nodes + pad) // unit) * unit)
            self.f = engine.populations_like(core.permute(0, 3, 2, 1))
            self.f_next = engine.empty_populations()
        else:
            self.f = core.permute(0, 3, 2, 1).contiguous()
            self.f_next = torch.empty_like(self.f)
        flow.f = None                       # the extended slab is not needed any more
        flow._f_next = None
        # one contiguous message per direction: [n_crossing, ny, nx]
        shape = [self._message_blocks(flow.stencil), ny, nx]
        new = lambda: torch.empty(shape, dtype=self.f.dtype, device=self.f.device)   # noqa: E731
        self._send_up, self._send_down, self._recv_up, self._recv_down = new(), new(), new(), new()
        self._comm = _comm_stream(self.context.device, comm_priority) if self.overlap else None
        # rehearsal switch: send to / receive from oneself through the process group even with a
        # single rank, to exercise the point-to-point path on a one-GPU box
        self._force_p2p = (os.environ.get("LT_SLAB_FORCE_P2P") == "1" and dist.is_available()
                           and dist.is_initialized())
        self._host_transport = ((slab.world_size > 1 or self._force_p2p)
                                and dist.get_backend(group) != "nccl")
        if transport not in ("rccl", "window", "copy"):
            raise LettuceException(f"unknown slab transport '{transport}'")
        self.transport = transport
        self._window = None
        self._cw = None
        if transport == "copy":
            if self.context.device.type != "cuda" or (slab.world_size > 1 and not (dist.is_available() and dist.is_initialized())):
                raise LettuceException("the copy transport needs device memory (and an initialised process group "
                                       "with more than one rank)")
            self._cw = _CopyWindow(shape, self.f.dtype, self.f.device, slab, group,
                                   engine=int(os.environ.get("LT_SLAB_COPY_ENGINE", "0")),
                                   flag_how=int(os.environ.get("LT_SLAB_FLAG_HOW", "1")), streams=copy_streams)
        if transport == "window":
            if self.context.device.type != "cuda" or not (dist.is_available() and dist.is_initialized()):
                raise LettuceException("the window transport needs device memory and an initialised "
                                       "process group")
            self._window = _PeerWindow(shape, self.f.dtype, self.f.device, slab, group)
        # window transport: the exchange can ride on the compute stream (see _fused_step); measured
        # on one MI355X the single-step driver is better off with two streams (0.437 vs 0.454
        # ms/step), the two-step driver marginally with one (0.356 vs 0.361) -- see
        # TwoStepSlabSimulation.ONE_STREAM_WINDOWS for why two streams are the default anyway
        self._one_stream = self._window is not None and self.overlap and self.ONE_STREAM_WINDOWS

    def _message_blocks(self, stencil) -> int:
        return len(self.up)

    # ---- the populations -------------------------------------------------------------------------
    # ``f`` holds post-streaming populations (lettuce's convention) whenever somebody looks; between batches
    # the driver keeps the post-collision populations of the last step (ghost planes exchanged) and carries on
    # from them, and the streaming pass that presents ``f`` runs when ``f`` / ``f_next`` are read (local_f,
    # gather_f, the observables).  A batch that nobody looks at costs its fused steps only.
    _pending = None          # (f*, scratch) while ``f`` is one streaming pass short
    _carry = None            # version of ``_f`` when ``_f_next`` held the f* it was streamed from
    _f = None
    _f_next = None

    @property
    def f(self) -> torch.Tensor:
        if self._pending is not None:
            self._present()
        return self._f

    @f.setter
    def f(self, value):
        self._pending, self._carry = None, None
        self._f = value

    @property
    def f_next(self) -> torch.Tensor:
        if self._pending is not None:
            self._present()
        return self._f_next

    @f_next.setter
    def f_next(self, value):
        if self._pending is not None:
            self._present()
        self._carry = None
        self._f_next = value

    def _present(self):
        from ._simulation import _version
        (cur, nxt), self._pending = self._pending, None
        self.engine.stream_planes(cur, nxt, self.lo, self.hi)
        self._f, self._f_next = nxt, cur
        self._carry = (_version(nxt), _version(cur))      # both buffers: f* lives in the second one

    def _start_batch(self, tau):
        """(f*, scratch, fused steps already owed): the post-collision populations to carry on from -- kept from
        the batch before, or one collide pass + exchange away from ``f``"""
        from ._simulation import _version
        if self._pending is not None:
            (cur, nxt), self._pending = self._pending, None
            return cur, nxt, True
        if self._carry is not None and self._carry == (_version(self._f), _version(self._f_next)):
            self._carry = None
            return self._f_next, self._f, True
        self._carry = None
        cur, nxt = self._f, self._f_next
        self.engine.collide_planes(cur, nxt, tau, self.lo, self.hi)
        self._exchange(nxt)()
        return nxt, cur, False

    # ---- views ---------------------------------------------------------------------------------
    def local_f(self) -> torch.Tensor:
        """this rank's populations as a ``[q, nx, ny, nz_local]`` view (reference axis order)"""
        return self.f[:, self.lo:self.hi].permute(0, 3, 2, 1)

    def gather_f(self, dst: int = 0) -> Optional[torch.Tensor]:
        """global ``[q, nx, ny, nz]`` tensor on rank ``dst`` (small grids / tests only)"""
        local = self.local_f().contiguous()
        if self.slab.world_size == 1:
            return local
        parts = ([torch.empty_like(local) for _ in range(self.slab.world_size)]
                 if self.slab.rank == dst else None)
        if dist.get_backend(self.group) == "nccl":
            full = [torch.empty_like(local) for _ in range(self.slab.world_size)]
            dist.all_gather(full, local, group=self.group)
            parts = full if self.slab.rank == dst else None
        else:
            dist.gather(local, parts, dst=dst, group=self.group)
        return torch.cat(parts, dim=3) if parts is not None else None

    # ---- halo exchange -------------------------------------------------------------------------
    def _pack(self, buf, plane, direction, out):
        if hasattr(self.engine, "pack"):
            self.engine.pack(buf, plane, direction, out)
        else:
            out.copy_(buf[self.up if direction > 0 else self.down, plane])

    def _unpack(self, buf, plane, direction, src):
        if hasattr(self.engine, "unpack"):
            self.engine.unpack(buf, plane, direction, src)
        else:
            buf[self.up if direction > 0 else self.down, plane] = src

    def _exchange(self, buf: torch.Tensor, packed: bool = False):
        """Fill the ghost planes of ``buf`` (post-collision populations) from the neighbours:
        the e_z = -1 populations of my plane 1 go to the lower neighbour's upper ghost plane, the
        e_z = +1 populations of my top plane to the upper neighbour's lower ghost plane.  One
        packed message per direction.  Returns a callable that completes the exchange (waits
        for the transfers and unpacks)."""
        nzl, s = self.nzl, self.slab
        if self._cw is not None:
            if not packed:
                self._pack(buf, 1, -1, self._send_down)
                self._pack(buf, nzl, +1, self._send_up)
            self._cw.send(self._send_down, self._send_up)

            def finish_copy():
                from_above, from_below = self._cw.wait()
                self._unpack(buf, nzl + 1, -1, from_above)
                self._unpack(buf, 0, +1, from_below)
            return finish_copy
        nzl, s = self.nzl, self.slab
        if self._window is not None:
            if not packed:
                to_prev, to_next = self._window.targets()
                self._pack(buf, 1, -1, to_prev)
                self._pack(buf, nzl, +1, to_next)
            self._window.signal()

            def finish_window():
                from_above, from_below = self._window.wait()
                self._unpack(buf, nzl + 1, -1, from_above)
                self._unpack(buf, 0, +1, from_below)
            return finish_window
        if not packed:
            self._pack(buf, 1, -1, self._send_down)
            self._pack(buf, nzl, +1, self._send_up)
        if s.world_size == 1 and not self._force_p2p:
            recv_down, recv_up, reqs = self._send_down, self._send_up, []
        else:
            recv_down, recv_up = self._recv_down, self._recv_up
            if self._host_transport and buf.is_cuda:
                # gloo moves device tensors from the host side without stream semantics (only
                # used by tests that put two ranks on one GPU): make the packed data visible
                torch.cuda.current_stream().synchronize()
            ops = [dist.P2POp(dist.isend, self._send_down, s.prev, self.group, tag=1),
                   dist.P2POp(dist.irecv, recv_down, s.next, self.group, tag=1),
                   dist.P2POp(dist.isend, self._send_up, s.next, self.group, tag=2),
                   dist.P2POp(dist.irecv, recv_up, s.prev, self.group, tag=2)]
            reqs = dist.batch_isend_irecv(ops)

        def finish():
            for r in reqs:
                r.wait()
            if reqs and self._host_transport and buf.is_cuda:
                torch.cuda.current_stream().synchronize()
            self._unpack(buf, nzl + 1, -1, recv_down)
            self._unpack(buf, 0, +1, recv_up)
        return finish

    # ---- stepping --------------------------------------------------------------------------------
    def _boundary_planes(self, cur, nxt, tau) -> bool:
        """Stream-collide planes 1 and nz_local; returns True when the launch also packed the
        crossing populations into the send buffers."""
        if hasattr(self.engine, "stream_collide_plane_pair_packed"):
            down, up = ((self._send_down, self._send_up) if self._window is None
                        else self._window.targets())
            self.engine.stream_collide_plane_pair_packed(cur, nxt, tau, 1, self.nzl, down, up)
            return True
        self.engine.stream_collide_planes(cur, nxt, tau, 1, 2)
        if self.nzl > 1:
            self.engine.stream_collide_planes(cur, nxt, tau, self.nzl, self.nzl + 1)
        return False

    def _fused_step(self, cur, nxt, tau):
        """One stream-collide of the slab.  With overlap the two boundary planes, the halo packing,
        the transfers and the unpacking run on the (high-priority) communication stream while the
        compute stream does the interior planes; both only read ``cur`` and write disjoint planes
        of ``nxt``.  The streams join before the next step."""
        eng, nzl = self.engine, self.nzl
        if self.overlap and not self._one_stream:
            compute = torch.cuda.current_stream()
            self._comm.wait_stream(compute)          # previous step complete (it read nxt)
            with torch.cuda.stream(self._comm):
                packed = self._boundary_planes(cur, nxt, tau)
                self._exchange(nxt, packed)()
            if nzl > 2:
                eng.stream_collide_planes(cur, nxt, tau, 2, nzl)
            compute.wait_stream(self._comm)
        else:
            # one stream: boundary planes, start of the exchange, interior planes, end of the
            # exchange.  With the window transport the "transfer" is the packing launch's stores into
            # the neighbour's memory followed by a signal, so the neighbour's data arrives while the
            # interior launch runs and no second stream (and no hardware-queue pairing) is involved.
            packed = self._boundary_planes(cur, nxt, tau)
            finish = self._exchange(nxt, packed)
            if nzl > 2:
                eng.stream_collide_planes(cur, nxt, tau, 2, nzl)
            finish()

    def _advance(self, n: int):
        """n whole steps: (collide, exchange,) n - 1 or -- carrying on from the batch before -- n fused steps; the
        streaming pass that completes the last one runs when ``f`` is read (as Flow.f does on one GPU)."""
        tau = float(self._tau(self.flow))
        cur, nxt, carried = self._start_batch(tau)
        for _ in range(n if carried else n - 1):
            self._fused_step(cur, nxt, tau)
            cur, nxt = nxt, cur
        self._pending = (cur, nxt)                    # streamed when somebody looks (``f``)

    def _next_report(self, limit):
        k = limit
        for r in self.reporter:
            interval = getattr(r, "interval", None)
            if (not getattr(r, "batchable", False) or not isinstance(interval, (int, np.integer))
                    or interval < 1):
                return 1
            k = min(k, interval - self.i % interval)
        return max(1, k)

    def __call__(self, num_steps: int) -> float:
        """Advance; returns this rank's MLUPS (local nodes only)."""
        beg = timer()
        if self.i == 0:
            for r in self.reporter:
                r(self)
        remaining = int(num_steps)
        while remaining > 0:
            k = self._next_report(remaining)
            self._advance(k)
            self.i += k
            self.flow.i = self.i            # reporters read simulation.flow.i
            remaining -= k
            for r in self.reporter:
                r(self)
        if self.context.device.type == "cuda":
            torch.cuda.synchronize(self.context.device)
        elapsed = timer() - beg
        if self._cw is not None and self._cw.timed_out():
            # a polling wave gave up (it waits about a second): the launch behind it read a message that had not
            # arrived.  Raised here, after the exchanges of the call, so that the ranks stay in step
            raise LettuceException("copy transport: timed out waiting for a neighbour's halo message; the "
                                   "populations of this call are not valid")
        nx, ny, _ = self.slab.global_resolution
        return num_steps * nx * ny * self.nzl / 1e6 / elapsed

    @property
    def units(self):
        return self.flow.units

    # ---- observables -----------------------------------------------------------------------------
    def kinetic_energy_pu(self) -> float:
        """IncompressibleKineticEnergy of the whole domain (all-reduced over the ranks)."""
        flow, units = self.flow, self.flow.units
        if hasattr(self.engine, "kinetic_energy_lu"):
            total = self.engine.kinetic_energy_lu(self.f)
        else:
            f = self.local_f()
            rho = torch.sum(f, dim=0)
            j = torch.einsum("qd,q...->d...", flow.torch_stencil.e, f)
            u = j / rho
            total = torch.sum(0.5 * torch.einsum("d...,d...->...", u, u)).double()
        if self.slab.world_size > 1:
            total = total.clone()
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        dx = units.convert_length_to_pu(1.0)
        return float(units.convert_incompressible_energy_to_pu(total) * dx ** 3)


    def _all_reduced(self, total: torch.Tensor) -> torch.Tensor:
        if self.slab.world_size > 1:
            total = total.clone()
            if self._host_transport and total.is_cuda:
                host = total.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                return host
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        return total

    def _velocity_halo(self, u_ext: torch.Tensor):
        """Fill the three outer planes per side of ``u_ext`` [3, nz_local + 6, ny, nx] with the neighbours' planes
        (periodic ring): my first three planes go down, my last three up."""
        nzl, s = self.nzl, self.slab
        if nzl < 3:
            raise LettuceException(f"enstrophy on slabs needs at least 3 planes per rank, this one has {nzl}")
        first, last = u_ext[:, 3:6], u_ext[:, nzl:nzl + 3]
        if s.world_size == 1:
            u_ext[:, nzl + 3:] = first
            u_ext[:, :3] = last
            return
        host = self._host_transport and u_ext.is_cuda
        send_down, send_up = first.contiguous(), last.contiguous()
        if host:
            send_down, send_up = send_down.cpu(), send_up.cpu()
        from_above, from_below = torch.empty_like(send_down), torch.empty_like(send_up)
        ops = [dist.P2POp(dist.isend, send_down, s.prev, self.group, tag=11),
               dist.P2POp(dist.irecv, from_above, s.next, self.group, tag=11),
               dist.P2POp(dist.isend, send_up, s.next, self.group, tag=12),
               dist.P2POp(dist.irecv, from_below, s.prev, self.group, tag=12)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        u_ext[:, nzl + 3:] = from_above.to(u_ext.device)
        u_ext[:, :3] = from_below.to(u_ext.device)

    def enstrophy_pu(self) -> float:
        """Enstrophy of the whole domain (periodic flows, as the reference's observable)."""
        flow, units, g, nzl = self.flow, self.flow.units, self.GHOST, self.nzl
        f = self.f
        dx = units.convert_length_to_pu(1.0)
        scale = float(units.convert_velocity_to_pu(1.0))
        if hasattr(self.engine, "slab_velocity"):
            u_ext = self.engine.slab_velocity(f)
        else:                                   # test stand-ins (CPU): the same field with whole-field torch ops
            own = f[:, self.lo:self.hi].permute(0, 3, 2, 1)                     # [q, nx, ny, nzl]
            rho = torch.sum(own, dim=0)
            u = torch.einsum("qd,q...->d...", flow.torch_stencil.e, own) / rho  # [3, nx, ny, nzl]
            u_ext = torch.zeros([3, nzl + 6] + list(f.shape[2:]), dtype=f.dtype, device=f.device)
            u_ext[:, 3:nzl + 3] = u.permute(0, 3, 2, 1)
        self._velocity_halo(u_ext)
        if hasattr(self.engine, "slab_enstrophy_sum"):
            total = self.engine.slab_enstrophy_sum(u_ext, scale, 1.0 / dx)
        else:
            from .util import torch_gradient
            # x and y are periodic within the rank: torch_gradient's rolls are exact there; along z only the own
            # planes [3, nzl + 3) are kept, whose differences never reach past the three neighbour planes
            u_pu = (u_ext * scale).permute(0, 3, 2, 1)                          # [3, nx, ny, nzl + 6]
            grad = [torch_gradient(u_pu[a], dx=dx, order=6) for a in range(3)]
            w_z, w_x, w_y = grad[0][1] - grad[1][0], grad[2][1] - grad[1][2], grad[0][2] - grad[2][0]
            node = w_z * w_z + (w_x * w_x + w_y * w_y)
            total = torch.sum(node[..., 3:nzl + 3]).double()
        total = self._all_reduced(total)
        return float(total) * float(dx) ** 3

    def mass_interior(self, no_mass_mask=None) -> float:
        """The reference's Mass observable of the whole domain: all populations of the nodes off the first / last
        y and GLOBAL z index, minus those of the nodes ``no_mass_mask`` (given on the extended slab) flags."""
        s, g, nzl, h = self.slab, self.GHOST, self.nzl, self.slab.halo
        f = self.f
        mask = None
        if no_mass_mask is not None:
            mask = torch.broadcast_to(torch.as_tensor(no_mass_mask, device=f.device), self.slab.extended_resolution)
            mask = mask[..., h - g:h + nzl + g].permute(2, 1, 0).contiguous()      # [nzl + 2g, ny, nx]
        nz = s.global_resolution[2]
        if hasattr(self.engine, "slab_mass_interior"):
            total = self.engine.slab_mass_interior(f, s.z_begin, nz, mask)
        else:
            own = f[:, self.lo:self.hi]                                             # [q, nzl, ny, nx]
            z = torch.arange(s.z_begin, s.z_begin + nzl, device=f.device)
            inner = ((z > 0) & (z < nz - 1)).reshape(-1, 1, 1)
            node = own.double().sum(dim=0)
            total = (node[:, 1:-1] * inner).sum()
            if mask is not None:
                total = total - (node * mask[self.lo:self.hi].to(node.dtype)).sum()
        return float(self._all_reduced(total))


class TwoStepSlabSimulation(SlabSimulation):
    """Slab driver for the two-step kernel (``lt_stream_collide_twice_planes``): two lattice updates
    per launch and ONE halo exchange per two updates.  The engine decides which lattices / dtypes /
    grids it supports (nx % 64 == 0 in fp32, % 32 in fp64; ny % 8 == 0, D3Q27: % 4) and which
    boundaries: bounce-back and equilibrium nodes anywhere and one anti-bounce-back outlet along x
    opposite an inlet face of equilibrium nodes (the Obstacle); every rank asks its engine and all
    ranks raise together when one of them has no two-step launch (use ``SlabSimulation`` then, as for
    outlets along y or z).

    Two ghost planes per side.  Before a two-step launch the lower ghost planes must hold the
    in-plane and upward populations of the lower neighbour's top plane and the upward populations
    of the plane below it (mirrored above): 9 + 5 + 5 = 19 plane-populations per direction for
    D3Q19 instead of 2 x 5 for two single steps, in half as many messages.  With boundaries the
    message also carries the downward populations of that top plane (24 blocks): a node of the ghost
    plane with no-streaming bits keeps them.

    Schedule of one double step: the communication stream computes the two output planes next to
    each cut (two small launches), packs, transfers and unpacks; the compute stream does the
    interior output planes meanwhile; the streams join before the next launch.
    """

    GHOST = 2
    POPULATION_PAD = 32832       # 128 KiB + 256 B in fp32: the engine's choice for its own buffers too
    # With windows the exchange could ride on the compute stream (stores + signal before the interior
    # launch, wait + unpack after it); on one GPU that is 1.5 % faster, but across xGMI the pack
    # launch takes as long as the transfer and would delay the interior launch: two streams.
    ONE_STREAM_WINDOWS = False

    def __init__(self, flow, collision, slab: ZSlab, fused_remote_pack: Optional[bool] = None,
                 signalled: Optional[bool] = None, direct: Optional[bool] = None, **kwargs):
        """``fused_remote_pack`` (window transport only): True = the edge launches store the halo
        message into the neighbour's window themselves and the whole exchange rides on the compute
        stream; False (default, see ``_edges``) = a separate pack launch on the communication
        stream.  None reads LT_SLAB_FUSED_REMOTE_PACK / LT_SLAB_ONE_STREAM.

        ``signalled`` (RCCL transport, periodic flows, HIP engine): True = ONE launch per double step
        covers all interior planes; its edge workgroups start first and count themselves done on a device
        counter, and the communication stream -- one polling wave, then the pack launches, the transfers
        and the unpack launches -- works beside the rest of the sweep
        (``lt_stream_collide_twice_slab``).  Saves the two edge launches (planes computed twice, launches
        competing with the interior one).  None reads LT_SLAB_SIGNALLED (default off)."""
        if signalled is None:
            signalled = os.environ.get("LT_SLAB_SIGNALLED") == "1"
        self._signalled = bool(signalled)
        # ``direct`` (RCCL / gloo transport, periodic flows): a double step is TWO launches on the compute stream
        # -- the edge launch, which reads the planes beyond the cuts straight from the receive buffers and writes
        # the outgoing messages itself (``lt_stream_collide_twice_edges_direct``), then the sweep over the planes in
        # between, which never reads a ghost plane -- and one exchange on the communication stream that starts
        # when the edge launch has finished: no pack launches, no unpack launches, no launches that compete for
        # the compute units.  None reads LT_SLAB_DIRECT (default on where the engine has the launch).
        if direct is None:
            direct = os.environ.get("LT_SLAB_DIRECT", "1") == "1" and not self._signalled
        self._direct = bool(direct)
        self._ghost_src = None        # (message from below, message from above) while the ghost planes of the
        self._parity = 0              # current populations are still in the receive buffers (direct schedule)
        if fused_remote_pack is None:
            self._fused_remote = os.environ.get("LT_SLAB_FUSED_REMOTE_PACK") == "1"
            self.ONE_STREAM_WINDOWS = os.environ.get("LT_SLAB_ONE_STREAM") == "1"
        else:
            self._fused_remote = bool(fused_remote_pack)
            self.ONE_STREAM_WINDOWS = bool(fused_remote_pack)
        self._masked = bool(flow.boundaries)
        if slab.nz_local < 4:
            raise LettuceException("the two-step slab driver needs at least 4 planes per rank")
        # output planes per cut computed ahead of the interior (>= 2: the halo message reads two).
        # Measured on MI355X (512 x 512 x 64, self exchange): 2 / 4 / 8 planes give 0.364 / 0.369 /
        # 0.367 ms per step with peer windows -- no reason to delay the exchange.
        self.edge_planes = max(2, int(os.environ.get("LT_SLAB_EDGE_PLANES", "2")))
        super().__init__(flow, collision, slab, **kwargs)
        self._send2 = None
        # every rank must take the same path: agree on whether all engines have a two-step launch
        why = self.engine.two_step_admitted() if hasattr(self.engine, "two_step_admitted") else None
        refused = torch.tensor([0 if why is None else 1], dtype=torch.int32)
        if slab.world_size > 1:
            backend = dist.get_backend(self.group)
            refused = refused.to(self.f.device if backend == "nccl" else "cpu")
            dist.all_reduce(refused, op=dist.ReduceOp.MAX, group=self.group)
        if int(refused.item()):
            raise LettuceException("the two-step slab driver cannot run this flow"
                                   + (f": {why}" if why else " (refused on another rank)"))
        if self._masked and flow.stencil.q == 27:
            import warnings
            warnings.warn("TwoStepSlabSimulation with boundaries on D3Q27: measured slower than SlabSimulation "
                          "(0.97 vs 0.83 ms per step at 512 x 512 x 64, DESIGN.md section 5)", stacklevel=2)
        # A two-step workgroup holds a CU's LDS for its whole segment, and RCCL's copy kernel needs
        # LDS of its own: beside one long interior segment per CU it starts only when the first
        # workgroups retire.  Shorter segments let it in earlier: a quarter of the interior planes
        # (15 of 60) measured 0.357 vs 0.369 ms/step in the self-exchange rehearsal, 10 and 6 planes
        # are slower again (0.371, 0.392); LT_SLAB_RCCL_SEGMENT overrides, 0 = the engine's choice.
        interior = self.hi - self.lo - 2 * self.edge_planes
        if self._signalled:
            interior = -(-(self.hi - self.lo - 2) // 4) * 4     # the one launch sweeps all planes but the upper edge
        # The direct schedule runs its launches one after the other on one stream, and the exchange needs no launch
        # of its own besides RCCL's copy, which finds a free compute unit when the edge launch retires: one segment per
        # tile (the engine's choice) is best there -- 512 x 512 x 64, self exchange: 0.307 ms per step against 0.315 /
        # 0.318 / 0.323 with segments of 30 / 20 / 15 planes (profiles/r03_slab_direct_probe.txt).
        default = 0 if self._direct_ok() else (interior // 4 if interior >= 32 else 0)
        seg = int(os.environ.get("LT_SLAB_RCCL_SEGMENT", str(default)))
        if seg > 0 and self._window is None and hasattr(self.engine, "set_two_step"):
            self.engine.set_two_step(1, seg)

    def _message_blocks(self, stencil) -> int:
        e = np.array(stencil.e)
        return int((e[:, 2] == 0).sum()) + (3 if self._masked else 2) * len(self.up)

    def _direct_ok(self) -> bool:
        return (self._direct and self._window is None and not self._masked
                and hasattr(self.engine, "stream_collide_twice_edges_direct")
                and self.hi - self.lo >= 2 * self.edge_planes)

    def _field_ghosts(self, buf):
        """the ghost planes of ``buf`` are about to be read by a launch that takes them from the field: scatter
        the messages the direct schedule left in the receive buffers"""
        if self._ghost_src is not None:
            from_below, from_above = self._ghost_src
            self._ghost_src = None
            self.engine.unpack_two_step(buf, +1, from_above)
            self.engine.unpack_two_step(buf, -1, from_below)

    def _present(self):
        if self._pending is not None:
            self._field_ghosts(self._pending[0])
        super()._present()

    def _transfer(self, send_down, send_up):
        """the halo messages of one double step travel (no packing, no unpacking); returns (message from below,
        message from above) once they have arrived -- in stream order"""
        s = self.slab
        if self._cw is not None:
            self._cw.send(send_down, send_up)
            from_above, from_below = self._cw.wait()
            return from_below, from_above
        if s.world_size == 1 and not self._force_p2p:
            return send_up, send_down                 # my own messages: what left upwards arrives from below
        host = self._host_transport and send_down.is_cuda
        if host:
            torch.cuda.current_stream().synchronize()
        ops = [dist.P2POp(dist.isend, send_down, s.prev, self.group, tag=1),
               dist.P2POp(dist.irecv, self._recv_down, s.next, self.group, tag=1),
               dist.P2POp(dist.isend, send_up, s.next, self.group, tag=2),
               dist.P2POp(dist.irecv, self._recv_up, s.prev, self.group, tag=2)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        if host:
            torch.cuda.current_stream().synchronize()
        return self._recv_up, self._recv_down

    def _double_step_direct(self, cur, nxt, tau):
        eng, lo, hi, edge = self.engine, self.lo, self.hi, self.edge_planes
        if self._send2 is None:
            # the messages of consecutive double steps alternate between two pairs of buffers: without a transport
            # (one rank) the launch that reads the last messages writes the next ones
            self._send2 = (torch.empty_like(self._send_down), torch.empty_like(self._send_up))
        send_down, send_up = (self._send_down, self._send_up) if self._parity == 0 else self._send2
        self._parity ^= 1
        below, above = self._ghost_src if self._ghost_src is not None else (None, None)
        eng.stream_collide_twice_edges_direct(cur, nxt, tau, edge, below, above, send_down, send_up)
        if self.overlap:
            compute = torch.cuda.current_stream()
            edges_done = torch.cuda.Event()
            edges_done.record(compute)
            if hi - lo > 2 * edge:
                eng.stream_collide_twice_planes(cur, nxt, tau, lo + edge, hi - edge)
            self._comm.wait_event(edges_done)
            with torch.cuda.stream(self._comm):
                self._ghost_src = self._transfer(send_down, send_up)
            compute.wait_stream(self._comm)
        else:
            self._ghost_src = self._transfer(send_down, send_up)
            if hi - lo > 2 * edge:
                eng.stream_collide_twice_planes(cur, nxt, tau, lo + edge, hi - edge)

    # ---- halo exchange -------------------------------------------------------------------------
    def _exchange(self, buf: torch.Tensor, packed: bool = False):
        """Fill the four ghost planes of ``buf`` (post-collision populations).  My message for the
        lower neighbour comes from my two lowest interior planes and lands in its upper ghost
        planes, and vice versa."""
        eng, s = self.engine, self.slab
        self._ghost_src = None                        # this exchange ends in the ghost planes of ``buf``
        if self._cw is not None:
            if not packed:
                eng.pack_two_step(buf, -1, self._send_down)
                eng.pack_two_step(buf, +1, self._send_up)
            self._cw.send(self._send_down, self._send_up)

            def finish_copy():
                from_above, from_below = self._cw.wait()
                eng.unpack_two_step(buf, +1, from_above)
                eng.unpack_two_step(buf, -1, from_below)
            return finish_copy
        if self._window is not None:
            if not packed:
                to_prev, to_next = self._window.targets()
                eng.pack_two_step(buf, -1, to_prev)
                eng.pack_two_step(buf, +1, to_next)
            self._window.signal()

            def finish_window():
                from_above, from_below = self._window.wait()
                eng.unpack_two_step(buf, +1, from_above)
                eng.unpack_two_step(buf, -1, from_below)
            return finish_window
        if not packed:
            eng.pack_two_step(buf, -1, self._send_down)
            eng.pack_two_step(buf, +1, self._send_up)
        if s.world_size == 1 and not self._force_p2p:
            from_above, from_below, reqs = self._send_down, self._send_up, []
        else:
            from_above, from_below = self._recv_down, self._recv_up
            if self._host_transport and buf.is_cuda:
                torch.cuda.current_stream().synchronize()
            ops = [dist.P2POp(dist.isend, self._send_down, s.prev, self.group, tag=1),
                   dist.P2POp(dist.irecv, from_above, s.next, self.group, tag=1),
                   dist.P2POp(dist.isend, self._send_up, s.next, self.group, tag=2),
                   dist.P2POp(dist.irecv, from_below, s.prev, self.group, tag=2)]
            reqs = dist.batch_isend_irecv(ops)

        def finish():
            for r in reqs:
                r.wait()
            if reqs and self._host_transport and buf.is_cuda:
                torch.cuda.current_stream().synchronize()
            eng.unpack_two_step(buf, +1, from_above)
            eng.unpack_two_step(buf, -1, from_below)
        return finish

    # ---- stepping --------------------------------------------------------------------------------
    def _edges(self, cur, nxt, tau) -> bool:
        """the output planes next to the two cuts; returns True when the launches also wrote the
        halo messages (into the send buffers or straight into the neighbours' windows)"""
        eng, lo, hi, edge = self.engine, self.lo, self.hi, self.edge_planes
        # Fused packing only into local send buffers.  With peer windows the message is 20 MB of
        # stores over one xGMI link per direction (~0.3-0.4 ms): inside the edge launch they would
        # hold all CUs (one workgroup per CU, 150 KB of LDS) for that long, so a separate light pack
        # launch does them beside the interior launch instead (fused_remote_pack=True to A/B; on
        # one GPU, where the "remote" stores are local, fusing is 2 % faster; bench.py times both).
        fuse = (self._window is None or self._fused_remote) and not self._masked   # fused packing: periodic plans
        # One launch for both edges (lt_stream_collide_twice_edges) measured no better (RCCL, fused
        # windows) or worse (separate pack on two streams: 0.389 vs 0.348 ms/step -- the bigger launch
        # competes with the interior launch for CUs and delays the exchange): two launches.
        if (hasattr(eng, "stream_collide_twice_edges") and os.environ.get("LT_SLAB_MERGED_EDGES") == "1"
                and not self._masked):
            if fuse:
                down, up = ((self._send_down, self._send_up) if self._window is None
                            else self._window.targets())
                eng.stream_collide_twice_edges(cur, nxt, tau, edge, pack_lower=down, pack_upper=up)
            else:
                eng.stream_collide_twice_edges(cur, nxt, tau, edge)
            return fuse
        if fuse and hasattr(eng, "stream_collide_twice_planes_packed"):
            down, up = ((self._send_down, self._send_up) if self._window is None
                        else self._window.targets())
            eng.stream_collide_twice_planes_packed(cur, nxt, tau, lo, lo + edge, pack_lower=down)
            eng.stream_collide_twice_planes_packed(cur, nxt, tau, hi - edge, hi, pack_upper=up)
            return True
        eng.stream_collide_twice_planes(cur, nxt, tau, lo, lo + edge)
        eng.stream_collide_twice_planes(cur, nxt, tau, hi - edge, hi)
        return False

    def _signalled_ok(self) -> bool:
        return (self._signalled and self.overlap and self._window is None and not self._masked
                and hasattr(self.engine, "stream_collide_twice_slab"))

    def _double_step(self, cur, nxt, tau):
        eng, lo, hi = self.engine, self.lo, self.hi
        edge = self.edge_planes
        if self._direct_ok():
            return self._double_step_direct(cur, nxt, tau)
        if self._signalled_ok():
            # one launch; the communication stream waits for the edge workgroups' count, not for the launch
            eng.stream_collide_twice_slab(cur, nxt, tau)
            with torch.cuda.stream(self._comm):
                eng.wait_edges()
                self._exchange(nxt)()
            torch.cuda.current_stream().wait_stream(self._comm)
            return
        if self._one_stream and hi - lo >= 2 * edge + 4:
            packed = self._edges(cur, nxt, tau)
            finish = self._exchange(nxt, packed)  # (pack into the neighbours' windows) + signal
            eng.stream_collide_twice_planes(cur, nxt, tau, lo + edge, hi - edge)
            finish()                              # wait for their signals + unpack
        elif self.overlap and hi - lo >= 2 * edge + 4:
            compute = torch.cuda.current_stream()
            self._comm.wait_stream(compute)
            with torch.cuda.stream(self._comm):
                packed = self._edges(cur, nxt, tau)
                self._exchange(nxt, packed)()
            eng.stream_collide_twice_planes(cur, nxt, tau, lo + edge, hi - edge)
            compute.wait_stream(self._comm)
        else:
            eng.stream_collide_twice_planes(cur, nxt, tau, lo, hi)
            self._exchange(nxt)()

    def _advance(self, n: int):
        """n whole steps: (collide, exchange,) the n - 1 or -- carrying on from the batch before -- n fused steps
        as double steps (+ one single fused step when their number is odd); the streaming pass that completes
        the last one runs when ``f`` is read."""
        tau = float(self._tau(self.flow))
        eng, lo, hi = self.engine, self.lo, self.hi
        cur, nxt, carried = self._start_batch(tau)
        fused = n if carried else n - 1
        signalled_steps = fused >= 2
        while fused >= 2:
            self._double_step(cur, nxt, tau)
            cur, nxt = nxt, cur
            fused -= 2
        if fused == 1:
            self._field_ghosts(cur)
            eng.stream_collide_planes(cur, nxt, tau, lo, hi)
            cur, nxt = nxt, cur
            self._exchange(cur)()
        self._pending = (cur, nxt)                    # streamed when somebody looks (``f``)
        if signalled_steps and self._signalled_ok() and eng.wait_timed_out():
            # the polling wave gave up before the edge workgroups reported (it waits about a second): the
            # exchange that followed sent planes that were not written yet
            raise LettuceException("signalled slab launch: the communication stream timed out waiting for the "
                                   "edge planes; the populations of this batch are not valid")
