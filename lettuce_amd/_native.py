"""ctypes binding of the C-ABI HIP engine (include/lettuce_hip.h).

This is the host side of the drop-in boundary: it plays the role of the
reference's generated python ``invoke`` + pybind module
(lettuce/cuda_native/_template.py:31-44,58-86) but binds a prebuilt shared
library instead of JIT-compiling one per configuration.  PyTorch tensors are
only containers here: the engine sees ``tensor.data_ptr()`` and the raw
``hipStream_t`` of torch's current stream.

There is no CPU fallback in this module: if the library is missing or a
combination is unsupported, a ``LettuceException`` is raised.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import torch

from ._errors import LettuceException

__all__ = ["NativeEngineError", "load_library", "library_path", "Plan", "STENCIL_IDS",
           "COLLISION_IDS", "BOUNDARY_KINDS"]

LT_ABI_VERSION = 2
LT_MAX_BOUNDARIES = 127
LT_MAX_Q = 27

STENCIL_IDS = {"D2Q9": 0, "D3Q19": 1, "D3Q27": 2, "D1Q3": 3, "D3Q15": 4}
DTYPE_IDS = {torch.float32: 0, torch.float64: 1}
COLLISION_IDS = {"none": 0, "bgk": 1, "kbc": 2}
BOUNDARY_KINDS = {"bounce_back": 1, "equilibrium": 2, "abb_outlet": 3}
LAYOUT_REFERENCE, LAYOUT_SLAB = 0, 1


class NativeEngineError(LettuceException):
    """Raised for every failure of the HIP engine (library missing, unsupported configuration,
    HIP error).  A ``LettuceException`` (lettuce/util/utility.py:21-22), so reference-style
    ``except LettuceException`` handlers see it; the interpreter is never aborted."""


class _BoundaryDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("axis", ctypes.c_int32), ("side", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("feq", ctypes.c_double * LT_MAX_Q),
                ("feq_field_dev", ctypes.c_void_p)]


class _PlanDesc(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_int32), ("stencil", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("collision", ctypes.c_int32),
                ("layout", ctypes.c_int32), ("ghost_planes", ctypes.c_int32),
                ("dims", ctypes.c_int32), ("n_boundaries", ctypes.c_int32),
                ("shape", ctypes.c_int64 * 3),
                ("boundaries", _BoundaryDesc * LT_MAX_BOUNDARIES)]


# every symbol include/lettuce_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
SYMBOLS = {
    "lt_abi_version": (ctypes.c_int, []),
    "lt_build_flags": (ctypes.c_int, []),
    "lt_last_error": (ctypes.c_char_p, []),
    "lt_plan_create": (ctypes.c_int, [ctypes.POINTER(_PlanDesc), ctypes.POINTER(_vp)]),
    "lt_plan_destroy": (ctypes.c_int, [_vp]),
    "lt_plan_set_masks": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lt_plan_update_boundary": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(_BoundaryDesc), _vp]),
    "lt_collide": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _vp]),
    "lt_stream": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lt_stream_collide": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _vp]),
    "lt_collide_planes": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _i64, _i64, _vp]),
    "lt_stream_planes": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "lt_stream_collide_planes": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _i64, _i64, _vp]),
    "lt_stream_collide_plane_pair": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _i64, _i64, _vp]),
    "lt_stream_collide_plane_pair_packed": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _i64, _i64, _vp, _vp, _vp]),
    "lt_slab_crossing": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "lt_slab_pack": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "lt_slab_unpack": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "lt_run": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _i64, _vp, ctypes.POINTER(_i32)]),
    "lt_continue": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _i64, _vp, ctypes.POINTER(_i32)]),
    "lt_macroscopic": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "lt_equilibrium": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "lt_kinetic_energy": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lt_mass": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lt_max_velocity": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lt_init_fneq": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _dbl, _vp, _vp]),
    "lt_enstrophy": (ctypes.c_int, [_vp, _vp, _vp, _dbl, _dbl, _vp, _vp]),
    "lt_mass_interior": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "lt_slab_velocity": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lt_slab_enstrophy": (ctypes.c_int, [_vp, _vp, _dbl, _dbl, _vp, _vp]),
    "lt_slab_mass_interior": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "lt_plan_kernel_info": (ctypes.c_int, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_i32),
                                           ctypes.POINTER(_i64)]),
    "lt_plan_kernel_name": (ctypes.c_char_p, [_vp]),
    "lt_plan_set_shift_policy": (ctypes.c_int, [_vp, _i32]),
    "lt_plan_set_graph_mode": (ctypes.c_int, [_vp, _i32]),
    "lt_plan_set_tuning": (ctypes.c_int, [_vp, _i32, _i32]),
    "lt_plan_set_residency": (ctypes.c_int, [_vp, _i32]),
    "lt_stream_collide_twice": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _vp]),
    "lt_plan_set_two_step": (ctypes.c_int, [_vp, _i32, _i32]),
    "lt_stream_collide_twice_planes": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _i64, _i64, _vp]),
    "lt_stream_collide_twice_planes_packed": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _i64, _i64, _vp, _vp, _vp]),
    "lt_stream_collide_twice_edges": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _i32, _vp, _vp, _vp]),
    "lt_stream_collide_twice_edges_direct": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _i32, _vp, _vp, _vp, _vp, _vp]),
    "lt_plan_set_deferred_stream": (ctypes.c_int, [_vp, _i32]),
    "lt_stream_collide_twice_slab": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _vp]),
    "lt_slab_wait_edges": (ctypes.c_int, [_vp, _vp]),
    "lt_slab_wait_timed_out": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), _vp]),
    "lt_slab_two_step_message_blocks": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    "lt_plan_two_step_admitted": (ctypes.c_int, [_vp]),
    "lt_two_step_limits": (ctypes.c_int, [ctypes.POINTER(_PlanDesc), _i32, ctypes.POINTER(_i32), ctypes.POINTER(_i32),
                                          ctypes.POINTER(_i32)]),
    "lt_slab_pack_two_step": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "lt_slab_unpack_two_step": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "lt_plan_set_population_stride": (ctypes.c_int, [_vp, _i64]),
    "lt_plan_population_stride": (ctypes.c_int, [_vp, ctypes.POINTER(_i64)]),
    "lt_plan_set_resident": (ctypes.c_int, [_vp, _i32, _i64]),
    "lt_resident_enabled": (ctypes.c_int, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_i64)]),
    "lt_resident_load": (ctypes.c_int, [_vp, _vp, _dbl, _vp]),
    "lt_resident_advance": (ctypes.c_int, [_vp, _dbl, _i64, _vp]),
    "lt_resident_store": (ctypes.c_int, [_vp, _vp, _vp]),
    "lt_resident_free": (ctypes.c_int, [_vp]),
    "lt_plan_set_fused_events": (ctypes.c_int, [_vp, _vp, _vp]),
    "lt_plan_last_run_info": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                             ctypes.POINTER(ctypes.c_int64)]),
    "lt_stream_collide_many": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _i32, _vp]),
    "lt_plan_set_many_step": (ctypes.c_int, [_vp, _i32]),
    "lt_probe_copy": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "lt_probe_div_cs": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "lt_ipc_alloc": (ctypes.c_int, [_i64, _i32, ctypes.POINTER(_vp), ctypes.c_char_p]),
    "lt_ipc_open": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_vp)]),
    "lt_ipc_close": (ctypes.c_int, [_vp]),
    "lt_ipc_free": (ctypes.c_int, [_vp]),
    "lt_halo_copy": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, ctypes.POINTER(_i32)]),
    "lt_flag_write": (ctypes.c_int, [_vp, ctypes.c_uint64, _i32, _vp]),
    "lt_flag_wait": (ctypes.c_int, [_vp, ctypes.c_uint64, _vp, _vp]),
    "lt_plan_set_canary": (ctypes.c_int, [_vp, _i32]),
    "lt_plan_canary_status": (ctypes.c_int, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_char_p)]),
}

# entry points of the experiments build (make -C lettuce_amd/csrc EXPERIMENTS=1): bound when the library has them
EXPERIMENT_SYMBOLS = {
    "lt_stream_collide_thrice": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _vp]),
    "lt_plan_set_arithmetic": (ctypes.c_int, [_vp, _i32]),
}

_LIB = None


def library_path() -> str:
    """the in-tree library; LT_ENGINE_LIBRARY names another build of it (A/B experiments with compiler flags)"""
    override = os.environ.get("LT_ENGINE_LIBRARY")
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblettuce_hip.so")


def load_library() -> ctypes.CDLL:
    """Load liblettuce_hip.so (built in-tree by ``__graft_entry__.build()`` /
    ``make -C lettuce_amd/csrc``) and bind every declared symbol."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise NativeEngineError(
            f"HIP engine library not found at {path}; build it with "
            f"`make -C lettuce_amd/csrc -j8` or `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:
        raise NativeEngineError(f"cannot load {path}: {exc}") from exc
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise NativeEngineError(f"{path} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.lt_build_flags() & 1:
        for name, (restype, argtypes) in EXPERIMENT_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
    if lib.lt_abi_version() != LT_ABI_VERSION:
        raise NativeEngineError(f"ABI mismatch: library {lib.lt_abi_version()}, "
                                f"binding {LT_ABI_VERSION}")
    _LIB = lib
    return lib


def experiments_built() -> bool:
    """was the library built with the kernels that lost their A/B (make EXPERIMENTS=1)?"""
    return bool(load_library().lt_build_flags() & 1)


def _stream_handle() -> int:
    return torch.cuda.current_stream().cuda_stream


def _on_device(method):
    """Run a Plan method with the plan's GPU as the current device (kernel launches, the current
    stream and allocations then all belong to it, whatever device the caller had selected)."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        if torch.cuda.current_device() == self.device.index:
            return method(self, *args, **kwargs)
        with torch.cuda.device(self.device):
            return method(self, *args, **kwargs)
    return wrapper


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def probe_copy(dst: torch.Tensor, src: torch.Tensor, cache_policy: int = 0, max_blocks: int = 0):
    """dst <- src with the engine's plain 16-byte streaming copy kernel (copy-ceiling probe)."""
    lib = load_library()
    nbytes = src.numel() * src.element_size()
    code = lib.lt_probe_copy(_ptr(dst), _ptr(src), nbytes, int(cache_policy), int(max_blocks),
                             _stream_handle())
    if code != 0:
        raise NativeEngineError(lib.lt_last_error().decode())


def probe_div_cs(x: torch.Tensor, which: int) -> torch.Tensor:
    """x / (2 cs^2) (which 0) or x / cs^2 (which 1) with the kernels' exact-division emulation (test hook)"""
    lib = load_library()
    out = torch.empty_like(x)
    code = lib.lt_probe_div_cs(_ptr(x), _ptr(out), x.numel(), DTYPE_IDS[x.dtype], int(which), _stream_handle())
    if code != 0:
        raise NativeEngineError(lib.lt_last_error().decode())
    return out


class Plan:
    """One engine configuration: lattice x dtype x collision x grid (x boundaries).

    ``boundaries`` is a list of dicts in index order (index = position + 1 in
    ``no_collision_mask``): ``{"kind": "bounce_back"}``,
    ``{"kind": "equilibrium", "feq": [q floats] | "field": tensor[q,*res]}``,
    ``{"kind": "abb_outlet", "axis": a, "side": +-1}``.
    """

    def __init__(self, stencil: str, dtype: torch.dtype, collision: str,
                 resolution: Sequence[int], boundaries: Sequence[dict] = (),
                 layout: int = LAYOUT_REFERENCE, ghost_planes: int = 0,
                 device: Optional[torch.device] = None):
        self._handle = None
        self.lib = load_library()
        if stencil not in STENCIL_IDS:
            raise NativeEngineError(f"stencil {stencil} has no HIP kernels "
                                    f"(available: {sorted(STENCIL_IDS)})")
        if dtype not in DTYPE_IDS:
            raise NativeEngineError(f"dtype {dtype} has no HIP kernels (float32/float64 only)")
        if collision not in COLLISION_IDS:
            raise NativeEngineError(f"collision {collision!r} has no HIP kernels")
        if len(boundaries) > LT_MAX_BOUNDARIES:
            raise NativeEngineError(f"{len(boundaries)} boundaries; the engine takes "
                                    f"{LT_MAX_BOUNDARIES}")
        self.stencil, self.dtype, self.collision = stencil, dtype, collision
        self.resolution = [int(r) for r in resolution]
        self.q = int(stencil.split("Q")[1])
        self.d = len(self.resolution)
        self.layout, self.ghost_planes = layout, ghost_planes
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._keepalive = {}            # boundary index -> field tensor the engine holds a pointer to
        self._const = {}                # answers of the engine that do not change while the masks stay
        self.resident_mode = -1         # lt_plan_set_resident: -1 automatic, 0 off, 1 requested
        self.pop_stride = 0             # elements between populations of the caller's buffers (0 = dense)
        desc = _PlanDesc()
        desc.abi_version = LT_ABI_VERSION
        desc.stencil = STENCIL_IDS[stencil]
        desc.dtype = DTYPE_IDS[dtype]
        desc.collision = COLLISION_IDS[collision]
        desc.layout = layout
        desc.ghost_planes = ghost_planes
        desc.dims = self.d
        desc.n_boundaries = len(boundaries)
        for a in range(3):
            desc.shape[a] = self.resolution[a] if a < self.d else 1
        for i, b in enumerate(boundaries):
            self._fill_boundary(desc.boundaries[i], b, i)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.lt_plan_create(ctypes.byref(desc), ctypes.byref(handle)))
        self._handle = handle
        self.n_boundaries = len(boundaries)

    # ------------------------------------------------------------------ helpers
    @_on_device
    def _fill_boundary(self, out: _BoundaryDesc, b: dict, index: int):
        out.kind = BOUNDARY_KINDS[b["kind"]]
        out.axis = int(b.get("axis", 0))
        out.side = int(b.get("side", 0))
        out.flags = 0 if b.get("present", True) else 1      # LT_BOUNDARY_ABSENT: another rank holds the outlet plane
        out.feq_field_dev = None
        if b["kind"] == "equilibrium":
            field = b.get("field")
            if field is not None:
                field = field.to(device=self.device, dtype=self.dtype).contiguous()
                old = self._keepalive.get(index)
                if old is not None and old is not field and old.is_cuda:
                    old.record_stream(torch.cuda.current_stream())   # launches in flight may still read it
                self._keepalive[index] = field          # a superseded field is released
                out.feq_field_dev = field.data_ptr()
            else:
                self._keepalive.pop(index, None)
                for q, v in enumerate(b["feq"]):
                    out.feq[q] = float(v)

    def _check(self, code: int):
        if code != 0:
            msg = self.lib.lt_last_error().decode("utf-8", "replace")
            exc = NativeEngineError(f"HIP engine error {code}: {msg}")
            exc.code = code                                  # LT_ERR_* (4 = LT_ERR_ALLOC)
            raise exc

    def _tensor_ok(self, t: torch.Tensor, shape=None):
        if t.device.type != "cuda":
            raise NativeEngineError(f"tensor on {t.device}; the HIP engine needs device memory")
        if t.dtype != self.dtype:
            raise NativeEngineError(f"tensor dtype {t.dtype}, plan dtype {self.dtype}")
        stride = getattr(self, "pop_stride", 0)
        populations = shape is not None and list(shape) == list(self.f_shape)
        if stride and populations:
            # the engine addresses population q at q * stride: a dense clone() / empty_like() of a padded tensor
            # would be written up to (q - 1) * pad elements past its end
            if not (t.dim() > 1 and t.stride(0) == stride and t[0].is_contiguous()):
                raise NativeEngineError(f"this plan's population buffers have {stride} elements between populations "
                                        f"(set_population_stride); got strides {tuple(t.stride())} -- allocate with "
                                        f"empty_populations() / populations_like()")
        elif not t.is_contiguous():
            raise NativeEngineError("population tensors must be contiguous (or carry the plan's population stride)")
        if shape is not None and list(t.shape) != list(shape):
            raise NativeEngineError(f"tensor shape {list(t.shape)}, expected {list(shape)}")

    def _populations_ok(self, *tensors):
        for t in tensors:
            self._tensor_ok(t, self.f_shape)

    def _message_ok(self, buf: Optional[torch.Tensor], blocks: int):
        """a halo message buffer: `blocks` planes of n1 * n0 values, contiguous"""
        if buf is None:
            return
        need = blocks * self.f_shape[-1] * self.f_shape[-2]
        if buf.device.type != "cuda" or buf.dtype != self.dtype or not buf.is_contiguous() or buf.numel() < need:
            raise NativeEngineError(f"halo message buffer: need a contiguous {self.dtype} device tensor of at least "
                                    f"{need} elements, got {tuple(buf.shape)} {buf.dtype} on {buf.device}")

    @property
    def f_shape(self):
        """Shape of a population tensor in this plan's memory layout."""
        if self.layout == LAYOUT_REFERENCE:
            return [self.q] + self.resolution
        nx, ny, nz = self.resolution
        return [self.q, nz + 2 * self.ghost_planes, ny, nx]

    def close(self):
        if getattr(self, "_handle", None) is not None:
            self.lib.lt_plan_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ masks
    @_on_device
    def set_masks(self, no_collision_mask: Optional[torch.Tensor],
                  no_streaming_mask: Optional[torch.Tensor]):
        """uint8 ``[*res]`` / uint8 ``[q, *res]`` device tensors (either may be None)."""
        ncm = nsm = None
        if no_collision_mask is not None:
            ncm = no_collision_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if list(ncm.shape) != self.f_shape[1:]:
                raise NativeEngineError(f"no_collision_mask shape {list(ncm.shape)}, expected "
                                        f"{self.f_shape[1:]}")
        if no_streaming_mask is not None:
            nsm = no_streaming_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if list(nsm.shape) != self.f_shape:
                raise NativeEngineError(f"no_streaming_mask shape {list(nsm.shape)}, expected "
                                        f"{self.f_shape}")
        self._check(self.lib.lt_plan_set_masks(self._handle, _ptr(ncm), _ptr(nsm), _stream_handle()))
        self._const.pop("blocks", None)
        # the compile kernel reads them asynchronously on the current stream
        if ncm is not None:
            ncm.record_stream(torch.cuda.current_stream())
        if nsm is not None:
            nsm.record_stream(torch.cuda.current_stream())

    @_on_device
    def update_boundary(self, index: int, b: dict):
        d = _BoundaryDesc()
        self._fill_boundary(d, b, index)
        self._check(self.lib.lt_plan_update_boundary(self._handle, index, ctypes.byref(d),
                                                     _stream_handle()))

    # ------------------------------------------------------------------ operators
    @_on_device
    def collide(self, f, out, tau):
        self._tensor_ok(f, self.f_shape); self._tensor_ok(out, self.f_shape)
        self._check(self.lib.lt_collide(self._handle, _ptr(f), _ptr(out), float(tau), _stream_handle()))
        return out

    @_on_device
    def stream(self, f, out):
        self._tensor_ok(f, self.f_shape); self._tensor_ok(out, self.f_shape)
        self._check(self.lib.lt_stream(self._handle, _ptr(f), _ptr(out), _stream_handle()))
        return out

    @_on_device
    def stream_collide(self, f, out, tau):
        self._tensor_ok(f, self.f_shape); self._tensor_ok(out, self.f_shape)
        self._check(self.lib.lt_stream_collide(self._handle, _ptr(f), _ptr(out), float(tau),
                                               _stream_handle()))
        return out

    @_on_device
    def collide_planes(self, f, out, tau, begin, end):
        self._populations_ok(f, out)
        self._check(self.lib.lt_collide_planes(self._handle, _ptr(f), _ptr(out), float(tau),
                                               int(begin), int(end), _stream_handle()))

    @_on_device
    def stream_planes(self, f, out, begin, end):
        self._populations_ok(f, out)
        self._check(self.lib.lt_stream_planes(self._handle, _ptr(f), _ptr(out), int(begin),
                                              int(end), _stream_handle()))

    @_on_device
    def stream_collide_planes(self, f, out, tau, begin, end):
        self._populations_ok(f, out)
        self._check(self.lib.lt_stream_collide_planes(self._handle, _ptr(f), _ptr(out), float(tau),
                                                      int(begin), int(end), _stream_handle()))

    def crossing(self, direction: int):
        """population indices whose velocity along the slowest memory axis is ``direction``"""
        key = ("crossing", int(direction))
        if key not in self._const:
            qs, n = (ctypes.c_int32 * 9)(), ctypes.c_int32()
            self._check(self.lib.lt_slab_crossing(self._handle, int(direction), qs, ctypes.byref(n)))
            self._const[key] = [int(qs[k]) for k in range(n.value)]
        return list(self._const[key])

    @_on_device
    def pack(self, f, plane, direction, buf):
        self._populations_ok(f); self._message_ok(buf, len(self.crossing(direction)))
        self._check(self.lib.lt_slab_pack(self._handle, _ptr(f), int(plane), int(direction), _ptr(buf),
                                          _stream_handle()))

    @_on_device
    def unpack(self, f, plane, direction, buf):
        self._populations_ok(f); self._message_ok(buf, len(self.crossing(direction)))
        self._check(self.lib.lt_slab_unpack(self._handle, _ptr(f), int(plane), int(direction), _ptr(buf),
                                            _stream_handle()))

    @_on_device
    def stream_collide_plane_pair(self, f, out, tau, first, second):
        self._populations_ok(f, out)
        self._check(self.lib.lt_stream_collide_plane_pair(self._handle, _ptr(f), _ptr(out), float(tau),
                                                          int(first), int(second), _stream_handle()))

    @_on_device
    def stream_collide_plane_pair_packed(self, f, out, tau, first, second, pack_first, pack_second):
        self._populations_ok(f, out)
        for buf in (pack_first, pack_second):
            self._message_ok(buf, len(self.crossing(1)))
        self._check(self.lib.lt_stream_collide_plane_pair_packed(
            self._handle, _ptr(f), _ptr(out), float(tau), int(first), int(second), _ptr(pack_first),
            _ptr(pack_second), _stream_handle()))

    @_on_device
    def set_deferred_stream(self, on: bool):
        """run() then stops before its last streaming pass and returns (f*, scratch): the post-collision
        populations of the last step and a buffer whose content is undefined (``stream(f*, scratch)``
        gives the post-streaming populations)"""
        self._check(self.lib.lt_plan_set_deferred_stream(self._handle, int(bool(on))))

    @_on_device
    def run(self, a, b, tau, n_steps, from_fstar=False):
        """n whole steps; returns (result, other): ``result`` holds the new post-streaming
        populations, ``other`` the post-collision populations of the last step (with
        ``set_deferred_stream(True)``: see there)."""
        self._tensor_ok(a, self.f_shape); self._tensor_ok(b, self.f_shape)
        which = ctypes.c_int32(0)
        fn = self.lib.lt_continue if from_fstar else self.lib.lt_run
        self._check(fn(self._handle, _ptr(a), _ptr(b), float(tau), int(n_steps), _stream_handle(),
                       ctypes.byref(which)))
        return (b, a) if which.value else (a, b)

    # ------------------------------------------------------------------ population stride / resident state
    def set_population_stride(self, stride: int = 0):
        """every population buffer given to this plan has ``stride`` elements between consecutive populations
        (0 = dense); see ``empty_populations``"""
        self._check(self.lib.lt_plan_set_population_stride(self._handle, int(stride)))
        self.pop_stride = int(stride)

    def empty_populations(self) -> torch.Tensor:
        """an uninitialised population tensor of this plan's shape and population stride (a strided view of one
        allocation when the plan is padded)"""
        stride = getattr(self, "pop_stride", 0)
        shape = self.f_shape
        with torch.cuda.device(self.device):
            if not stride:
                return torch.empty(shape, dtype=self.dtype, device=self.device)
            flat = torch.empty(self.q * stride, dtype=self.dtype, device=self.device)
        inner = torch.empty(shape[1:], device="meta").stride()
        return flat.as_strided(shape, (stride,) + tuple(inner))

    def populations_like(self, f: torch.Tensor) -> torch.Tensor:
        """``f`` copied into a tensor with this plan's population stride"""
        out = self.empty_populations()
        out.copy_(f)
        return out

    def set_resident(self, mode: int = -1, pad_elements: int = -1):
        """engine-owned padded buffers for the fused steps: -1 automatic, 0 off, 1 on"""
        self._check(self.lib.lt_plan_set_resident(self._handle, int(mode), int(pad_elements)))
        self.resident_mode = int(mode)

    @_on_device
    def resident_enabled(self):
        """(enabled, stride in elements of the resident buffers)"""
        on, stride = ctypes.c_int32(0), ctypes.c_int64(0)
        self._check(self.lib.lt_resident_enabled(self._handle, ctypes.byref(on), ctypes.byref(stride)))
        return bool(on.value), int(stride.value)

    @_on_device
    def resident_load(self, f, tau):
        """collide pass from the caller's dense populations into the engine's padded buffers, which are allocated
        here on first use (raw hipMalloc: two more population fields).  If that fails, memory torch merely caches is
        returned to the driver and the allocation tried once more."""
        self._tensor_ok(f, self.f_shape)
        code = self.lib.lt_resident_load(self._handle, _ptr(f), float(tau), _stream_handle())
        if code == 4:                                        # LT_ERR_ALLOC
            torch.cuda.empty_cache()
            code = self.lib.lt_resident_load(self._handle, _ptr(f), float(tau), _stream_handle())
        self._check(code)

    @_on_device
    def resident_advance(self, tau, n_steps):
        self._check(self.lib.lt_resident_advance(self._handle, float(tau), int(n_steps), _stream_handle()))

    @_on_device
    def resident_store(self, out):
        self._tensor_ok(out, self.f_shape)
        self._check(self.lib.lt_resident_store(self._handle, _ptr(out), _stream_handle()))
        return out

    def resident_free(self):
        self._check(self.lib.lt_resident_free(self._handle))

    @_on_device
    def macroscopic(self, f, want_rho=True, want_u=True):
        self._tensor_ok(f, self.f_shape)
        grid = self.f_shape[1:]
        rho = torch.empty(grid, dtype=self.dtype, device=f.device) if want_rho else None
        u = torch.empty([self.d] + grid, dtype=self.dtype, device=f.device) if want_u else None
        self._check(self.lib.lt_macroscopic(self._handle, _ptr(f), _ptr(rho), _ptr(u), _stream_handle()))
        return rho, u

    @_on_device
    def equilibrium(self, rho, u):
        grid = self.f_shape[1:]
        rho = rho.reshape(grid).contiguous()
        self._tensor_ok(rho, grid); self._tensor_ok(u, [self.d] + grid)
        feq = torch.empty(self.f_shape, dtype=self.dtype, device=rho.device)
        self._check(self.lib.lt_equilibrium(self._handle, _ptr(rho), _ptr(u), _ptr(feq), _stream_handle()))
        return feq

    @_on_device
    def kinetic_energy_lu(self, f):
        """0-d float64 device tensor: sum over nodes of 0.5 u.u (lattice units)."""
        self._tensor_ok(f, self.f_shape)
        out = torch.empty((), dtype=torch.float64, device=f.device)
        self._check(self.lib.lt_kinetic_energy(self._handle, _ptr(f), _ptr(out), _stream_handle()))
        return out

    @_on_device
    def mass(self, f):
        self._tensor_ok(f, self.f_shape)
        out = torch.empty((), dtype=torch.float64, device=f.device)
        self._check(self.lib.lt_mass(self._handle, _ptr(f), _ptr(out), _stream_handle()))
        return out

    @_on_device
    def max_velocity_lu(self, f):
        """0-d float64 device tensor: max over nodes of |u| (lattice units)"""
        self._tensor_ok(f, self.f_shape)
        out = torch.empty((), dtype=torch.float64, device=f.device)
        self._check(self.lib.lt_max_velocity(self._handle, _ptr(f), _ptr(out), _stream_handle()))
        return out

    @_on_device
    def init_fneq(self, rho, u, tau: float, identity_cs2: float):
        """feq(rho, u) - f_neq(grad u) in one launch (initialize_f_neq); rho [*res] or [1, *res], u [d, *res]"""
        grid = self.f_shape[1:]
        rho = rho.reshape(grid).contiguous()
        u = u.contiguous()
        self._tensor_ok(rho, grid); self._tensor_ok(u, [self.d] + grid)
        f = torch.empty(self.f_shape, dtype=self.dtype, device=rho.device)
        self._check(self.lib.lt_init_fneq(self._handle, _ptr(rho), _ptr(u), float(tau), float(identity_cs2), _ptr(f),
                                          _stream_handle()))
        for t in (rho, u):
            t.record_stream(torch.cuda.current_stream())
        return f

    @_on_device
    def enstrophy_sum(self, f, u_scale: float, inv_dx: float):
        """0-d float64 device tensor: sum over nodes of |curl(u_scale * u)|^2 (6th-order periodic
        differences times inv_dx); one [d, *res] scratch field for u, nothing else is materialised."""
        self._tensor_ok(f, self.f_shape)
        scratch = torch.empty([self.d] + self.f_shape[1:], dtype=self.dtype, device=f.device)
        out = torch.empty((), dtype=torch.float64, device=f.device)
        self._check(self.lib.lt_enstrophy(self._handle, _ptr(f), _ptr(scratch), float(u_scale), float(inv_dx),
                                          _ptr(out), _stream_handle()))
        scratch.record_stream(torch.cuda.current_stream())
        return out

    @_on_device
    def mass_interior(self, f, no_mass_mask: Optional[torch.Tensor] = None):
        """0-d float64 device tensor: the reference's Mass observable (borders of the two fastest axes
        excluded, the nodes of ``no_mass_mask`` subtracted)."""
        self._tensor_ok(f, self.f_shape)
        mask = None
        if no_mass_mask is not None:
            mask = torch.broadcast_to(no_mass_mask, self.f_shape[1:]).to(device=f.device, dtype=torch.uint8).contiguous()
        out = torch.empty((), dtype=torch.float64, device=f.device)
        self._check(self.lib.lt_mass_interior(self._handle, _ptr(f), _ptr(mask), _ptr(out), _stream_handle()))
        if mask is not None:
            mask.record_stream(torch.cuda.current_stream())
        return out

    # ---- a rank's share of the observables (slab plans; the slab driver exchanges / all-reduces) ----------------
    @_on_device
    def slab_velocity(self, f):
        """[3, nz + 6, ny, nx] velocity field (lattice units) of this rank's planes with room for three planes of
        either neighbour: planes [3 - g, nz + 3 + g) are written here, the outer ones by the caller."""
        self._tensor_ok(f, self.f_shape)
        nx, ny, nz = self.resolution
        u = torch.empty([3, nz + 6, ny, nx], dtype=self.dtype, device=f.device)
        self._check(self.lib.lt_slab_velocity(self._handle, _ptr(f), _ptr(u), _stream_handle()))
        return u

    @_on_device
    def slab_enstrophy_sum(self, u_ext, u_scale: float, inv_dx: float):
        """0-d float64 device tensor: sum over this rank's own nodes of |curl(u_scale * u)|^2"""
        nx, ny, nz = self.resolution
        self._tensor_ok(u_ext, [3, nz + 6, ny, nx])
        out = torch.empty((), dtype=torch.float64, device=u_ext.device)
        self._check(self.lib.lt_slab_enstrophy(self._handle, _ptr(u_ext), float(u_scale), float(inv_dx), _ptr(out),
                                               _stream_handle()))
        u_ext.record_stream(torch.cuda.current_stream())
        return out

    @_on_device
    def slab_mass_interior(self, f, z_begin: int, nz_global: int, no_mass_mask: Optional[torch.Tensor] = None):
        """0-d float64 device tensor: this rank's share of the Mass observable; ``no_mass_mask`` [nz + 2 g, ny, nx]"""
        self._tensor_ok(f, self.f_shape)
        mask = None
        if no_mass_mask is not None:
            if list(no_mass_mask.shape) != self.f_shape[1:]:
                raise NativeEngineError(f"no_mass_mask shape {list(no_mass_mask.shape)}, expected {self.f_shape[1:]}")
            mask = no_mass_mask.to(device=f.device, dtype=torch.uint8).contiguous()
        out = torch.empty((), dtype=torch.float64, device=f.device)
        self._check(self.lib.lt_slab_mass_interior(self._handle, _ptr(f), _ptr(mask), int(z_begin), int(nz_global),
                                                   _ptr(out), _stream_handle()))
        if mask is not None:
            mask.record_stream(torch.cuda.current_stream())
        return out

    # ------------------------------------------------------------------ introspection
    def kernel_info(self):
        vec, tpb, blocks = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        self._check(self.lib.lt_plan_kernel_info(self._handle, ctypes.byref(vec), ctypes.byref(tpb),
                                                 ctypes.byref(blocks)))
        return {"vec": vec.value, "threads_per_block": tpb.value, "blocks": blocks.value}

    @_on_device
    def kernel_name(self) -> str:
        return self.lib.lt_plan_kernel_name(self._handle).decode()

    def set_shift_policy(self, policy: int):
        self._check(self.lib.lt_plan_set_shift_policy(self._handle, int(policy)))

    def set_graph_mode(self, mode: int):
        """-1 automatic (small grids), 0 never, 1 always replay the fused launches as a hipGraph"""
        self._check(self.lib.lt_plan_set_graph_mode(self._handle, int(mode)))

    def set_tuning(self, cache_policy: int = -1, wide: bool = False):
        self._check(self.lib.lt_plan_set_tuning(self._handle, int(cache_policy), int(bool(wide))))

    @_on_device
    def set_fused_events(self, start=None, stop=None):
        """torch.cuda.Event pair recorded by lt_run around its fused launches (None, None: off)"""
        if start is None:
            self._check(self.lib.lt_plan_set_fused_events(self._handle, None, None))
            return
        for e in (start, stop):
            e.record()                      # torch creates the hipEvent_t lazily
        self._check(self.lib.lt_plan_set_fused_events(self._handle, ctypes.c_void_p(start.cuda_event),
                                                      ctypes.c_void_p(stop.cuda_event)))

    def last_run_info(self):
        single, twice, many = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.lt_plan_last_run_info(self._handle, ctypes.byref(single), ctypes.byref(twice),
                                                   ctypes.byref(many)))
        return {"single_step_launches": single.value, "two_step_launches": twice.value,
                "many_step_launches": many.value}

    def set_many_step(self, mode: int = -1):
        """lt_run on small 2-D grids: several steps per launch (-1 automatic, 0 never, 1 when supported)"""
        self._check(self.lib.lt_plan_set_many_step(self._handle, int(mode)))

    @_on_device
    def stream_collide_many(self, f, out, tau, n_steps):
        """out = (collide o stream)^n_steps f in one launch (small 2-D grids, n_steps <= 8)"""
        self._tensor_ok(f, self.f_shape); self._tensor_ok(out, self.f_shape)
        self._check(self.lib.lt_stream_collide_many(self._handle, _ptr(f), _ptr(out), float(tau), int(n_steps),
                                                    _stream_handle()))
        return out

    def set_two_step(self, mode: int = -1, planes_per_workgroup: int = 0):
        """lt_run pairs fused steps into two-step launches: -1 automatic, 0 never, 1 when supported"""
        self._check(self.lib.lt_plan_set_two_step(self._handle, int(mode), int(planes_per_workgroup)))

    @_on_device
    def stream_collide_thrice(self, f, out, tau):
        """out = (collide o stream)^3 f in one launch (both intermediate states in LDS; lbm3_kernel)"""
        self._tensor_ok(f, self.f_shape); self._tensor_ok(out, self.f_shape)
        if not experiments_built():
            raise NativeEngineError("three steps per launch: a kernel of the experiments build "
                                    "(make -C lettuce_amd/csrc EXPERIMENTS=1)")
        self._check(self.lib.lt_stream_collide_thrice(self._handle, _ptr(f), _ptr(out), float(tau),
                                                      _stream_handle()))
        return out

    @_on_device
    def stream_collide_twice(self, f, out, tau):
        """out = (collide o stream)^2 f in one launch (LDS-staged intermediate state)"""
        self._tensor_ok(f, self.f_shape); self._tensor_ok(out, self.f_shape)
        self._check(self.lib.lt_stream_collide_twice(self._handle, _ptr(f), _ptr(out), float(tau),
                                                     _stream_handle()))
        return out

    @_on_device
    def stream_collide_twice_planes(self, f, out, tau, begin, end):
        self._populations_ok(f, out)
        self._check(self.lib.lt_stream_collide_twice_planes(self._handle, _ptr(f), _ptr(out), float(tau),
                                                            int(begin), int(end), _stream_handle()))

    @_on_device
    def stream_collide_twice_planes_packed(self, f, out, tau, begin, end, pack_lower=None, pack_upper=None):
        self._populations_ok(f, out)
        for buf in (pack_lower, pack_upper):
            self._message_ok(buf, self.two_step_message_blocks())
        self._check(self.lib.lt_stream_collide_twice_planes_packed(
            self._handle, _ptr(f), _ptr(out), float(tau), int(begin), int(end), _ptr(pack_lower),
            _ptr(pack_upper), _stream_handle()))

    @_on_device
    def stream_collide_twice_edges(self, f, out, tau, edge_planes, pack_lower=None, pack_upper=None):
        self._populations_ok(f, out)
        for buf in (pack_lower, pack_upper):
            self._message_ok(buf, self.two_step_message_blocks())
        self._check(self.lib.lt_stream_collide_twice_edges(
            self._handle, _ptr(f), _ptr(out), float(tau), int(edge_planes), _ptr(pack_lower), _ptr(pack_upper),
            _stream_handle()))

    @_on_device
    def stream_collide_twice_edges_direct(self, f, out, tau, edge_planes, recv_lower, recv_upper, pack_lower, pack_upper):
        """the two edges of the slab in one launch that reads the planes beyond the cuts from the received halo
        messages (no unpack) and writes both outgoing messages (no pack)"""
        self._populations_ok(f, out)
        for buf in (recv_lower, recv_upper, pack_lower, pack_upper):
            self._message_ok(buf, self.two_step_message_blocks())
        self._check(self.lib.lt_stream_collide_twice_edges_direct(
            self._handle, _ptr(f), _ptr(out), float(tau), int(edge_planes), _ptr(recv_lower), _ptr(recv_upper),
            _ptr(pack_lower), _ptr(pack_upper), _stream_handle()))

    @_on_device
    def stream_collide_twice_slab(self, f, out, tau):
        """all interior planes in one launch whose edge workgroups run first and count themselves done on a
        device counter (``wait_edges`` on another stream waits for it)"""
        self._populations_ok(f, out)
        self._check(self.lib.lt_stream_collide_twice_slab(self._handle, _ptr(f), _ptr(out), float(tau), _stream_handle()))

    @_on_device
    def wait_edges(self):
        self._check(self.lib.lt_slab_wait_edges(self._handle, _stream_handle()))

    @_on_device
    def wait_timed_out(self) -> bool:
        flag = ctypes.c_int32(0)
        self._check(self.lib.lt_slab_wait_timed_out(self._handle, ctypes.byref(flag), _stream_handle()))
        return bool(flag.value)

    def two_step_message_blocks(self) -> int:
        if "blocks" not in self._const:            # depends on whether the plan has masks: set_masks forgets it
            n = ctypes.c_int32(0)
            self._check(self.lib.lt_slab_two_step_message_blocks(self._handle, ctypes.byref(n)))
            self._const["blocks"] = int(n.value)
        return self._const["blocks"]

    @_on_device
    def two_step_admitted(self) -> Optional[str]:
        """None when the plan has a two-step launch; else the engine's reason"""
        if self.lib.lt_plan_two_step_admitted(self._handle) == 0:
            return None
        return self.lib.lt_last_error().decode()

    @_on_device
    def pack_two_step(self, f, side, buf):
        self._populations_ok(f); self._message_ok(buf, self.two_step_message_blocks())
        self._check(self.lib.lt_slab_pack_two_step(self._handle, _ptr(f), int(side), _ptr(buf), _stream_handle()))

    @_on_device
    def unpack_two_step(self, f, side, buf):
        self._populations_ok(f); self._message_ok(buf, self.two_step_message_blocks())
        self._check(self.lib.lt_slab_unpack_two_step(self._handle, _ptr(f), int(side), _ptr(buf), _stream_handle()))

    def set_arithmetic(self, mode):
        """"exact" / 0: the reference's arithmetic operation for operation (bit-identical periodic BGK flows);
        "fast" / 1: the same collision to rounding level in half the instructions (BGK, periodic 3-D plans)"""
        mode = {"exact": 0, "fast": 1}.get(mode, mode)
        if not experiments_built():
            if int(mode) == 0:
                return
            raise NativeEngineError("fast arithmetic: a kernel of the experiments build (make -C lettuce_amd/csrc "
                                    "EXPERIMENTS=1): it missed its bar (DESIGN.md section 4)")
        self._check(self.lib.lt_plan_set_arithmetic(self._handle, int(mode)))

    def set_canary(self, mode: int = 1):
        """first-use check of the masked two-step kernels: 1 = on (default), 0 = trust the kernel, 2 = report a
        mismatch without launching (test hook)"""
        self._check(self.lib.lt_plan_set_canary(self._handle, int(mode)))

    def canary_status(self) -> dict:
        """{"status": 0 not run / 1 passed / 2 skipped / -1 failed, "mismatches": n, "message": reason}"""
        status, bad, msg = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_char_p()
        self._check(self.lib.lt_plan_canary_status(self._handle, ctypes.byref(status), ctypes.byref(bad), ctypes.byref(msg)))
        return {"status": int(status.value), "mismatches": int(bad.value), "message": (msg.value or b"").decode("utf-8", "replace")}

    def set_residency(self, workgroups_per_cu: int = -1):
        """-1 automatic, 0 no cap, 2..8 workgroups resident per CU for the chip-filling launches"""
        self._check(self.lib.lt_plan_set_residency(self._handle, int(workgroups_per_cu)))
