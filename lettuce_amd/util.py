"""Exceptions, warnings and the small numerical helpers the hot path's callers need.

Mirrors the public names of lettuce/util/utility.py:21-34 (exception/warning classes),
:37-99 (``torch_gradient``) and :158-161 (``append_axes``) of the reference.  IO helpers
(HDF5, datasets), the Jacobi pressure solver and moment transforms are out of scope
(SURVEY.md section 2).
"""
import inspect

import torch

from ._errors import LettuceException, LettuceWarning, InefficientCodeWarning, ExperimentalWarning
from ._native import NativeEngineError

__all__ = ["LettuceException", "LettuceWarning", "InefficientCodeWarning", "ExperimentalWarning",
           "NativeEngineError", "torch_gradient", "append_axes", "get_subclasses"]


def get_subclasses(cls, module):
    for _, obj in inspect.getmembers(module):
        if hasattr(obj, "__bases__") and cls in obj.__bases__:
            yield obj


# central finite-difference weights on a periodic grid, keyed by order of accuracy:
# (weight, offset) pairs such that d/dx g(x) ~ sum w * g(x + offset)
_CENTRAL = {
    2: ((-1 / 2, -1), (1 / 2, 1)),
    4: ((1 / 12, -2), (-2 / 3, -1), (2 / 3, 1), (-1 / 12, 2)),
    6: ((-1 / 60, -3), (3 / 20, -2), (-3 / 4, -1), (3 / 4, 1), (-3 / 20, 2), (1 / 60, 3)),
}


def torch_gradient(f, dx=1, order=2):
    """First derivative of a periodic 2-D/3-D field along every axis; returns ``[dim, *f.shape]``.

    Same stencils, term order and final ``* 1/dx`` as lettuce/util/utility.py:37-99
    (``g(x + k)`` is ``roll(g, -k)``), so that TGV initialisation matches to rounding."""
    if f.ndim not in (2, 3):
        raise LettuceException("Invalid dimension!")
    if order not in _CENTRAL:
        raise LettuceException(f"order {order} not implemented (2, 4, 6)")
    with torch.no_grad():
        out = torch.empty((f.ndim,) + tuple(f.shape), dtype=f.dtype, device=f.device)
        scale = torch.tensor(1.0 / dx, dtype=f.dtype, device=f.device)
        for axis in range(f.ndim):
            acc = None
            for weight, offset in _CENTRAL[order]:
                term = weight * torch.roll(f, shifts=-offset, dims=axis)
                acc = term if acc is None else acc + term
            out[axis] = acc * scale
    return out


def append_axes(array, n):
    """Append ``n`` singleton axes (lettuce/util/utility.py:158-161)."""
    index = (Ellipsis,) + (None,) * n
    return array[index]
