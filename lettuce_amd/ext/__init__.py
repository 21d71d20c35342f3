"""Operators, flows and reporters of the hot path, re-exported under one namespace so that
``from lettuce_amd.ext import BGKCollision`` works like ``from lettuce.ext import BGKCollision``."""
import importlib

_MODULES = ("_equilibrium", "_collision", "_boundary", "_flows", "_reporter")
__all__ = ["D1Q3", "D2Q9", "D3Q15", "D3Q19", "D3Q27"]

from .._stencil import D1Q3, D2Q9, D3Q15, D3Q19, D3Q27  # noqa: E402,F401

for _name in _MODULES:
    _module = importlib.import_module(f"{__name__}.{_name}")
    for _symbol in _module.__all__:
        globals()[_symbol] = getattr(_module, _symbol)
    __all__ += list(_module.__all__)
del _name, _module, _symbol
