"""Lattice constants.

Public surface of lettuce/_stencil.py:12-46 and lettuce/ext/_stencil/*.py: classes
``D1Q3, D2Q9, D3Q15, D3Q19, D3Q27`` with list attributes ``e, w, opposite`` and
``cs = 1/sqrt(3)``; the velocity ORDER is part of the contract (it is the q index of
``flow.f``) and is identical to the reference's.  The HIP engine has kernels for all five
(lettuce_amd/csrc/lattice.hpp holds the same tables).
"""
from abc import ABC
from typing import List

import numpy as np
import torch

__all__ = ["Stencil", "TorchStencil", "D1Q3", "D2Q9", "D3Q15", "D3Q19", "D3Q27"]

# building blocks of the 3-D sets, in the reference's order
_REST = [[0, 0, 0]]
_FACES = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
_EDGES = [[0, 1, 1], [0, -1, -1], [0, 1, -1], [0, -1, 1], [1, 0, 1], [-1, 0, -1],
          [1, 0, -1], [-1, 0, 1], [1, 1, 0], [-1, -1, 0], [1, -1, 0], [-1, 1, 0]]
_CORNERS = [[1, 1, 1], [-1, -1, -1], [1, 1, -1], [-1, -1, 1], [1, -1, 1], [-1, 1, -1],
            [1, -1, -1], [-1, 1, 1]]


def _swap_pairs(q):
    """opposite table of a set stored as rest + (v, -v) pairs"""
    table = [0]
    for k in range(1, q, 2):
        table.extend((k + 1, k))
    return table


class Stencil(ABC):
    e: List[List[int]]
    w: List[float]
    opposite: List[int]
    cs: float = 1 / np.sqrt(3.0)

    @property
    def d(self):
        return len(self.e[0])

    @property
    def q(self):
        return len(self.e)


class TorchStencil:
    """e, w, opposite as tensors on the context's device (lettuce/_stencil.py:30-46)."""
    cs: float = 1 / np.sqrt(3.0)

    def __init__(self, stencil: "Stencil", context: "Context"):
        self.e = context.convert_to_tensor(stencil.e)
        self.w = context.convert_to_tensor(stencil.w)
        self.opposite = context.convert_to_tensor(stencil.opposite)

    @property
    def d(self):
        return self.e.shape[1]

    @property
    def q(self):
        return self.e.shape[0]


class D1Q3(Stencil):
    def __init__(self):
        self.e = [[0], [1], [-1]]
        self.w = [2.0 / 3.0, 1.0 / 6.0, 1.0 / 6.0]
        self.opposite = [0, 2, 1]


class D2Q9(Stencil):
    def __init__(self):
        axes = [[1, 0], [0, 1], [-1, 0], [0, -1]]
        diagonals = [[1, 1], [-1, 1], [-1, -1], [1, -1]]
        self.e = [[0, 0]] + axes + diagonals
        self.w = [4.0 / 9.0] + [1.0 / 9.0] * 4 + [1.0 / 36.0] * 4
        self.opposite = [0, 3, 4, 1, 2, 7, 8, 5, 6]


class D3Q15(Stencil):
    def __init__(self):
        self.e = [list(v) for v in _REST + _FACES + _CORNERS]
        self.w = [2.0 / 9.0] + [1.0 / 9.0] * 6 + [1.0 / 72.0] * 8
        self.opposite = _swap_pairs(15)


class D3Q19(Stencil):
    def __init__(self):
        self.e = [list(v) for v in _REST + _FACES + _EDGES]
        self.w = [1.0 / 3.0] + [1.0 / 18.0] * 6 + [1.0 / 36.0] * 12
        self.opposite = _swap_pairs(19)


class D3Q27(Stencil):
    def __init__(self):
        self.e = [list(v) for v in _REST + _FACES + _EDGES + _CORNERS]
        self.w = [8.0 / 27.0] + [2.0 / 27.0] * 6 + [1.0 / 54.0] * 12 + [1.0 / 216.0] * 8
        self.opposite = _swap_pairs(27)
