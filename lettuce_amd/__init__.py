"""MI355X-native lattice-Boltzmann stream-and-collide engine behind lettuce's Python API.

``import lettuce_amd as lt`` gives the flat namespace of the reference
(lettuce/__init__.py:10-21): ``lt.Context``, ``lt.D3Q19``, ``lt.TaylorGreenVortex``,
``lt.BGKCollision``, ``lt.Simulation`` ...  With ``Context(use_native=True)`` (the default
when a GPU is visible) the per-step hot loop runs in hand-written gfx950 HIP kernels
(lettuce_amd/csrc, C ABI in include/lettuce_hip.h).
"""
__version__ = "0.1.0"

from .util import *
from ._context import *
from ._stencil import *
from ._unit import *
from ._flow import *
from ._simulation import *
from .ext import *
from ._slab import *
from . import util, ext
