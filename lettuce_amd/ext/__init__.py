"""Operators, flows and reporters (flat namespace, like ``lettuce.ext``)."""
from .._stencil import D1Q3, D2Q9, D3Q15, D3Q19, D3Q27
from ._equilibrium import *
from ._collision import *
from ._boundary import *
from ._flows import *
from ._reporter import *
