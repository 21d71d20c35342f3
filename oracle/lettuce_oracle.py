"""CPU oracle for the lattice-Boltzmann stream-and-collide hot path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (torch-CPU whole-field
ops, so that it costs what the reference's PyTorch CPU path costs) of the
algorithm of PhiSpel/lettuce @ 2024_10_08 for the path in SURVEY.md section 8.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; nothing under ``lettuce_amd/`` does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against ``tests/golden/*.npz``, which were produced by running the
reference's own CPU path in the build container (``oracle/gen_golden.py``;
the reference itself never travels).  The reference holds no stored golden
vectors of its own, so its invariants (mass/momentum
conservation, bounce-back = opposite permutation, analytic ABB formula) are
additionally restated in ``tests/test_reference_invariants.py``.

All citations are ``path:line`` under ``/root/reference/``.

State convention (same as the reference): ``f[q, x, (y, (z))]`` holds the
post-streaming / pre-collision populations; one step is
``collide -> boundaries -> stream`` (lettuce/_simulation.py:92-94,160-189).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

# --------------------------------------------------------------------------- #
# lattice constants  (lettuce/ext/_stencil/d2q9.py:8-10, d3q19.py:8-13,
# d3q27.py:8-12, d1q3.py, d3q15.py; cs: lettuce/_stencil.py:17)
# --------------------------------------------------------------------------- #

CS = 1.0 / np.sqrt(3.0)     # numpy double, exactly as the reference builds it
CS2 = CS ** 2               # 0.33333333333333337 (not 1/3)
CS4 = CS ** 4


def _pairs_opposite(q: int) -> List[int]:
    """3-D stencils of the reference pair (1,2),(3,4),... as opposites."""
    out = [0]
    for k in range(1, q, 2):
        out += [k + 1, k]
    return out


_AXIS3 = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
_EDGE3 = [[0, 1, 1], [0, -1, -1], [0, 1, -1], [0, -1, 1],
          [1, 0, 1], [-1, 0, -1], [1, 0, -1], [-1, 0, 1],
          [1, 1, 0], [-1, -1, 0], [1, -1, 0], [-1, 1, 0]]
_CORNER3 = [[1, 1, 1], [-1, -1, -1], [1, 1, -1], [-1, -1, 1],
            [1, -1, 1], [-1, 1, -1], [1, -1, -1], [-1, 1, 1]]


@dataclass(frozen=True)
class Lattice:
    name: str
    e: tuple            # q x d integers
    w: tuple            # q doubles
    opposite: tuple     # q integers

    @property
    def q(self):
        return len(self.e)

    @property
    def d(self):
        return len(self.e[0])


def _lat(name, e, w, opp):
    return Lattice(name, tuple(tuple(v) for v in e), tuple(w), tuple(opp))


LATTICES = {
    "D1Q3": _lat("D1Q3", [[0], [1], [-1]], [2 / 3, 1 / 6, 1 / 6], [0, 2, 1]),
    "D2Q9": _lat("D2Q9",
                 [[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1],
                  [1, 1], [-1, 1], [-1, -1], [1, -1]],
                 [4 / 9] + [1 / 9] * 4 + [1 / 36] * 4,
                 [0, 3, 4, 1, 2, 7, 8, 5, 6]),
    "D3Q15": _lat("D3Q15", [[0, 0, 0]] + _AXIS3 + _CORNER3,
                  [2 / 9] + [1 / 9] * 6 + [1 / 72] * 8, _pairs_opposite(15)),
    "D3Q19": _lat("D3Q19", [[0, 0, 0]] + _AXIS3 + _EDGE3,
                  [1 / 3] + [1 / 18] * 6 + [1 / 36] * 12, _pairs_opposite(19)),
    "D3Q27": _lat("D3Q27", [[0, 0, 0]] + _AXIS3 + _EDGE3 + _CORNER3,
                  [8 / 27] + [2 / 27] * 6 + [1 / 54] * 12 + [1 / 216] * 8,
                  _pairs_opposite(27)),
}


def lattice_tensors(lat: Lattice, dtype):
    """e, w as tensors of the working dtype (lettuce/_stencil.py:37-41)."""
    return (torch.tensor(lat.e, dtype=dtype), torch.tensor(lat.w, dtype=dtype))


# --------------------------------------------------------------------------- #
# unit conversion scalars  (lettuce/_unit.py:34-60,62-68,99-101,132-135)
# --------------------------------------------------------------------------- #

@dataclass
class Units:
    reynolds_number: float
    mach_number: float
    characteristic_length_lu: float = 1
    characteristic_length_pu: float = 1
    characteristic_velocity_pu: float = 1
    characteristic_density_lu: float = 1
    characteristic_density_pu: float = 1

    @property
    def u_char_lu(self):
        return CS * self.mach_number

    @property
    def viscosity_lu(self):
        return self.characteristic_length_lu * self.u_char_lu / self.reynolds_number

    @property
    def tau(self):
        return self.viscosity_lu / CS ** 2 + 0.5

    def velocity_to_lu(self, u_pu):
        return u_pu / self.characteristic_velocity_pu * self.u_char_lu

    def pressure_pu_to_density_lu(self, p_pu):
        p_char_pu = self.characteristic_density_pu * self.characteristic_velocity_pu ** 2
        p_char_lu = self.characteristic_density_lu * self.u_char_lu ** 2
        return p_pu / p_char_pu * p_char_lu / CS ** 2 + self.characteristic_density_lu

    def length_to_pu(self, l_lu):
        return l_lu * self.characteristic_length_pu / self.characteristic_length_lu

    def time_to_pu(self, t_lu):
        a = self.characteristic_length_lu / self.u_char_lu
        b = self.characteristic_length_pu / self.characteristic_velocity_pu
        return t_lu / a * b

    def incompressible_energy_to_pu(self, e_lu):
        return e_lu * self.characteristic_velocity_pu ** 2 / self.u_char_lu ** 2


# --------------------------------------------------------------------------- #
# macroscopic moments  (lettuce/_flow.py:136-138,152-172,178-181)
# --------------------------------------------------------------------------- #

def density(f):
    """rho = sum_q f_q, shape [1, *res]."""
    return torch.sum(f, dim=0)[None, ...]


def momentum(f, e):
    """j_d = sum_q e_qd f_q, shape [d, *res]."""
    return torch.einsum("qd,q...->d...", e, f)


def velocity(f, e, rho=None):
    rho = density(f) if rho is None else rho
    return momentum(f, e) / rho


def incompressible_energy(f, e):
    """0.5 * u.u per node (lettuce/_flow.py:178-181)."""
    u = velocity(f, e)
    return 0.5 * torch.einsum("d...,d...->...", u, u)


def kinetic_energy_pu(f, lat: Lattice, units: Units):
    """IncompressibleKineticEnergy observable
    (lettuce/ext/_reporter/observable_reporter.py:34-42)."""
    e, _ = lattice_tensors(lat, f.dtype)
    dx = units.length_to_pu(1.0)
    k = units.incompressible_energy_to_pu(torch.sum(incompressible_energy(f, e)))
    k = k * dx ** lat.d
    return k


# --------------------------------------------------------------------------- #
# equilibrium  (lettuce/ext/_equilibrium/quadratic_equilibrium.py:11-25)
# --------------------------------------------------------------------------- #

def quadratic_equilibrium(rho, u, e, w):
    """feq_q = w_q rho ((2 e.u - u.u)/(2 cs^2) + 0.5 (e.u/cs^2)^2 + 1).

    ``rho`` may be [1,*res] or 0-d, ``u`` [d,*res] or [d] (inlet boundary)."""
    exu = torch.tensordot(e, u, dims=1)
    uxu = torch.einsum("d...,d...->...", u, u)
    inner = rho * ((2 * exu - uxu) / (2 * CS ** 2) + 0.5 * (exu / (CS ** 2)) ** 2 + 1)
    if inner.dim() == 1:
        return w * inner
    return torch.einsum("q,q...->q...", w, inner)


# --------------------------------------------------------------------------- #
# collisions
# --------------------------------------------------------------------------- #

def bgk(f, tau, e, w):
    """f - (f - feq)/tau  (lettuce/ext/_collision/bgk_collision.py:17-22)."""
    u = velocity(f, e)
    feq = quadratic_equilibrium(density(f), u, e, w)
    return f - 1.0 / tau * (f - feq)


def _kbc_moments(f, e):
    """The seven KBC moments of f, rho-normalised except m000
    (lettuce/ext/_collision/kbc_collision.py:25-39,44-52,76-81)."""
    d = e.shape[1]
    rho = torch.sum(f, dim=0)

    def mom(*powers):
        coeff = torch.ones(e.shape[0], dtype=f.dtype)
        for ax, p in enumerate(powers):
            coeff = coeff * e[:, ax] ** p
        return torch.einsum("q,q...->...", coeff, f) / rho

    if d == 3:
        return rho, dict(xx=mom(2, 0, 0), yy=mom(0, 2, 0), zz=mom(0, 0, 2),
                         xy=mom(1, 1, 0), xz=mom(1, 0, 1), yz=mom(0, 1, 1))
    return rho, dict(xx=mom(2, 0), yy=mom(0, 2), xy=mom(1, 1))


def _kbc_shear_part(f, e):
    """s_q built from (T, N, Pi) (kbc_collision.py:44-94); zero for the corner
    populations 19..26 of D3Q27."""
    rho, m = _kbc_moments(f, e)
    s = torch.zeros_like(f)
    if e.shape[1] == 3:
        T = m["xx"] + m["yy"] + m["zz"]
        nxz = m["xx"] - m["zz"]
        nyz = m["yy"] - m["zz"]
        s[0] = rho * -T
        s[1] = 1. / 6. * rho * (2 * nxz - nyz + T)
        s[2] = s[1]
        s[3] = 1. / 6. * rho * (2 * nyz - nxz + T)
        s[4] = s[3]
        s[5] = 1. / 6. * rho * (-nxz - nyz + T)
        s[6] = s[5]
        s[7] = 1. / 4 * rho * m["yz"]
        s[8] = s[7]
        s[9] = -1. / 4 * rho * m["yz"]
        s[10] = s[9]
        s[11] = 1. / 4 * rho * m["xz"]
        s[12] = s[11]
        s[13] = -1. / 4 * rho * m["xz"]
        s[14] = s[13]
        s[15] = 1. / 4 * rho * m["xy"]
        s[16] = s[15]
        s[17] = -1. / 4 * rho * m["xy"]
        s[18] = s[17]
    else:
        T = m["xx"] + m["yy"]
        N = m["xx"] - m["yy"]
        s[0] = rho * -T
        s[1] = 1. / 2. * rho * (0.5 * (T + N))
        s[2] = 1. / 2. * rho * (0.5 * (T - N))
        s[3] = s[1]
        s[4] = s[2]
        s[5] = 1. / 4. * rho * m["xy"]
        s[6] = -s[5]
        s[7] = s[5]
        s[8] = -s[7]
    return s


def kbc(f, tau, e, w):
    """Entropic KBC (kbc_collision.py:96-160).  ``tau`` here is the value the
    reference actually uses: flow.units.relaxation_parameter_lu (:97-99)."""
    beta = 1. / (2 * tau)
    rho = density(f)
    feq = quadratic_equilibrium(rho, velocity(f, e), e, w)
    delta_s = _kbc_shear_part(f, e) - _kbc_shear_part(feq, e)
    delta_h = f - feq - delta_s
    sum_s = density(delta_s * delta_h / feq)
    sum_h = density(delta_h * delta_h / feq)
    gamma = 1. / beta - (2 - 1. / beta) * sum_s / sum_h
    gamma[gamma < 1e-15] = 2.0
    gamma[torch.isnan(gamma)] = 2.0
    return f - beta * (2 * delta_s + gamma * delta_h)


# --------------------------------------------------------------------------- #
# boundaries
# --------------------------------------------------------------------------- #

@dataclass
class OracleBoundary:
    """kind: 'bounce_back' | 'equilibrium_pu' | 'abb_outlet'.

    ``sort_key`` reproduces ``str(boundary)`` ordering of the reference
    (lettuce/_simulation.py:57-58): the default repr sorts by module path, so
    anti_bounce_back_outlet < bounce_back_boundary < equilibrium_boundary_pu."""
    kind: str
    mask: Optional[torch.Tensor] = None          # bool [*res] (bb, eq)
    velocity_pu: Optional[torch.Tensor] = None   # eq: [d] or broadcastable
    pressure_pu: Optional[torch.Tensor] = None   # eq: 0-d or broadcastable
    direction: Optional[Sequence[int]] = None    # abb: one-hot +-1
    no_streaming_mask: Optional[torch.Tensor] = None  # explicit override

    @property
    def sort_key(self):
        return {"abb_outlet": "0", "bounce_back": "1", "equilibrium_pu": "2"}[self.kind]


def bounce_back(f, lat: Lattice):
    """f[opposite] (lettuce/ext/_boundary/bounce_back_boundary.py:17-18)."""
    return f[list(lat.opposite)]


def equilibrium_pu(f, b: OracleBoundary, units: Units, e, w):
    """feq(rho(p), u_lu(v)) broadcast to the field
    (lettuce/ext/_boundary/equilibrium_boundary_pu.py:27-32)."""
    rho = units.pressure_pu_to_density_lu(b.pressure_pu)
    u = units.velocity_to_lu(b.velocity_pu)
    feq = quadratic_equilibrium(rho, u, e, w)
    if feq.dim() == 1:
        feq = feq.reshape([-1] + [1] * (f.dim() - 1))
    return feq * torch.ones_like(f)


def _abb_sets(lat: Lattice, direction):
    d = len(direction)
    vel = [q for q in range(lat.q)
           if sum(lat.e[q][a] * direction[a] for a in range(d)) > 1 - 1e-6]
    index, neighbor = [], []
    for c in direction:
        if c == 0:
            index.append(slice(None)); neighbor.append(slice(None))
        elif c == 1:
            index.append(-1); neighbor.append(-2)
        else:
            index.append(0); neighbor.append(1)
    return vel, index, neighbor


def abb_outlet_inplace(f, b: OracleBoundary, lat: Lattice, e, w):
    """Anti-bounce-back outlet; writes into ``f`` and returns it
    (lettuce/ext/_boundary/anti_bounce_back_outlet.py:72-91)."""
    vel, index, neighbor = _abb_sets(lat, list(b.direction))
    u = velocity(f, e)
    sl = [slice(None)]
    u_w = u[tuple(sl + index)] + 0.5 * (u[tuple(sl + index)] - u[tuple(sl + neighbor)])
    rho_w = density(f)[tuple(sl + index)]                      # [1, plane]
    ev = e[vel]                                                # [c, d]
    eu = torch.einsum("cd,d...->c...", ev, u_w)                # [c, plane]
    wv = w[vel].reshape([-1] + [1] * (u_w.dim() - 1))
    new = (-f[tuple([vel] + index)]
           + wv * rho_w * (2 + eu ** 2 / CS ** 4
                           - (torch.norm(u_w, dim=0) / CS) ** 2))
    f[tuple([[lat.opposite[q] for q in vel]] + index)] = new
    return f


def abb_masks(shape_f, b: OracleBoundary, lat: Lattice):
    """(no_collision_mask [*res], no_streaming_mask [q,*res])
    (anti_bounce_back_outlet.py:93-103)."""
    vel, index, _ = _abb_sets(lat, list(b.direction))
    nsm = torch.zeros(list(shape_f), dtype=torch.bool)
    nsm[tuple([[lat.opposite[q] for q in vel]] + index)] = True
    ncm = torch.zeros(list(shape_f[1:]), dtype=torch.bool)
    ncm[tuple(index)] = True
    return ncm, nsm


# --------------------------------------------------------------------------- #
# the step  (lettuce/_simulation.py:57-86,160-189)
# --------------------------------------------------------------------------- #

@dataclass
class OracleSimulation:
    lat: Lattice
    f: torch.Tensor
    collision: str                       # 'bgk' | 'kbc' | 'none'
    tau: float
    units: Optional[Units] = None
    boundaries: List[OracleBoundary] = field(default_factory=list)
    no_collision_mask: Optional[torch.Tensor] = None
    no_streaming_mask: Optional[torch.Tensor] = None
    i: int = 0

    def __post_init__(self):
        self.e, self.w = lattice_tensors(self.lat, self.f.dtype)
        self.boundaries = sorted(self.boundaries, key=lambda b: b.sort_key)
        if self.boundaries:
            res = list(self.f.shape[1:])
            ncm = torch.zeros(res, dtype=torch.uint8)
            nsm = torch.zeros([self.lat.q] + res, dtype=torch.uint8)
            for idx, b in enumerate(self.boundaries, start=1):
                if b.kind == "abb_outlet":
                    m, s = abb_masks(self.f.shape, b, self.lat)
                else:
                    m, s = b.mask, b.no_streaming_mask
                if m is not None:
                    ncm[m.to(torch.bool)] = idx
                if s is not None:
                    nsm |= s.to(torch.uint8)
            self.no_collision_mask, self.no_streaming_mask = ncm, nsm

    # -- pieces -------------------------------------------------------------
    def _collision(self, f):
        if self.collision == "bgk":
            return bgk(f, self.tau, self.e, self.w)
        if self.collision == "kbc":
            return kbc(f, self.tau, self.e, self.w)
        return f

    def _boundary(self, b, f):
        if b.kind == "bounce_back":
            return bounce_back(f, self.lat)
        if b.kind == "equilibrium_pu":
            return equilibrium_pu(f, b, self.units, self.e, self.w)
        return abb_outlet_inplace(f, b, self.lat, self.e, self.w)

    def collide(self):
        """_simulation.py:177-189."""
        if self.no_collision_mask is None:
            self.f = self._collision(self.f)
            for b in self.boundaries:
                self.f = self._boundary(b, self.f)
        else:
            ncm = self.no_collision_mask
            self.f = torch.where(ncm == 0, self._collision(self.f), self.f)
            for idx, b in enumerate(self.boundaries, start=1):
                self.f = torch.where(ncm == idx, self._boundary(b, self.f), self.f)
        return self.f

    def stream(self):
        """_simulation.py:160-175: population q moves by +e_q, periodic; where the
        no-streaming mask is set the destination keeps its own value."""
        dims = tuple(range(self.lat.d))
        for q in range(1, self.lat.q):
            moved = torch.roll(self.f[q], shifts=tuple(self.lat.e[q]), dims=dims)
            if self.no_streaming_mask is None:
                self.f[q] = moved
            else:
                self.f[q] = torch.where(self.no_streaming_mask[q] == 1, self.f[q], moved)
        return self.f

    def step(self, n=1):
        for _ in range(n):
            self.collide()
            self.stream()
            self.i += 1
        return self.f


# --------------------------------------------------------------------------- #
# "next" row F1: Taylor-Green initial condition
# (lettuce/ext/_flows/taylorgreen.py:43-94, lettuce/_flow.py:106-122,309-336,
#  lettuce/util/utility.py:37-99)
# --------------------------------------------------------------------------- #

_FD6 = ((-1 / 60, 3), (3 / 20, 2), (-3 / 4, 1), (3 / 4, -1), (-3 / 20, -2), (1 / 60, -3))


def periodic_gradient6(g):
    """6th-order periodic central differences, dx = 1; out[a] = d g / d x_a."""
    out = []
    for axis in range(g.dim()):
        acc = None
        for coeff, shift in _FD6:
            term = coeff * torch.roll(g, shifts=shift, dims=axis)
            acc = term if acc is None else acc + term
        # the reference adds two zero-weighted rolls as well; x + 0*y == x exactly
        out.append(acc * torch.tensor(1.0, dtype=g.dtype))
    return torch.stack(out)


def enstrophy_pu(f, lat: Lattice, units: Units):
    """Enstrophy observable (lettuce/ext/_reporter/observable_reporter.py:45-68): squared vorticity of
    u_pu from 6th-order periodic differences (util/utility.py:37-99), summed, times dx^d."""
    e, _ = lattice_tensors(lat, f.dtype)
    u = velocity(f, e) * (units.characteristic_velocity_pu / units.u_char_lu)
    dx = units.length_to_pu(1.0)
    inv = torch.tensor(1.0 / dx, dtype=f.dtype)
    g = [periodic_gradient6(u[a]) * inv for a in range(lat.d)]
    vort = torch.sum((g[0][1] - g[1][0]) * (g[0][1] - g[1][0]))
    if lat.d == 3:
        vort = vort + torch.sum((g[2][1] - g[1][2]) * (g[2][1] - g[1][2])
                                + ((g[0][2] - g[2][0]) * (g[0][2] - g[2][0])))
    return vort * dx ** lat.d


def mass_observable(f, no_mass_mask=None):
    """Mass observable (observable_reporter.py:140-158): the first / last index of the two last axes
    does not count; the nodes of ``no_mass_mask`` are subtracted (wherever they are)."""
    mass = f[..., 1:-1, 1:-1].sum()
    if no_mass_mask is not None:
        mass = mass - (f * no_mass_mask.to(dtype=torch.float)).sum()
    return mass


def tgv_units(resolution, reynolds_number, mach_number):
    return Units(reynolds_number, mach_number,
                 characteristic_length_lu=resolution[0],
                 characteristic_length_pu=2 * math.pi,
                 characteristic_velocity_pu=1)


def tgv_initial_pu(resolution, dtype):
    axes = [torch.linspace(0, 2 * math.pi * (1 - 1 / n), steps=n, dtype=dtype)
            for n in resolution]
    g = torch.meshgrid(*axes, indexing="ij")
    if len(resolution) == 2:
        u = torch.stack([torch.cos(g[0]) * torch.sin(g[1]),
                         -torch.sin(g[0]) * torch.cos(g[1])])
        p = -torch.stack([0.25 * (torch.cos(2 * g[0]) + torch.cos(2 * g[1]))])
    else:
        u = torch.stack([torch.sin(g[0]) * torch.cos(g[1]) * torch.cos(g[2]),
                         -torch.cos(g[0]) * torch.sin(g[1]) * torch.cos(g[2]),
                         torch.zeros_like(g[0])])
        p = torch.stack([1 / 16. * (torch.cos(2 * g[0]) + torch.cos(2 * g[1]))
                         * (torch.cos(2 * g[2]) + 2)])
    return p, u


def initialize_from_pu(p, u, lat: Lattice, units: Units, dtype, fneq=True):
    """Flow.initialize + initialize_f_neq (lettuce/_flow.py:106-122,309-336)."""
    e, w = lattice_tensors(lat, dtype)
    rho0 = units.pressure_pu_to_density_lu(p).to(dtype)
    u0 = units.velocity_to_lu(u).to(dtype)
    f = quadratic_equilibrium(rho0, u0, e, w)
    if not fneq:
        return f
    rho = density(f)
    uu = velocity(f, e)
    S = torch.cat([periodic_gradient6(uu[a])[None, ...] for a in range(lat.d)])
    pi1 = 1.0 * units.tau * rho * S / CS ** 2
    # the reference builds the identity with torch's default dtype (float32), so the
    # cs^2 on the diagonal is rounded to fp32 even in an fp64 run (_flow.py:328-330)
    Q = (torch.einsum("ia,ib->iab", e, e) - torch.eye(lat.d) * CS ** 2)
    pi1q = torch.einsum("ab...,iab->i...", pi1, Q)
    f_neq = torch.einsum("i,i...->i...", w, pi1q)
    return quadratic_equilibrium(rho, uu, e, w) - f_neq


def taylor_green(resolution, reynolds_number, mach_number, lattice_name, dtype,
                 collision="bgk"):
    """Convenience: an OracleSimulation of the reference's TaylorGreenVortex."""
    lat = LATTICES[lattice_name]
    units = tgv_units(resolution, reynolds_number, mach_number)
    p, u = tgv_initial_pu(resolution, dtype)
    f = initialize_from_pu(p, u, lat, units, dtype, fneq=True)
    return OracleSimulation(lat, f, collision, units.tau, units)
