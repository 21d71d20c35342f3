"""Generate tests/golden/*.npz by running the REFERENCE's own PyTorch CPU path.

Build-container only: it imports /root/reference (read-only) with inert stubs for
the three IO modules the image lacks (h5py, pyevtk, mmh3 -- SURVEY.md 8(c),
Appendix D).  The reference never travels; only the small vectors written here
are committed.  Run:  python oracle/gen_golden.py

Each file holds inputs (initial f, masks, scalar parameters) and the
reference's outputs (f after k steps, kinetic-energy series, rho/u ...).
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    for name in ("h5py", "pyevtk", "pyevtk.hl", "mmh3"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pyevtk"].hl = sys.modules["pyevtk.hl"]
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import lettuce as lt
    return lt


lt = import_reference()
torch.set_num_threads(8)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ke(flow):
    return float(lt.IncompressibleKineticEnergy(flow)())


def enstrophy(flow):
    """the reference's Enstrophy observable (periodic flows only; observable_reporter.py:45-68)"""
    return float(lt.Enstrophy(flow)())


ONLY = sys.argv[1:]          # optional: substrings of the case names to (re)generate


def wanted(name):
    return not ONLY or any(k in name for k in ONLY)


def save(name, **arrays):
    if not wanted(name):
        return
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:44s} {os.path.getsize(path) / 1024:9.1f} KiB")


def npy(t):
    return t.detach().cpu().numpy().copy()


DT = {"f64": torch.float64, "f32": torch.float32}


# --------------------------------------------------------------------------- #
def run_series(flow, collision, snapshots, energy_every, periodic=False):
    """Step to max(snapshots); record f at the snapshot steps and KE series (periodic flows: also
    the enstrophy and the Mass observable of the reference at the same steps)."""
    sim = quiet(lt.Simulation, flow, collision, [])
    out = {"f0": npy(flow.f)}
    steps, energies = [0], [ke(flow)]
    ens = [enstrophy(flow)] if periodic else None
    mass = [float(lt.Mass(flow)(flow.f))] if periodic else None
    last = max(snapshots)
    for i in range(1, last + 1):
        quiet(sim, 1)
        if i in snapshots:
            out[f"f{i}"] = npy(flow.f)
        if i % energy_every == 0:
            steps.append(i)
            energies.append(ke(flow))
            if periodic:
                ens.append(enstrophy(flow))
                mass.append(float(lt.Mass(flow)(flow.f)))
    out["energy_steps"] = np.array(steps)
    out["energy_pu"] = np.array(energies, dtype=np.float64)
    if periodic:
        out["enstrophy_pu"] = np.array(ens, dtype=np.float64)
        out["mass_observable"] = np.array(mass, dtype=np.float64)
    out["rho_final"] = npy(flow.rho())
    out["u_final"] = npy(flow.u())
    if sim.no_collision_mask is not None:
        out["no_collision_mask"] = npy(sim.no_collision_mask)
        out["no_streaming_mask"] = np.packbits(npy(sim.no_streaming_mask).astype(bool), axis=None)
        out["no_streaming_mask_shape"] = np.array(sim.no_streaming_mask.shape)
    return out


def tgv_case(name, res, stencil, re, ma, dt, coll, snapshots, energy_every):
    if not wanted(name):
        return
    ctx = lt.Context(device="cpu", dtype=DT[dt], use_native=False)
    flow = quiet(lt.TaylorGreenVortex, ctx, res, re, ma, stencil)
    tau = flow.units.relaxation_parameter_lu
    collision = lt.BGKCollision(tau) if coll == "bgk" else lt.KBCCollision()
    out = run_series(flow, collision, snapshots, energy_every, periodic=True)
    save(name, tau=np.float64(tau), reynolds=np.float64(re), mach=np.float64(ma),
         resolution=np.array(flow.resolution), **out)


def obstacle_case(name, res, stencil, dt, coll, snapshots, domain_length_x, center, radius):
    if not wanted(name):
        return
    ctx = lt.Context(device="cpu", dtype=DT[dt], use_native=False)
    flow = quiet(lt.Obstacle, ctx, list(res), 100, 0.1, domain_length_x, stencil=stencil)
    grid = flow.grid
    r2 = sum((g - c) ** 2 for g, c in zip(grid, center))
    flow.mask = (r2 < radius ** 2)
    quiet(flow.initialize)   # initial_pu depends on the mask (obstacle.py:94-99)
    tau = flow.units.relaxation_parameter_lu
    collision = lt.BGKCollision(tau) if coll == "bgk" else lt.KBCCollision()
    out = run_series(flow, collision, snapshots, 1)
    b_names = [type(b).__name__ for b in
               sorted(flow.boundaries, key=lambda b: str(b))]
    save(name, tau=np.float64(tau), obstacle_mask=npy(flow.mask),
         boundary_order=np.array(b_names),
         u_char_lu=np.float64(flow.units.characteristic_velocity_lu),
         char_length_lu=np.float64(flow.char_length_lu),
         domain_length_x=np.float64(domain_length_x),
         resolution=np.array(flow.resolution), **out)


# --------------------------------------------------------------------------- #
# analogues of the reference's tests/native/* (hand-set f, 16x16 D2Q9)
class _Dummy(lt.ExtFlow):
    def __init__(self, context, resolution=16, stencil=None, re=1.0, ma=1.0):
        lt.ExtFlow.__init__(self, context, resolution, re, ma, stencil)

    def make_resolution(self, resolution, stencil=None):
        if isinstance(resolution, int):
            return [resolution] * (stencil.d if stencil is not None else 2)
        return resolution

    def make_units(self, reynolds_number, mach_number, _):
        return lt.UnitConversion(reynolds_number=reynolds_number, mach_number=mach_number)

    def initial_pu(self):
        ...

    def initialize(self):
        self.fill()

    def fill(self):
        ...

    @property
    def boundaries(self):
        return []


def hand_set_cases():
    ctx = lt.Context(device="cpu", dtype=torch.float32, use_native=False)

    # tagged populations streamed once (tests/native/test_native_streaming.py:9-51)
    class Tagged(_Dummy):
        def fill(self):
            self.f.zero_()
            for q in range(9):
                self.f[q, q + 1, q + 1] = q + 1.0
    fl = Tagged(ctx)
    f0 = npy(fl.f)
    quiet(lt.Simulation(fl, lt.NoCollision(), []), 1)
    save("native_streaming_d2q9_f32", f0=f0, f1=npy(fl.f))

    # one BGK step on a bump, tau = 2 (tests/native/test_native_bgk_collision.py:26-61)
    class Bump(_Dummy):
        def fill(self):
            self.f[:, :, :] = 1.0
            self.f[:, 2, 2] = 2.0
    fl = Bump(ctx)
    f0 = npy(fl.f)
    quiet(lt.Simulation(fl, lt.BGKCollision(2.0), []), 1)
    save("native_bgk_d2q9_f32", f0=f0, f1=npy(fl.f), tau=np.float64(2.0))

    # bounce-back box, two steps, no collision (tests/native/test_native_bounce_back.py:12-74)
    class BoxBB(lt.BounceBackBoundary):
        def make_no_collision_mask(self, shape, context):
            m = context.zero_tensor(shape, dtype=bool)
            m[0, :] = True; m[:, 0] = True; m[2:, :] = True; m[:, 2:] = True
            return m

    class BBFlow(_Dummy):
        def fill(self):
            self.f.zero_()
            self.f[:, 1, 1] = 1.0

        @property
        def boundaries(self):
            return [BoxBB(torch.ones(self.resolution))]
    fl = BBFlow(ctx)
    sim = lt.Simulation(fl, lt.NoCollision(), [])
    f0 = npy(fl.f)
    quiet(sim, 1); f1 = npy(fl.f)
    quiet(sim, 1); f2 = npy(fl.f)
    save("native_bounce_back_d2q9_f32", f0=f0, f1=f1, f2=f2,
         no_collision_mask=npy(sim.no_collision_mask))

    # equilibrium-PU column with an all-ones no-streaming mask, per-node u/rho tensors
    # (tests/native/test_native_equilibrium_pu.py:13-74)
    ctx64 = lt.Context(device="cpu", dtype=torch.float64, use_native=False)

    class EqCol(lt.EquilibriumBoundaryPU):
        def make_no_collision_mask(self, shape, context):
            a = context.zero_tensor(shape, dtype=bool)
            a[:, 1] = True
            return a

        def make_no_streaming_mask(self, shape, context):
            return context.one_tensor(shape, dtype=bool)

    class TGVb(lt.TaylorGreenVortex):
        _b = None

        @property
        def boundaries(self):
            return [] if self._b is None else [self._b]
    fl = quiet(TGVb, ctx64, [16, 16], 1, 0.1)
    # (the reference test passes its arguments shifted by one position and only ever runs on
    #  CUDA; here they are given by keyword so that the CPU path accepts them)
    fl._b = EqCol(ctx64, mask=None, velocity=np.ones([2, 16, 16]), pressure=np.ones([16, 16]))
    sim = lt.Simulation(fl, lt.NoCollision(), [])
    f0 = npy(fl.f)
    quiet(sim, 1)
    save("native_equilibrium_pu_d2q9_f64", f0=f0, f1=npy(fl.f),
         u_char_lu=np.float64(fl.units.characteristic_velocity_lu))

    # grid-shaped all-zero no_streaming_mask set after construction
    # (tests/native/test_native_no_streaming_mask.py:4-22); uniform flow stays put
    class Uniform(_Dummy):
        def initial_pu(self):
            u = 1.01 * np.ones([self.stencil.d] + self.resolution)
            p = 0.01 * np.ones([1] + self.resolution)
            return p, u

        def initialize(self):
            lt.Flow.initialize(self)

        def make_units(self, reynolds_number, mach_number, resolution):
            return lt.UnitConversion(reynolds_number, mach_number,
                                     characteristic_length_lu=resolution[0])
    fl = Uniform(ctx, 16, lt.D2Q9(), 1, 0.01)
    sim = lt.Simulation(fl, lt.NoCollision(), [])
    sim.no_streaming_mask = ctx.zero_tensor(fl.resolution, dtype=bool)
    f0 = npy(fl.f)
    quiet(sim, 64)
    save("native_no_streaming_mask_d2q9_f32", f0=f0, f64=npy(fl.f))


# --------------------------------------------------------------------------- #
def operator_cases():
    """Whole-field operators on a seeded random f near equilibrium: rho/j/u,
    feq, BGK, KBC, bounce-back, ABB outlet in every axis direction."""
    g = torch.Generator().manual_seed(20241008)
    for sname, res in (("D2Q9", [12, 10]), ("D3Q19", [8, 6, 10]), ("D3Q27", [8, 6, 10])):
        for dt in ("f64", "f32"):
            ctx = lt.Context(device="cpu", dtype=DT[dt], use_native=False)
            st = getattr(lt, sname)()
            fl = quiet(lt.TaylorGreenVortex, ctx, res, 50, 0.1, st)
            noise = 1 + 0.02 * (torch.rand(fl.f.shape, generator=g, dtype=torch.float64) - 0.5)
            fl.f = (fl.f.double() * noise).to(DT[dt])
            f_in = npy(fl.f)
            tau = 0.6
            out = dict(f=f_in, tau=np.float64(tau), rho=npy(fl.rho()), j=npy(fl.j()),
                       u=npy(fl.u()), feq=npy(fl.equilibrium(fl)),
                       energy=npy(fl.incompressible_energy()),
                       bgk=npy(lt.BGKCollision(tau)(fl)),
                       bounce_back=npy(lt.BounceBackBoundary(None)(fl)),
                       tau_units=np.float64(fl.units.relaxation_parameter_lu))
            if sname in ("D2Q9", "D3Q27"):
                out["kbc"] = npy(lt.KBCCollision()(fl))
            for axis in range(st.d):
                for sign in (1, -1):
                    direction = [0] * st.d
                    direction[axis] = sign
                    fl.f = torch.tensor(f_in)
                    abb = lt.AntiBounceBackOutlet(direction, fl)
                    tag = f"abb_{'xyz'[axis]}{'p' if sign > 0 else 'm'}"
                    out[tag] = npy(abb(fl))
                    out[tag + "_ncm"] = npy(abb.make_no_collision_mask(list(fl.f.shape[1:]), ctx))
                    out[tag + "_nsm"] = npy(abb.make_no_streaming_mask(list(fl.f.shape), ctx))
            save(f"operators_{sname.lower()}_{dt}", **out)


# --------------------------------------------------------------------------- #
def two_outlets_case(name, res, stencil, dt, outlets=None):
    """A flow with TWO anti-bounce-back outlets (+x and +y faces, which meet in an edge), an equilibrium
    inlet on x = 0 and a bounce-back block: the reference accepts any list of boundaries
    (lettuce/_simulation.py:57-86); among equal classes the order is that of the objects' addresses
    (str(b) of the default repr), so the order this run produced is stored with the vectors."""
    if not wanted(name):
        return
    ctx = lt.Context(device="cpu", dtype=DT[dt], use_native=False)
    d = stencil.d

    class TwoOutlets(lt.TaylorGreenVortex):
        made = None

        @property
        def boundaries(self):
            if self.made is None:
                x = self.grid[0]
                block = torch.zeros(self.resolution, dtype=torch.bool)
                block[tuple(slice(n // 2 - 1, n // 2 + 1) for n in self.resolution)] = True
                inlet = [0.3] + [0.0] * (d - 1)
                # default: +x and +y; `outlets` (round 3): any list of directions, e.g. +x, +y, -y -- three outlets on
                # two axes, whose planes meet pairwise in edges
                dirs = outlets if outlets is not None else [[1] + [0] * (d - 1), [0, 1] + [0] * (d - 2)]
                self.made = ([lt.EquilibriumBoundaryPU(self.context, torch.abs(x) < 1e-6, inlet)]
                             + [lt.AntiBounceBackOutlet(list(v), self) for v in dirs]
                             + [lt.BounceBackBoundary(block)])
            return self.made
    flow = quiet(TwoOutlets, ctx, res, 100, 0.05, stencil)
    tau = flow.units.relaxation_parameter_lu
    sim = quiet(lt.Simulation, flow, lt.BGKCollision(tau), [])
    out = {"f0": npy(flow.f)}
    for i in range(1, 7):
        quiet(sim, 1)
        if i in (1, 2, 6):
            out[f"f{i}"] = npy(flow.f)
    kinds, dirs = [], []
    for b in sim.boundaries[1:]:
        kinds.append(type(b).__name__)
        # the reference's outlet keeps `index` (-1 / 0 on its axis, slices elsewhere), not the direction
        index = getattr(b, "index", None)
        dirs.append([0] * d if index is None else
                    [0 if isinstance(i, slice) else (1 if i == -1 else -1) for i in index])
    eq = [b for b in sim.boundaries[1:] if type(b).__name__ == "EquilibriumBoundaryPU"][0]
    bb = [b for b in sim.boundaries[1:] if type(b).__name__ == "BounceBackBoundary"][0]
    save(name, tau=np.float64(tau), boundary_order=np.array(kinds), boundary_direction=np.array(dirs),
         inlet_mask=npy(eq.make_no_collision_mask(list(flow.resolution), ctx)),
         inlet_velocity_pu=np.array([0.3] + [0.0] * (d - 1)),
         block_mask=npy(bb.make_no_collision_mask(list(flow.resolution), ctx)),
         u_char_lu=np.float64(flow.units.characteristic_velocity_lu),
         no_collision_mask=npy(sim.no_collision_mask),
         no_streaming_mask=np.packbits(npy(sim.no_streaming_mask).astype(bool), axis=None),
         no_streaming_mask_shape=np.array(sim.no_streaming_mask.shape),
         resolution=np.array(flow.resolution), **out)


def shear3d_case(dt):
    """cfg5 physics: a build-defined periodic 3-D shear layer run
    through the reference Simulation, so that step parity is pinned even though
    the reference has no such flow.  The initial field is built here with numpy."""
    n = 16
    ctx = lt.Context(device="cpu", dtype=DT[dt], use_native=False)

    class Shear3D(_Dummy):
        def make_units(self, reynolds_number, mach_number, resolution):
            return lt.UnitConversion(reynolds_number, mach_number,
                                     characteristic_length_lu=resolution[0],
                                     characteristic_length_pu=1, characteristic_velocity_pu=1)

        def initial_pu(self):
            ax = np.arange(n) / n
            x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
            ux = np.where(y <= 0.5, np.tanh(80 * (y - 0.25)), np.tanh(80 * (0.75 - y)))
            uy = 0.05 * np.sin(2 * np.pi * (x + 0.25))
            return np.zeros((1, n, n, n)), np.stack([ux, uy, np.zeros_like(ux)])

        def initialize(self):
            lt.Flow.initialize(self)
    fl = quiet(Shear3D, ctx, n, lt.D3Q19(), 1000, 0.1)
    tau = fl.units.relaxation_parameter_lu
    out = run_series(fl, lt.BGKCollision(tau), {5, 20}, 5, periodic=True)
    save(f"shear3d_d3q19_bgk_{dt}", tau=np.float64(tau), **out)


# --------------------------------------------------------------------------- #
if __name__ == "__main__":
    # cfg1 (examples/00_simplest_TGV.py): 128^2 fp64; keep only energies + a 100-step f
    tgv_case("tgv2d_d2q9_bgk_128_f64", 128, lt.D2Q9(), 100, 0.05, "f64", "bgk", {100}, 100)
    tgv_case("tgv2d_d2q9_bgk_32_f64", 32, lt.D2Q9(), 100, 0.05, "f64", "bgk", {10, 100}, 10)
    tgv_case("tgv2d_d2q9_bgk_32_f32", 32, lt.D2Q9(), 100, 0.05, "f32", "bgk", {10, 100}, 10)
    tgv_case("tgv2d_d2q9_kbc_32_f64", 32, lt.D2Q9(), 1000, 0.05, "f64", "kbc", {10, 50}, 10)
    tgv_case("tgv3d_d3q19_bgk_16_f64", 16, lt.D3Q19(), 1600, 0.1, "f64", "bgk", {10, 100}, 10)
    tgv_case("tgv3d_d3q19_bgk_16_f32", 16, lt.D3Q19(), 1600, 0.1, "f32", "bgk", {10, 100}, 10)
    tgv_case("tgv3d_d3q19_bgk_ragged_f64", [12, 10, 20], lt.D3Q19(), 400, 0.1, "f64", "bgk", {7}, 7)
    tgv_case("tgv3d_d3q27_bgk_16_f64", 16, lt.D3Q27(), 1600, 0.1, "f64", "bgk", {10}, 10)
    tgv_case("tgv3d_d3q27_kbc_16_f64", 16, lt.D3Q27(), 1600, 0.1, "f64", "kbc", {10, 50}, 10)
    tgv_case("tgv3d_d3q27_kbc_16_f32", 16, lt.D3Q27(), 1600, 0.1, "f32", "kbc", {10}, 10)
    # grids the kernels with two lattice updates per launch take (last extent % 64 in fp32 / % 32 in fp64,
    # middle extent % 8, D3Q27: % 4): direct vectors for lbm2_kernel and the two-step slab driver
    tgv_case("tgv3d_d3q19_bgk_8x16x64_f32", [8, 16, 64], lt.D3Q19(), 400, 0.1, "f32", "bgk", {1, 2, 3, 10}, 5)
    tgv_case("tgv3d_d3q19_bgk_8x8x32_f64", [8, 8, 32], lt.D3Q19(), 400, 0.1, "f64", "bgk", {1, 2, 3, 10}, 5)
    tgv_case("tgv3d_d3q27_bgk_4x8x64_f32", [4, 8, 64], lt.D3Q27(), 400, 0.1, "f32", "bgk", {2, 3, 10}, 5)
    tgv_case("tgv3d_d3q15_bgk_8x8x64_f32", [8, 8, 64], lt.D3Q15(), 400, 0.1, "f32", "bgk", {2, 3, 10}, 5)
    # the slab layout has x contiguous: a grid the two-step slab driver takes (x % 64, y % 8)
    tgv_case("tgv3d_d3q19_bgk_64x8x12_f32", [64, 8, 12], lt.D3Q19(), 400, 0.1, "f32", "bgk", {2, 9, 10}, 5)
    # anchors of SURVEY.md 8(c): TGV3D D3Q19 32^3, energies only + f after 10 steps (fp32)
    tgv_case("tgv3d_d3q19_bgk_32_f32", 32, lt.D3Q19(), 1600, 0.1, "f32", "bgk", {10}, 10)
    obstacle_case("obstacle2d_d2q9_bgk_f64", [32, 20], lt.D2Q9(), "f64", "bgk", {1, 2, 10},
                  4.0, (1.0, 1.25), 0.4)
    # extents that are multiples of 8: the many-steps-per-launch kernel with masks
    obstacle_case("obstacle2d_d2q9_bgk_40x24_f64", [40, 24], lt.D2Q9(), "f64", "bgk", {1, 2, 9, 20},
                  4.0, (1.0, 1.2), 0.4)
    # contiguous extent % 64: the two-step kernel with masks in two dimensions
    obstacle_case("obstacle2d_d2q9_bgk_24x64_f64", [24, 64], lt.D2Q9(), "f64", "bgk", {1, 2, 3, 8},
                  4.0, (1.0, 5.0), 0.7)
    obstacle_case("obstacle2d_d2q9_bgk_40x24_f32", [40, 24], lt.D2Q9(), "f32", "bgk", {1, 2, 9, 20},
                  4.0, (1.0, 1.2), 0.4)
    obstacle_case("obstacle3d_d3q27_kbc_f64", [20, 12, 12], lt.D3Q27(), "f64", "kbc", {1, 2, 8},
                  4.0, (1.0, 1.2, 1.2), 0.5)
    obstacle_case("obstacle3d_d3q27_kbc_f32", [20, 12, 12], lt.D3Q27(), "f32", "kbc", {2, 8},
                  4.0, (1.0, 1.2, 1.2), 0.5)
    obstacle_case("obstacle3d_d3q19_bgk_f64", [16, 12, 8], lt.D3Q19(), "f64", "bgk", {2, 8},
                  4.0, (1.0, 1.5, 1.0), 0.5)
    # obstacle flows on tile-compatible grids (two-step kernel with masks)
    obstacle_case("obstacle3d_d3q19_bgk_12x16x64_f32", [12, 16, 64], lt.D3Q19(), "f32", "bgk", {1, 2, 3, 8},
                  4.0, (1.3, 2.0, 9.0), 0.8)
    obstacle_case("obstacle3d_d3q27_bgk_10x8x64_f32", [10, 8, 64], lt.D3Q27(), "f32", "bgk", {1, 2, 3, 8},
                  4.0, (1.3, 1.6, 12.0), 0.7)
    obstacle_case("obstacle3d_d3q19_bgk_10x8x32_f64", [10, 8, 32], lt.D3Q19(), "f64", "bgk", {1, 2, 3, 8},
                  4.0, (1.3, 1.6, 6.0), 0.7)
    # ... and on grids the slab layout's tiles take (x contiguous: two-step slab driver with boundaries)
    obstacle_case("obstacle3d_d3q19_bgk_64x8x16_f32", [64, 8, 16], lt.D3Q19(), "f32", "bgk", {1, 2, 3, 8},
                  4.0, (1.3, 0.25, 0.5), 0.2)
    obstacle_case("obstacle3d_d3q27_bgk_64x8x16_f32", [64, 8, 16], lt.D3Q27(), "f32", "bgk", {1, 2, 3, 8},
                  4.0, (1.3, 0.25, 0.5), 0.2)
    two_outlets_case("two_outlets_d2q9_bgk_f64", [12, 10], lt.D2Q9(), "f64")
    two_outlets_case("two_outlets_d3q19_bgk_f64", [8, 7, 6], lt.D3Q19(), "f64")
    two_outlets_case("two_outlets_d3q19_bgk_f32", [8, 7, 6], lt.D3Q19(), "f32")
    two_outlets_case("three_outlets_d2q9_bgk_f64", [12, 10], lt.D2Q9(), "f64", outlets=[[1, 0], [0, 1], [0, -1]])
    two_outlets_case("three_outlets_d3q19_bgk_f32", [8, 7, 6], lt.D3Q19(), "f32", outlets=[[1, 0, 0], [0, 0, 1], [0, 0, -1]])
    two_outlets_case("four_outlets_d3q27_bgk_f64", [6, 8, 7], lt.D3Q27(), "f64",
                     outlets=[[0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
    # round 4: outlets on all THREE axes (their planes meet in corners: an outlet's neighbour depends on two earlier ones)
    two_outlets_case("outlets_on_three_axes_d3q19_bgk_f64", [8, 7, 6], lt.D3Q19(), "f64",
                     outlets=[[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    two_outlets_case("outlets_on_three_axes_d3q27_bgk_f32", [7, 8, 6], lt.D3Q27(), "f32",
                     outlets=[[1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, 1], [0, 0, -1]])
    two_outlets_case("outlets_on_three_axes_d3q15_bgk_f64", [6, 7, 8], lt.D3Q15(), "f64",
                     outlets=[[0, 0, -1], [0, 1, 0], [1, 0, 0], [0, -1, 0]])
    if wanted("native") or wanted("hand"):
        hand_set_cases()
    if wanted("operators"):
        operator_cases()
    if wanted("shear3d"):
        shear3d_case("f64")
        shear3d_case("f32")
