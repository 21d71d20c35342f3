"""Pin the CPU oracle (oracle/lettuce_oracle.py) against vectors produced by the
reference's own PyTorch CPU path (tests/golden, made by oracle/gen_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import golden, unpack_nsm, TORCH_DT
from oracle import lettuce_oracle as orc

# the oracle mirrors the reference op for op; fp64 agreement is at rounding level
TOL = {"f64": dict(rtol=0, atol=2e-14), "f32": dict(rtol=0, atol=2e-6)}


def close(a, b, dt, scale=1.0):
    tol = TOL[dt]
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=0, atol=tol["atol"] * scale)


TGV = [
    ("tgv2d_d2q9_bgk_32_f64", "D2Q9", "bgk", "f64", (10, 100)),
    ("tgv2d_d2q9_bgk_32_f32", "D2Q9", "bgk", "f32", (10, 100)),
    ("tgv2d_d2q9_kbc_32_f64", "D2Q9", "kbc", "f64", (10, 50)),
    ("tgv3d_d3q19_bgk_16_f64", "D3Q19", "bgk", "f64", (10, 100)),
    ("tgv3d_d3q19_bgk_16_f32", "D3Q19", "bgk", "f32", (10, 100)),
    ("tgv3d_d3q19_bgk_ragged_f64", "D3Q19", "bgk", "f64", (7,)),
    ("tgv3d_d3q27_bgk_16_f64", "D3Q27", "bgk", "f64", (10,)),
    ("tgv3d_d3q27_kbc_16_f64", "D3Q27", "kbc", "f64", (10, 50)),
    ("tgv3d_d3q27_kbc_16_f32", "D3Q27", "kbc", "f32", (10,)),
    # grids the kernels with two lattice updates per launch take
    ("tgv3d_d3q19_bgk_8x16x64_f32", "D3Q19", "bgk", "f32", (1, 2, 3, 10)),
    ("tgv3d_d3q19_bgk_8x8x32_f64", "D3Q19", "bgk", "f64", (1, 2, 3, 10)),
    ("tgv3d_d3q27_bgk_4x8x64_f32", "D3Q27", "bgk", "f32", (2, 3, 10)),
    ("tgv3d_d3q15_bgk_8x8x64_f32", "D3Q15", "bgk", "f32", (2, 3, 10)),
    ("tgv3d_d3q19_bgk_64x8x12_f32", "D3Q19", "bgk", "f32", (2, 9, 10)),
]


@pytest.mark.parametrize("name,lat,coll,dt,snaps", TGV, ids=[t[0] for t in TGV])
def test_tgv_initial_condition_and_steps(name, lat, coll, dt, snaps):
    g = golden(name)
    res = [int(r) for r in g["resolution"]]
    sim = orc.taylor_green(res, float(g["reynolds"]), float(g["mach"]), lat, TORCH_DT[dt], coll)
    assert sim.tau == pytest.approx(float(g["tau"]), rel=1e-15)
    close(sim.f.numpy(), g["f0"], dt)                      # F1: init incl. f_neq
    energies = {0: float(orc.kinetic_energy_pu(sim.f, sim.lat, sim.units))}
    for i in range(1, max(snaps) + 1):
        sim.step()
        if i in snaps:
            close(sim.f.numpy(), g[f"f{i}"], dt, scale=max(1.0, i / 10))
        if i in g["energy_steps"]:
            energies[i] = float(orc.kinetic_energy_pu(sim.f, sim.lat, sim.units))
    ref = dict(zip(g["energy_steps"].tolist(), g["energy_pu"].tolist()))
    for i, e in energies.items():
        assert e == pytest.approx(ref[i], rel=1e-12 if dt == "f64" else 2e-6)


@pytest.mark.parametrize("name,lat,coll,dt,snaps", TGV, ids=[t[0] for t in TGV])
def test_enstrophy_and_mass_observables(name, lat, coll, dt, snaps):
    """The reference's Enstrophy and Mass observables at step 0 and at the stored snapshots."""
    g = golden(name)
    steps = g["energy_steps"].tolist()
    lattice = orc.LATTICES[lat]
    res = [int(r) for r in g["resolution"]]
    units = orc.tgv_units(res, float(g["reynolds"]), float(g["mach"]))
    for i in [0] + [n for n in snaps if n in steps]:
        f = torch.as_tensor(g[f"f{i}"])
        at = steps.index(i)
        assert float(orc.enstrophy_pu(f, lattice, units)) == pytest.approx(float(g["enstrophy_pu"][at]),
                                                                           rel=1e-12 if dt == "f64" else 1e-5)
        assert float(orc.mass_observable(f)) == pytest.approx(float(g["mass_observable"][at]),
                                                              rel=1e-13 if dt == "f64" else 1e-6)


def test_cfg1_anchor_energies():
    """SURVEY.md 8(c) anchors for examples/00_simplest_TGV.py (128^2 fp64)."""
    g = golden("tgv2d_d2q9_bgk_128_f64")
    assert float(g["tau"]) == pytest.approx(0.610851251684408, rel=1e-14)
    assert g["energy_pu"][0] == pytest.approx(9.86960440108935, rel=1e-13)
    assert g["energy_pu"][1] == pytest.approx(9.52452318761345, rel=1e-13)
    sim = orc.taylor_green([128, 128], 100, 0.05, "D2Q9", torch.float64)
    close(sim.f.numpy(), g["f0"], "f64")
    sim.step(100)
    close(sim.f.numpy(), g["f100"], "f64", scale=10)
    assert float(orc.kinetic_energy_pu(sim.f, sim.lat, sim.units)) == pytest.approx(
        float(g["energy_pu"][1]), rel=1e-12)


OBST = [("obstacle2d_d2q9_bgk_f64", "D2Q9", "bgk", "f64", (1, 2, 10)),
        ("obstacle2d_d2q9_bgk_40x24_f64", "D2Q9", "bgk", "f64", (1, 2, 9, 20)),
        ("obstacle2d_d2q9_bgk_24x64_f64", "D2Q9", "bgk", "f64", (1, 2, 3, 8)),
        ("obstacle2d_d2q9_bgk_40x24_f32", "D2Q9", "bgk", "f32", (1, 2, 9, 20)),
        ("obstacle3d_d3q27_kbc_f64", "D3Q27", "kbc", "f64", (1, 2, 8)),
        ("obstacle3d_d3q27_kbc_f32", "D3Q27", "kbc", "f32", (2, 8)),
        ("obstacle3d_d3q19_bgk_f64", "D3Q19", "bgk", "f64", (2, 8)),
        # the grids the two-step kernels' tiles take (reference layout: z % 64; slab layout: x % 64)
        ("obstacle3d_d3q19_bgk_12x16x64_f32", "D3Q19", "bgk", "f32", (1, 2, 3, 8)),
        ("obstacle3d_d3q27_bgk_10x8x64_f32", "D3Q27", "bgk", "f32", (1, 2, 3, 8)),
        ("obstacle3d_d3q19_bgk_10x8x32_f64", "D3Q19", "bgk", "f64", (1, 2, 3, 8)),
        ("obstacle3d_d3q19_bgk_64x8x16_f32", "D3Q19", "bgk", "f32", (1, 2, 3, 8)),
        ("obstacle3d_d3q27_bgk_64x8x16_f32", "D3Q27", "bgk", "f32", (1, 2, 3, 8))]


def obstacle_oracle(g, lat_name, coll, dt):
    lat = orc.LATTICES[lat_name]
    res = [int(r) for r in g["resolution"]]
    units = orc.Units(100, 0.1, characteristic_length_lu=float(g["char_length_lu"]),
                      characteristic_length_pu=1, characteristic_velocity_pu=1)
    dtype = TORCH_DT[dt]
    x = torch.meshgrid(*[units.length_to_pu(torch.arange(n)) for n in res], indexing="ij")[0]
    direction = [1] + [0] * (lat.d - 1)
    bnds = [
        orc.OracleBoundary("equilibrium_pu", mask=torch.abs(x) < 1e-6,
                           velocity_pu=torch.tensor([1.0] + [0.0] * (lat.d - 1), dtype=dtype),
                           pressure_pu=torch.tensor(0, dtype=dtype)),
        orc.OracleBoundary("abb_outlet", direction=direction),
        orc.OracleBoundary("bounce_back", mask=torch.tensor(g["obstacle_mask"])),
    ]
    return orc.OracleSimulation(lat, torch.tensor(g["f0"]), coll, float(g["tau"]), units, bnds)


@pytest.mark.parametrize("name,lat,coll,dt,snaps", OBST, ids=[t[0] for t in OBST])
def test_obstacle_masks_and_steps(name, lat, coll, dt, snaps):
    g = golden(name)
    assert list(g["boundary_order"]) == ["AntiBounceBackOutlet", "BounceBackBoundary",
                                         "EquilibriumBoundaryPU"]
    sim = obstacle_oracle(g, lat, coll, dt)
    assert sim.tau == pytest.approx(orc.Units(100, 0.1, float(g["char_length_lu"])).tau, rel=1e-15)
    np.testing.assert_array_equal(sim.no_collision_mask.numpy(), g["no_collision_mask"])
    np.testing.assert_array_equal(sim.no_streaming_mask.numpy(), unpack_nsm(g))
    for i in range(1, max(snaps) + 1):
        sim.step()
        if i in snaps:
            close(sim.f.numpy(), g[f"f{i}"], dt, scale=2)


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_shear3d_steps(dt):
    g = golden(f"shear3d_d3q19_bgk_{dt}")
    sim = orc.OracleSimulation(orc.LATTICES["D3Q19"], torch.tensor(g["f0"]), "bgk", float(g["tau"]))
    sim.step(5)
    close(sim.f.numpy(), g["f5"], dt)
    sim.step(15)
    close(sim.f.numpy(), g["f20"], dt, scale=4)


def test_hand_set_streaming_bgk_bounce_back():
    lat = orc.LATTICES["D2Q9"]
    g = golden("native_streaming_d2q9_f32")
    sim = orc.OracleSimulation(lat, torch.tensor(g["f0"]), "none", 1.0)
    np.testing.assert_array_equal(sim.step().numpy(), g["f1"])      # pure data movement
    g = golden("native_bgk_d2q9_f32")
    sim = orc.OracleSimulation(lat, torch.tensor(g["f0"]), "bgk", float(g["tau"]))
    close(sim.step().numpy(), g["f1"], "f32")
    g = golden("native_bounce_back_d2q9_f32")
    b = orc.OracleBoundary("bounce_back", mask=torch.tensor(g["no_collision_mask"]).bool())
    sim = orc.OracleSimulation(lat, torch.tensor(g["f0"]), "none", 1.0, boundaries=[b])
    np.testing.assert_array_equal(sim.step().numpy(), g["f1"])
    np.testing.assert_array_equal(sim.step().numpy(), g["f2"])


def test_hand_set_equilibrium_pu_and_no_streaming_mask():
    lat = orc.LATTICES["D2Q9"]
    g = golden("native_equilibrium_pu_d2q9_f64")
    units = orc.tgv_units([16, 16], 1, 0.1)
    assert units.u_char_lu == pytest.approx(float(g["u_char_lu"]), rel=1e-15)
    mask = torch.zeros(16, 16, dtype=torch.bool)
    mask[:, 1] = True
    b = orc.OracleBoundary("equilibrium_pu", mask=mask,
                           velocity_pu=torch.ones(2, 16, 16, dtype=torch.float64),
                           pressure_pu=torch.ones(16, 16, dtype=torch.float64),
                           no_streaming_mask=torch.ones(9, 16, 16, dtype=torch.bool))
    sim = orc.OracleSimulation(lat, torch.tensor(g["f0"]), "none", 1.0, units, [b])
    close(sim.step().numpy(), g["f1"], "f64")
    # a grid-shaped (not [q,*res]) all-zero mask assigned after construction
    g = golden("native_no_streaming_mask_d2q9_f32")
    sim = orc.OracleSimulation(lat, torch.tensor(g["f0"]), "none", 1.0)
    sim.no_streaming_mask = torch.zeros(16, 16, dtype=torch.bool)
    close(sim.step(64).numpy(), g["f64"], "f32")


OPS = [(s, d) for s in ("d2q9", "d3q19", "d3q27") for d in ("f64", "f32")]


@pytest.mark.parametrize("sname,dt", OPS, ids=[f"{s}-{d}" for s, d in OPS])
def test_whole_field_operators(sname, dt):
    g = golden(f"operators_{sname}_{dt}")
    lat = orc.LATTICES[sname.upper()]
    f = torch.tensor(g["f"])
    e, w = orc.lattice_tensors(lat, f.dtype)
    close(orc.density(f).numpy(), g["rho"], dt)
    close(orc.momentum(f, e).numpy(), g["j"], dt)
    close(orc.velocity(f, e).numpy(), g["u"], dt)
    close(orc.incompressible_energy(f, e).numpy(), g["energy"], dt)
    close(orc.quadratic_equilibrium(orc.density(f), orc.velocity(f, e), e, w).numpy(), g["feq"], dt)
    close(orc.bgk(f, float(g["tau"]), e, w).numpy(), g["bgk"], dt)
    np.testing.assert_array_equal(orc.bounce_back(f, lat).numpy(), g["bounce_back"])
    if "kbc" in g:
        close(orc.kbc(f, float(g["tau_units"]), e, w).numpy(), g["kbc"], dt, scale=4)
    for axis in range(lat.d):
        for sign, tag in ((1, "p"), (-1, "m")):
            direction = [0] * lat.d
            direction[axis] = sign
            b = orc.OracleBoundary("abb_outlet", direction=direction)
            key = f"abb_{'xyz'[axis]}{tag}"
            close(orc.abb_outlet_inplace(f.clone(), b, lat, e, w).numpy(), g[key], dt)
            ncm, nsm = orc.abb_masks(f.shape, b, lat)
            np.testing.assert_array_equal(ncm.numpy(), g[key + "_ncm"])
            np.testing.assert_array_equal(nsm.numpy(), g[key + "_nsm"])


TWO_OUTLETS = [("two_outlets_d2q9_bgk_f64", "D2Q9", "f64"), ("two_outlets_d3q19_bgk_f64", "D3Q19", "f64"),
               ("two_outlets_d3q19_bgk_f32", "D3Q19", "f32"),
               # round 3: any number of outlets on at most two axes (+x, +y, -y; +x, +z, -z; +-y, +-z)
               ("three_outlets_d2q9_bgk_f64", "D2Q9", "f64"), ("three_outlets_d3q19_bgk_f32", "D3Q19", "f32"),
               ("four_outlets_d3q27_bgk_f64", "D3Q27", "f64"),
               # round 4: outlets on all three axes (+x +y +z; +x -y +y +z -z; -z +y +x -y), planes meeting in corners
               ("outlets_on_three_axes_d3q19_bgk_f64", "D3Q19", "f64"),
               ("outlets_on_three_axes_d3q27_bgk_f32", "D3Q27", "f32"),
               ("outlets_on_three_axes_d3q15_bgk_f64", "D3Q15", "f64")]


def two_outlets_boundaries(g, dtype):
    """the boundaries of a two_outlets_* fixture in the order the reference applied them"""
    bnds = []
    for kind, direction in zip(g["boundary_order"].tolist(), g["boundary_direction"].tolist()):
        if kind == "AntiBounceBackOutlet":
            bnds.append(orc.OracleBoundary("abb_outlet", direction=direction))
        elif kind == "BounceBackBoundary":
            bnds.append(orc.OracleBoundary("bounce_back", mask=torch.tensor(g["block_mask"])))
        else:
            bnds.append(orc.OracleBoundary("equilibrium_pu", mask=torch.tensor(g["inlet_mask"]),
                                           velocity_pu=torch.tensor(g["inlet_velocity_pu"], dtype=dtype),
                                           pressure_pu=torch.tensor(0, dtype=dtype)))
    return bnds


@pytest.mark.parametrize("name,lat,dt", TWO_OUTLETS, ids=[t[0] for t in TWO_OUTLETS])
def test_two_anti_bounce_back_outlets(name, lat, dt):
    """The reference takes any list of boundaries (lettuce/_simulation.py:57-86): two outlets whose planes meet
    in an edge, both orders (the fixtures hold whichever order the reference's sort by address produced)."""
    g = golden(name)
    L = orc.LATTICES[lat]
    dtype = TORCH_DT[dt]
    units = orc.tgv_units([int(r) for r in g["resolution"]], 100, 0.05)
    assert units.u_char_lu == pytest.approx(float(g["u_char_lu"]), rel=1e-15)
    bnds = two_outlets_boundaries(g, dtype)
    sim = orc.OracleSimulation(L, torch.tensor(g["f0"]), "bgk", float(g["tau"]), units, bnds)
    sim.boundaries = bnds                                   # the stored order, not the oracle's own sort
    sim.no_collision_mask = torch.tensor(g["no_collision_mask"])
    sim.no_streaming_mask = torch.tensor(unpack_nsm(g))
    for i in range(1, 7):
        sim.step()
        if i in (1, 2, 6):
            close(sim.f.numpy(), g[f"f{i}"], dt, scale=2)
