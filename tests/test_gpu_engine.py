"""GPU parity tests proper: the HIP engine, called through its C ABI
(lettuce_amd/_native.py -> liblettuce_hip.so), against the CPU oracle and the golden
vectors of the reference CPU path.

Tolerances: fp64 max|df| <= 1e-12 over <= 100 steps; fp32
max|df| <= 1e-5 * max|f| after 10 steps.  Pure data movement (streaming, bounce-back,
masks) must be bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import golden, unpack_nsm, TORCH_DT
from oracle import lettuce_oracle as orc

pytestmark = pytest.mark.gpu

ATOL = {"f64": 1e-12, "f32": 1e-5}


def _values(t):
    """the plain tuple of a parameter set, also when it carries marks (pytest.param)"""
    return t.values if hasattr(t, "values") and hasattr(t, "marks") else t


def dev(x, dtype=None):
    t = torch.as_tensor(x)
    return t.to(device="cuda", dtype=dtype or t.dtype).contiguous()


def plan_for(lat, dtype, coll, res, boundaries=()):
    from lettuce_amd._native import Plan
    return Plan(lat, dtype, coll, res, boundaries)


def run_engine(plan, f0, tau, n):
    a = dev(f0)
    b = torch.empty_like(a)
    out, _ = plan.run(a, b, tau, n)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def assert_close(got, want, dt, scale=1.0):
    tol = ATOL[dt] * max(1.0, float(np.abs(want).max())) * scale
    np.testing.assert_allclose(got, want, rtol=0, atol=tol)


# --------------------------------------------------------------------------- periodic flows
TGV = [
    ("tgv2d_d2q9_bgk_32_f64", "D2Q9", "bgk", "f64", (10, 100)),
    ("tgv2d_d2q9_bgk_32_f32", "D2Q9", "bgk", "f32", (10,)),
    ("tgv3d_d3q19_bgk_16_f64", "D3Q19", "bgk", "f64", (10, 100)),
    ("tgv3d_d3q19_bgk_16_f32", "D3Q19", "bgk", "f32", (10,)),
    ("tgv3d_d3q19_bgk_ragged_f64", "D3Q19", "bgk", "f64", (7,)),
    ("tgv3d_d3q27_bgk_16_f64", "D3Q27", "bgk", "f64", (10,)),
    ("tgv3d_d3q27_kbc_16_f64", "D3Q27", "kbc", "f64", (10, 50)),
    ("tgv3d_d3q27_kbc_16_f32", "D3Q27", "kbc", "f32", (10,)),
    ("tgv3d_d3q19_bgk_32_f32", "D3Q19", "bgk", "f32", (10,)),
    ("shear3d_d3q19_bgk_f64", "D3Q19", "bgk", "f64", (5, 20)),
    ("shear3d_d3q19_bgk_f32", "D3Q19", "bgk", "f32", (5,)),
]


@pytest.mark.parametrize("name,lat,coll,dt,snaps", TGV, ids=[t[0] for t in TGV])
def test_periodic_steps_match_reference_vectors(name, lat, coll, dt, snaps):
    g = golden(name)
    f0 = g["f0"]
    plan = plan_for(lat, TORCH_DT[dt], coll, f0.shape[1:])
    for n in snaps:
        got = run_engine(plan, f0, float(g["tau"]), n)
        assert_close(got, g[f"f{n}"], dt, scale=max(1.0, n / 10) if dt == "f32" else 1.0)


def test_kbc_d2q9_one_step_on_perturbed_state():
    """KBC on a smooth 2-D TGV is ill-conditioned in the reference itself (gamma is a 0/0 of
    rounding noise), so D2Q9-KBC is pinned on states with a resolved higher-order part."""
    for dt in ("f64", "f32"):
        g = golden(f"operators_d2q9_{dt}")
        f = g["f"]
        plan = plan_for("D2Q9", TORCH_DT[dt], "kbc", f.shape[1:])
        out = plan.collide(dev(f), torch.empty_like(dev(f)), float(g["tau_units"]))
        assert_close(out.cpu().numpy(), g["kbc"], dt)


@pytest.mark.parametrize("lat,res", [("D2Q9", [7, 5]), ("D3Q19", [6, 5, 7]), ("D3Q27", [3, 4, 5]),
                                     ("D3Q19", [4, 4, 8]), ("D2Q9", [1, 8]), ("D3Q27", [2, 1, 4]),
                                     ("D1Q3", [37]), ("D1Q3", [2]), ("D3Q15", [5, 6, 7]), ("D3Q15", [8, 8, 8])])
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_ragged_and_tiny_grids_against_oracle(lat, res, dt):
    """Sizes the 16-byte vector path cannot take, and sizes where every node wraps."""
    torch.manual_seed(7)
    L = orc.LATTICES[lat]
    w = torch.tensor(L.w, dtype=torch.float64).reshape([-1] + [1] * len(res))
    f0 = (w * (1 + 0.1 * torch.rand([L.q] + res, dtype=torch.float64))).to(TORCH_DT[dt])
    for coll in ("none", "bgk"):
        sim = orc.OracleSimulation(L, f0.clone(), coll, 0.8)
        sim.step(3)
        plan = plan_for(lat, TORCH_DT[dt], coll, res)
        got = run_engine(plan, f0.numpy(), 0.8, 3)
        if coll == "none":
            np.testing.assert_array_equal(got, sim.f.numpy())
        else:
            assert_close(got, sim.f.numpy(), dt)


@pytest.mark.parametrize("lat", ["D1Q3", "D2Q9", "D3Q15", "D3Q19", "D3Q27"])
def test_streaming_is_exact_and_periodic(lat):
    """Tagged populations: S^n with n = grid size returns to the start; one step equals
    torch.roll by +e_q (tests/native/test_native_streaming.py)."""
    L = orc.LATTICES[lat]
    res = [8] * L.d
    f0 = torch.arange(L.q * int(np.prod(res)), dtype=torch.float32).reshape([L.q] + res)
    plan = plan_for(lat, torch.float32, "none", res)
    sim = orc.OracleSimulation(L, f0.clone(), "none", 1.0)
    np.testing.assert_array_equal(run_engine(plan, f0.numpy(), 1.0, 1), sim.step().numpy())
    np.testing.assert_array_equal(run_engine(plan, f0.numpy(), 1.0, 8), f0.numpy())


def test_hand_set_cases_of_the_reference_native_tests():
    g = golden("native_streaming_d2q9_f32")
    plan = plan_for("D2Q9", torch.float32, "none", [16, 16])
    np.testing.assert_array_equal(run_engine(plan, g["f0"], 1.0, 1), g["f1"])
    g = golden("native_bgk_d2q9_f32")
    plan = plan_for("D2Q9", torch.float32, "bgk", [16, 16])
    got = run_engine(plan, g["f0"], float(g["tau"]), 1)
    np.testing.assert_allclose(got, g["f1"], rtol=1e-6, atol=0)   # pytest.approx default of the reference test
    g = golden("native_bounce_back_d2q9_f32")
    plan = plan_for("D2Q9", torch.float32, "none", [16, 16], [{"kind": "bounce_back"}])
    plan.set_masks(dev(g["no_collision_mask"]), None)
    np.testing.assert_array_equal(run_engine(plan, g["f0"], 1.0, 1), g["f1"])
    np.testing.assert_array_equal(run_engine(plan, g["f0"], 1.0, 2), g["f2"])


def test_equilibrium_boundary_with_field_and_full_no_streaming_mask():
    g = golden("native_equilibrium_pu_d2q9_f64")
    L = orc.LATTICES["D2Q9"]
    units = orc.tgv_units([16, 16], 1, 0.1)
    e, w = orc.lattice_tensors(L, torch.float64)
    b = orc.OracleBoundary("equilibrium_pu", velocity_pu=torch.ones(2, 16, 16, dtype=torch.float64),
                           pressure_pu=torch.ones(16, 16, dtype=torch.float64))
    field = orc.equilibrium_pu(torch.tensor(g["f0"]), b, units, e, w)   # host-side constant
    ncm = torch.zeros(16, 16, dtype=torch.uint8)
    ncm[:, 1] = 1
    plan = plan_for("D2Q9", torch.float64, "none", [16, 16], [{"kind": "equilibrium", "field": field}])
    plan.set_masks(dev(ncm), dev(torch.ones(9, 16, 16, dtype=torch.uint8)))
    got = run_engine(plan, g["f0"], 1.0, 1)
    assert_close(got, g["f1"], "f64")


# --------------------------------------------------------------------------- obstacle flows
def obstacle_plan(g, lat, coll, dt):
    L = orc.LATTICES[lat]
    units = orc.Units(100, 0.1, characteristic_length_lu=float(g["char_length_lu"]))
    e, w = orc.lattice_tensors(L, TORCH_DT[dt])
    u_in = units.velocity_to_lu(torch.tensor([1.0] + [0.0] * (L.d - 1), dtype=TORCH_DT[dt]))
    rho_in = units.pressure_pu_to_density_lu(torch.tensor(0, dtype=TORCH_DT[dt]))
    feq_in = orc.quadratic_equilibrium(rho_in, u_in, e, w)
    bnds = [{"kind": "abb_outlet", "axis": 0, "side": 1}, {"kind": "bounce_back"},
            {"kind": "equilibrium", "feq": feq_in.double().tolist()}]
    plan = plan_for(lat, TORCH_DT[dt], coll, g["f0"].shape[1:], bnds)
    plan.set_masks(dev(g["no_collision_mask"]), dev(unpack_nsm(g)))
    return plan


OBST = [("obstacle2d_d2q9_bgk_f64", "D2Q9", "bgk", "f64", (1, 2, 10)),
        ("obstacle3d_d3q27_kbc_f64", "D3Q27", "kbc", "f64", (1, 2, 8)),
        ("obstacle3d_d3q27_kbc_f32", "D3Q27", "kbc", "f32", (2, 8)),
        ("obstacle3d_d3q19_bgk_f64", "D3Q19", "bgk", "f64", (2, 8))]


@pytest.mark.parametrize("name,lat,coll,dt,snaps", OBST, ids=[t[0] for t in OBST])
def test_obstacle_inlet_outlet_bounce_back(name, lat, coll, dt, snaps):
    g = golden(name)
    plan = obstacle_plan(g, lat, coll, dt)
    for n in snaps:
        got = run_engine(plan, g["f0"], float(g["tau"]), n)
        assert_close(got, g[f"f{n}"], dt, scale=10 if dt == "f64" else 1)


@pytest.mark.parametrize("lat", ["D2Q9", "D3Q19", "D3Q27"])
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_abb_outlet_every_direction(lat, dt):
    """collide-only pass with NoCollision == one application of the boundary
    (tests/boundary/test_antibounceback_outlet_bc.py)."""
    g = golden(f"operators_{lat.lower()}_{dt}")
    L = orc.LATTICES[lat]
    f = g["f"]
    for axis in range(L.d):
        for side, tag in ((1, "p"), (-1, "m")):
            key = f"abb_{'xyz'[axis]}{tag}"
            plan = plan_for(lat, TORCH_DT[dt], "none", f.shape[1:],
                            [{"kind": "abb_outlet", "axis": axis, "side": side}])
            plan.set_masks(dev(g[key + "_ncm"].astype(np.uint8)), dev(g[key + "_nsm"].astype(np.uint8)))
            out = plan.collide(dev(f), torch.empty_like(dev(f)), 1.0)
            assert_close(out.cpu().numpy(), g[key], dt)


# --------------------------------------------------------------------------- operators
@pytest.mark.parametrize("lat", ["D2Q9", "D3Q19", "D3Q27"])
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_whole_field_operators(lat, dt):
    g = golden(f"operators_{lat.lower()}_{dt}")
    L = orc.LATTICES[lat]
    f = dev(g["f"])
    plan = plan_for(lat, TORCH_DT[dt], "bgk", g["f"].shape[1:])
    rho, u = plan.macroscopic(f)
    assert_close(rho.cpu().numpy()[None], g["rho"], dt)
    assert_close(u.cpu().numpy(), g["u"], dt)
    feq = plan.equilibrium(dev(g["rho"]), dev(g["u"]))
    assert_close(feq.cpu().numpy(), g["feq"], dt)
    out = plan.collide(f, torch.empty_like(f), float(g["tau"]))
    assert_close(out.cpu().numpy(), g["bgk"], dt)
    ke = float(plan.kinetic_energy_lu(f).cpu())
    assert ke == pytest.approx(float(g["energy"].astype(np.float64).sum()), rel=1e-12 if dt == "f64" else 2e-6)
    umax = float(plan.max_velocity_lu(f).cpu())
    assert umax == pytest.approx(float(np.sqrt((g["u"].astype(np.float64) ** 2).sum(axis=0)).max()), rel=1e-12 if dt == "f64" else 1e-6)
    mass = float(plan.mass(f).cpu())
    assert mass == pytest.approx(float(g["f"].astype(np.float64).sum()), rel=1e-12 if dt == "f64" else 1e-6)
    bb = plan_for(lat, TORCH_DT[dt], "none", g["f"].shape[1:], [{"kind": "bounce_back"}])
    bb.set_masks(dev(np.ones(g["f"].shape[1:], dtype=np.uint8)), None)
    out = bb.collide(f, torch.empty_like(f), 1.0)
    np.testing.assert_array_equal(out.cpu().numpy(), g["bounce_back"])


def test_fused_equals_collide_then_stream_and_ab_variants_agree():
    """fused == stream then collide; the 16-byte A/B variants of the hot kernel (three ways of
    resolving the shift along the contiguous axis, cached / nontemporal) give the same
    populations as the default one-node-per-thread kernel."""
    for name, lat in (("tgv3d_d3q19_bgk_32_f32", "D3Q19"), ("tgv3d_d3q27_bgk_16_f64", "D3Q27")):
        g = golden(name)
        f = dev(g["f0"])
        tau = float(g["tau"])
        plan = plan_for(lat, f.dtype, "bgk", g["f0"].shape[1:])
        assert plan.kernel_info()["vec"] == 1
        tmp, a, b = torch.empty_like(f), torch.empty_like(f), torch.empty_like(f)
        plan.stream(f, tmp)
        plan.collide(tmp, a, tau)
        plan.stream_collide(f, b, tau)
        # same arithmetic on the same values; only fp-contraction choices may differ between
        # kernel instantiations, so compare at the 1-ulp level instead of bit for bit
        tol = 2e-7 if f.dtype == torch.float32 else 4e-16
        torch.testing.assert_close(b, a, rtol=0, atol=tol)
        for cache in (0, 3):
            plan.set_tuning(cache, False)
            c = torch.empty_like(f)
            plan.stream_collide(f, c, tau)
            assert torch.equal(c, b)
        from conftest import experiments_built
        for shift, cache in ((0, 0), (1, 0), (2, 0), (0, 2), (2, 3)) if experiments_built() else ():
            plan.set_tuning(cache, True)
            plan.set_shift_policy(shift)
            assert plan.kernel_info()["vec"] == 16 // f.element_size()
            c = torch.empty_like(f)
            plan.stream_collide(f, c, tau)
            torch.testing.assert_close(c, b, rtol=0, atol=tol)


OBSERVED = [("tgv2d_d2q9_bgk_32_f64", "D2Q9", "f64", (0, 10, 100)), ("tgv2d_d2q9_bgk_32_f32", "D2Q9", "f32", (0, 10)),
            ("tgv3d_d3q19_bgk_16_f64", "D3Q19", "f64", (0, 10, 100)), ("tgv3d_d3q19_bgk_16_f32", "D3Q19", "f32", (0, 10)),
            ("tgv3d_d3q19_bgk_ragged_f64", "D3Q19", "f64", (0, 7)), ("tgv3d_d3q27_kbc_16_f64", "D3Q27", "f64", (0, 10, 50)),
            ("tgv3d_d3q19_bgk_8x16x64_f32", "D3Q19", "f32", (0, 10)), ("shear3d_d3q19_bgk_f64", "D3Q19", "f64", (0, 5, 20))]


@pytest.mark.parametrize("name,lat,dt,snaps", OBSERVED, ids=[t[0] for t in OBSERVED])
def test_enstrophy_and_mass_device_reductions_match_the_reference(name, lat, dt, snaps):
    """lt_enstrophy (u pass + 6th-order vorticity stencil, fp64 reduction) and lt_mass_interior against the
    values of the reference's Enstrophy / Mass observables on the same populations (row F3)."""
    g = golden(name)
    steps = g["energy_steps"].tolist()
    res = g["f0"].shape[1:]
    plan = plan_for(lat, TORCH_DT[dt], "none", res)
    if name.startswith("tgv"):
        units = orc.tgv_units([int(r) for r in g["resolution"]], float(g["reynolds"]), float(g["mach"]))
        u_scale, dx = units.characteristic_velocity_pu / units.u_char_lu, units.length_to_pu(1.0)
    else:
        u_scale = dx = None
    for i in snaps:
        f = dev(g[f"f{i}"])
        at = steps.index(i)
        mass = float(plan.mass_interior(f))
        # fp32: the reference sums 1e5 .. 1e6 terms in fp32 (torch.sum), the kernel in fp64
        assert mass == pytest.approx(float(g["mass_observable"][at]), rel=1e-13 if dt == "f64" else 1e-5)
        if u_scale is not None:
            ens = float(plan.enstrophy_sum(f, u_scale, 1.0 / dx)) * dx ** len(res)
            assert ens == pytest.approx(float(g["enstrophy_pu"][at]), rel=1e-9 if dt == "f64" else 1e-5)
    # a no-mass mask: flagged nodes are subtracted wherever they are (borders included)
    torch.manual_seed(5)
    mask = (torch.rand(list(res)) < 0.2)
    f = torch.as_tensor(g["f0"])
    want = float(orc.mass_observable(f.double(), mask))
    got = float(plan.mass_interior(dev(g["f0"]), mask.cuda()))
    assert got == pytest.approx(want, rel=1e-12 if dt == "f64" else 1e-5)


def test_energy_decay_series_fp64():
    """TGV kinetic-energy decay vs the reference's series (north-star parity metric, 1e-6 rel)."""
    g = golden("tgv3d_d3q19_bgk_16_f64")
    units = orc.tgv_units([16] * 3, float(g["reynolds"]), float(g["mach"]))
    plan = plan_for("D3Q19", torch.float64, "bgk", [16] * 3)
    a = dev(g["f0"]); b = torch.empty_like(a)
    scale = units.incompressible_energy_to_pu(1.0) * units.length_to_pu(1.0) ** 3
    series = [float(plan.kinetic_energy_lu(a).cpu()) * scale]
    cur, other = a, b
    for _ in range(10):
        cur, other = plan.run(cur, other, float(g["tau"]), 10)
        series.append(float(plan.kinetic_energy_lu(cur).cpu()) * scale)
    np.testing.assert_allclose(series, g["energy_pu"], rtol=1e-9)


# --------------------------------------------------------------------------- full size
def test_full_size_properties_256cubed_fp32():
    """BASELINE cfg2 size (D3Q19 256^3 fp32): size-independent properties -- mass is conserved
    by collide-stream, streaming 256 times is the identity, lt_continue == lt_run."""
    n = 256
    plan = plan_for("D3Q19", torch.float32, "bgk", [n] * 3)
    assert plan.kernel_info()["vec"] == 1
    torch.manual_seed(0)
    w = torch.tensor(orc.LATTICES["D3Q19"].w, dtype=torch.float32, device="cuda").reshape(19, 1, 1, 1)
    f0 = w * (1 + 0.05 * torch.rand(19, n, n, n, device="cuda"))
    m0 = float(plan.mass(f0).cpu())
    a, b = f0.clone(), torch.empty_like(f0)
    r1, fstar = plan.run(a, b, 0.6, 5)
    m1 = float(plan.mass(r1).cpu())
    assert m1 == pytest.approx(m0, rel=1e-6)
    ten_direct = r1.clone()
    # continue 5 more from f* and compare with 10 steps in one go
    r2, _ = plan.run(fstar, r1, 0.6, 5, from_fstar=True)
    a2, b2 = f0.clone(), torch.empty_like(f0)
    r3, _ = plan.run(a2, b2, 0.6, 10)
    assert torch.equal(r2, r3)
    del ten_direct
    none = plan_for("D3Q19", torch.float32, "none", [n] * 3)
    a, b = f0.clone(), torch.empty_like(f0)
    r, _ = none.run(a, b, 1.0, n)
    assert torch.equal(r, f0)


def test_more_than_2_to_the_31_elements_tiled_periodic_field_is_bit_identical():
    """BASELINE cfg3's global size on one GPU (D3Q19 512^3 fp32: 2.55e9 elements per buffer, beyond
    32-bit element offsets).  A field with period 256 in every direction must evolve exactly like
    the 256^3 field it is tiled from: the per-node arithmetic is the same, only the addressing
    differs."""
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2 ** 30:
        pytest.skip("needs 40 GiB of free device memory")
    n, steps = 256, 6
    torch.manual_seed(1)
    w = torch.tensor(orc.LATTICES["D3Q19"].w, dtype=torch.float32, device="cuda").reshape(19, 1, 1, 1)
    f0 = w * (1 + 0.05 * torch.rand(19, n, n, n, device="cuda"))
    small = plan_for("D3Q19", torch.float32, "bgk", [n] * 3)
    ref, _ = small.run(f0.clone(), torch.empty_like(f0), 0.6, steps)
    big = plan_for("D3Q19", torch.float32, "bgk", [2 * n] * 3)
    a = f0.repeat(1, 2, 2, 2).contiguous()
    assert a.numel() > 2 ** 31
    b = torch.empty_like(a)
    out, other = big.run(a, b, 0.6, steps)
    torch.cuda.synchronize()
    del other
    for ix in range(2):
        for iy in range(2):
            for iz in range(2):
                tile = out[:, ix * n:(ix + 1) * n, iy * n:(iy + 1) * n, iz * n:(iz + 1) * n]
                assert torch.equal(tile, ref), (ix, iy, iz)
    m = float(big.mass(out).cpu())
    assert m == pytest.approx(8 * float(small.mass(ref).cpu()), rel=1e-7)


def test_hipgraph_replay_gives_identical_populations():
    """Launch-bound grids replay the fused launches as a captured hipGraph (32 steps per graph);
    the result must be bit-identical to eager launches, also when the graph is reused, when the
    step count is not a multiple of the chunk and when lt_continue carries on."""
    g = golden("tgv2d_d2q9_bgk_32_f64")
    tau = float(g["tau"])
    plan_g = plan_for("D2Q9", torch.float64, "bgk", [32, 32])
    plan_e = plan_for("D2Q9", torch.float64, "bgk", [32, 32])
    plan_g.set_graph_mode(1)
    plan_e.set_graph_mode(0)
    for n in (100, 100, 171, 65):
        a, b = dev(g["f0"]), torch.empty(g["f0"].shape, dtype=torch.float64, device="cuda")
        rg, og = plan_g.run(a, b, tau, n)
        a2, b2 = dev(g["f0"]), torch.empty_like(b)
        re_, oe = plan_e.run(a2, b2, tau, n)
        assert torch.equal(rg, re_) and torch.equal(og, oe)
        rg2, _ = plan_g.run(og, rg, tau, 70, from_fstar=True)
        re2, _ = plan_e.run(oe, re_, tau, 70, from_fstar=True)
        assert torch.equal(rg2, re2)
    got = run_engine(plan_g, g["f0"], tau, 100)
    assert_close(got, g["f100"], "f64")


# --------------------------------------------------------------------------- mask semantics
def _random_state(L, res, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.tensor(L.w, dtype=torch.float64).reshape([-1] + [1] * len(res))
    return (w * (1 + 0.2 * torch.rand([L.q] + res, generator=g, dtype=torch.float64))).to(dtype)


@pytest.mark.parametrize("lat,res", [("D2Q9", [12, 10]), ("D3Q19", [8, 6, 10]), ("D3Q27", [6, 8, 7]),
                                     ("D1Q3", [40]), ("D3Q15", [6, 7, 8])])
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_random_masks_bounce_back_equilibrium_and_sparse_no_streaming(lat, res, dt):
    """Random no_collision_mask with two boundary kinds + random sparse no_streaming bits on fluid
    and boundary nodes, BGK in between: exercises the 1-byte node descriptor, the sparse bit set,
    'highest index wins' and the destination-side no-streaming rule against the oracle."""
    L = orc.LATTICES[lat]
    dtype = TORCH_DT[dt]
    g = torch.Generator().manual_seed(11)
    f0 = _random_state(L, res, dtype, 5)
    e, w = orc.lattice_tensors(L, dtype)
    bb_mask = torch.rand(res, generator=g) < 0.15
    eq_mask = torch.rand(res, generator=g) < 0.10
    nsm = (torch.rand([L.q] + res, generator=g) < 0.05)
    units = orc.Units(10, 0.1)
    vel = torch.tensor([0.3, -0.2, 0.1][:L.d], dtype=dtype)
    b_bb = orc.OracleBoundary("bounce_back", mask=bb_mask, no_streaming_mask=nsm)
    b_eq = orc.OracleBoundary("equilibrium_pu", mask=eq_mask, velocity_pu=vel,
                              pressure_pu=torch.tensor(0.02, dtype=dtype))
    sim = orc.OracleSimulation(L, f0.clone(), "bgk", 0.7, units, [b_bb, b_eq])
    feq = orc.quadratic_equilibrium(units.pressure_pu_to_density_lu(b_eq.pressure_pu),
                                    units.velocity_to_lu(vel), e, w)
    plan = plan_for(lat, dtype, "bgk", res, [{"kind": "bounce_back"},
                                             {"kind": "equilibrium", "feq": feq.double().tolist()}])
    plan.set_masks(dev(sim.no_collision_mask), dev(sim.no_streaming_mask))
    sim.step(4)
    got = run_engine(plan, f0.numpy(), 0.7, 4)
    assert_close(got, sim.f.numpy(), dt, scale=4)


@pytest.mark.parametrize("lat,res,dt,n_b,two_step", [("D3Q19", [8, 6, 10], "f64", 40, False), ("D2Q9", [16, 12], "f32", 127, False),
                                                     ("D3Q19", [4, 8, 64], "f32", 15, True), ("D3Q19", [4, 8, 64], "f32", 16, True)])
def test_many_boundaries_per_flow(lat, res, dt, n_b, two_step):
    """Round 4: LT_MAX_BOUNDARIES 7 -> 127 (what the node byte's seven index bits can name; the reference's uint8 mask
    takes 255, lettuce/_simulation.py:63-86).  n_b alternating bounce-back / equilibrium boundaries with their own
    equilibria on random nodes, 'highest index wins', against the oracle; up to 15 boundaries the two-step kernel with
    boundaries takes the plan (bit-identical to one-step launches), from 16 on it keeps the one-step kernel."""
    L = orc.LATTICES[lat]
    dtype = TORCH_DT[dt]
    g = torch.Generator().manual_seed(23)
    f0 = _random_state(L, res, dtype, 6)
    e, w = orc.lattice_tensors(L, dtype)
    units = orc.Units(10, 0.1)
    bounds = []
    for i in range(n_b):
        mask = torch.rand(res, generator=g) < 0.02
        if i % 2 == 0:
            bounds.append(orc.OracleBoundary("bounce_back", mask=mask))
        else:
            vel = torch.tensor([0.3, -0.2, 0.1][:L.d], dtype=dtype) * (1 + 0.01 * i)
            bounds.append(orc.OracleBoundary("equilibrium_pu", mask=mask, velocity_pu=vel, pressure_pu=torch.tensor(0.001 * i, dtype=dtype)))
    sim = orc.OracleSimulation(L, f0.clone(), "bgk", 0.7, units, bounds)
    entries = []                                          # in the oracle's (= the reference's sorted) index order
    for b in sim.boundaries:
        if b.kind == "bounce_back":
            entries.append({"kind": "bounce_back"})
        else:
            feq = orc.quadratic_equilibrium(units.pressure_pu_to_density_lu(b.pressure_pu), units.velocity_to_lu(b.velocity_pu), e, w)
            entries.append({"kind": "equilibrium", "feq": feq.double().tolist()})
    assert int(sim.no_collision_mask.max()) > min(n_b, 100) - 8               # high indices really occur
    plan = plan_for(lat, dtype, "bgk", res, entries)
    plan.set_masks(dev(sim.no_collision_mask), None if sim.no_streaming_mask is None else dev(sim.no_streaming_mask))
    if two_step:
        plan.set_two_step(1, 0)
        admitted = plan.two_step_admitted() is None
        assert admitted == (n_b <= 15)
    sim.step(5)
    got = run_engine(plan, f0.numpy(), 0.7, 5)
    assert_close(got, sim.f.numpy(), dt, scale=4)
    if two_step:
        assert plan.last_run_info()["two_step_launches"] == (2 if n_b <= 15 else 0)
        one = plan_for(lat, dtype, "bgk", res, entries)
        one.set_masks(dev(sim.no_collision_mask), None if sim.no_streaming_mask is None else dev(sim.no_streaming_mask))
        one.set_two_step(0)
        np.testing.assert_array_equal(got, run_engine(one, f0.numpy(), 0.7, 5))


@pytest.mark.parametrize("lat,res,axis,side", [("D2Q9", [10, 8], 1, -1), ("D3Q19", [8, 6, 7], 0, 1),
                                               ("D3Q27", [6, 7, 8], 2, 1), ("D1Q3", [24], 0, 1),
                                               ("D3Q15", [7, 6, 8], 1, 1)])
def test_abb_outlet_after_lower_index_boundaries(lat, res, axis, side):
    """An AntiBounceBackOutlet whose index is HIGHER than a bounce-back and an equilibrium
    boundary that touch the outlet plane and the plane next to it: the outlet must see the
    velocities of the already bounced / overwritten populations (boundaries are applied in index
    order on the whole field, lettuce/_simulation.py:186-188)."""
    L = orc.LATTICES[lat]
    dtype = torch.float64
    f0 = _random_state(L, res, dtype, 9)
    e, w = orc.lattice_tensors(L, dtype)
    g = torch.Generator().manual_seed(3)
    bb_mask = torch.rand(res, generator=g) < 0.3
    eq_mask = (torch.rand(res, generator=g) < 0.3) & ~bb_mask
    direction = [0] * L.d
    direction[axis] = side
    units = orc.Units(10, 0.1)
    vel = torch.tensor([0.2, 0.1, -0.3][:L.d], dtype=dtype)
    bnds = [orc.OracleBoundary("bounce_back", mask=bb_mask),
            orc.OracleBoundary("equilibrium_pu", mask=eq_mask, velocity_pu=vel,
                               pressure_pu=torch.tensor(0.0, dtype=dtype)),
            orc.OracleBoundary("abb_outlet", direction=direction)]
    sim = orc.OracleSimulation(L, f0.clone(), "bgk", 0.8, units, bnds)
    # the oracle sorts like the reference (ABB first); force the order of this test instead
    sim.boundaries = bnds
    ncm = torch.zeros(res, dtype=torch.uint8)
    nsm = torch.zeros([L.q] + res, dtype=torch.uint8)
    ncm[bb_mask] = 1
    ncm[eq_mask] = 2
    m, s = orc.abb_masks(f0.shape, bnds[2], L)
    ncm[m] = 3
    nsm |= s.to(torch.uint8)
    sim.no_collision_mask, sim.no_streaming_mask = ncm, nsm
    feq = orc.quadratic_equilibrium(units.pressure_pu_to_density_lu(bnds[1].pressure_pu),
                                    units.velocity_to_lu(vel), e, w)
    plan = plan_for(lat, dtype, "bgk", res, [{"kind": "bounce_back"},
                                             {"kind": "equilibrium", "feq": feq.tolist()},
                                             {"kind": "abb_outlet", "axis": axis, "side": side}])
    plan.set_masks(dev(ncm), dev(nsm))
    sim.step(3)
    got = run_engine(plan, f0.numpy(), 0.8, 3)
    # the outlet's neighbour velocity is taken from post-collision populations in the reference
    # and from pre-collision ones in the kernel (collision conserves rho and j to rounding)
    np.testing.assert_allclose(got, sim.f.numpy(), rtol=0, atol=1e-12)


# --------------------------------------------------------------------------- two steps per launch, 2-D
@pytest.mark.parametrize("res,seg", [([8, 64], 0), ([12, 128], 4), ([7, 192], 7), ([16, 512], 0), ([6, 1024], 3),
                                     ([1, 64], 1), ([40, 256], 10)])
@pytest.mark.parametrize("coll", ["none", "bgk"])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_two_step_launch_on_2d_lattices_is_bit_identical_to_two_single_steps(res, seg, coll, dt):
    """lbm2d2_kernel (twostep2d.hpp): strips of 64 .. 512 columns, every segment length incl. 1 and rows that
    wrap; res = [x, y] with y contiguous."""
    dtype = TORCH_DT[dt]
    plan = plan_for("D2Q9", dtype, coll, res)
    plan.set_many_step(0)
    f = dev(_random_state(orc.LATTICES["D2Q9"], res, dtype, 4))
    a, b, c = torch.empty_like(f), torch.empty_like(f), torch.full_like(f, float("nan"))
    plan.stream_collide(f, a, 0.7)
    plan.stream_collide(a, b, 0.7)
    plan.set_two_step(1, seg)
    assert "lbm2d2_kernel" in plan.kernel_name()
    plan.stream_collide_twice(f, c, 0.7)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(c.cpu().numpy(), b.cpu().numpy())


def test_two_step_2d_kernel_reproduces_the_cfg1_vectors_bit_for_bit():
    """BASELINE configs[0] (examples/00_simplest_TGV.py, 128^2 fp64) through the 2-D two-step kernel: the
    reference's populations after 100 steps, exactly."""
    g = golden("tgv2d_d2q9_bgk_128_f64")
    plan = plan_for("D2Q9", torch.float64, "bgk", [128, 128])
    plan.set_many_step(0)
    plan.set_two_step(1)
    np.testing.assert_array_equal(run_engine(plan, g["f0"], float(g["tau"]), 100), g["f100"])
    info = plan.last_run_info()
    assert info["two_step_launches"] == 49 and info["single_step_launches"] == 1 and info["many_step_launches"] == 0


# --------------------------------------------------------------------------- two steps per launch, with boundaries
def _masked_case(lat, res, dtype, abb, seed, with_field=False, abb_first=False, inlet_face=False):
    """Random bounce-back and equilibrium nodes (the latter optionally with a per-node field) plus one
    anti-bounce-back outlet (axis, side) with the masks the reference's boundary builds; abb = None: no outlet.
    inlet_face: the face opposite the outlet consists of equilibrium nodes (an inlet, as in the Obstacle)."""
    L = orc.LATTICES[lat]
    g = torch.Generator().manual_seed(seed)
    f0 = _random_state(L, res, dtype, seed)
    e, w = orc.lattice_tensors(L, dtype)
    bb_mask = torch.rand(res, generator=g) < 0.2
    eq_mask = (torch.rand(res, generator=g) < 0.15) & ~bb_mask
    if inlet_face:
        face = [slice(None)] * L.d
        face[abb[0]] = 0 if abb[1] == 1 else res[abb[0]] - 1
        eq_mask[tuple(face)] = True
        bb_mask[tuple(face)] = False
    units = orc.Units(10, 0.1)
    vel = torch.tensor([0.2, 0.1, -0.3][:L.d], dtype=dtype)
    feq = orc.quadratic_equilibrium(units.pressure_pu_to_density_lu(torch.tensor(0.01, dtype=dtype)),
                                    units.velocity_to_lu(vel), e, w)
    ncm = torch.zeros(res, dtype=torch.uint8)
    nsm = torch.zeros([L.q] + res, dtype=torch.uint8)
    entries = [{"kind": "bounce_back"}, {"kind": "equilibrium", "feq": feq.double().tolist()}]
    if with_field:
        field = (feq.reshape([-1] + [1] * len(res)) * (1 + 0.1 * torch.rand([L.q] + res, generator=g, dtype=torch.float64))).to(dtype)
        entries[1] = {"kind": "equilibrium", "field": field.cuda()}
    m = None
    if abb is not None:
        axis, side = abb
        direction = [0] * L.d
        direction[axis] = side
        m, sm = orc.abb_masks(f0.shape, orc.OracleBoundary("abb_outlet", direction=direction), L)
        nsm |= sm.to(torch.uint8)
        entry = {"kind": "abb_outlet", "axis": axis, "side": side}
        entries = [entry] + entries if abb_first else entries + [entry]
    # later boundaries overwrite earlier ones in no_collision_mask (_simulation.py:63-86)
    for index, e_ in enumerate(entries, start=1):
        ncm[{"bounce_back": bb_mask, "equilibrium": eq_mask}.get(e_["kind"], m)] = index
    return f0, ncm, nsm, entries


MASKED_TWO_STEP = [("D3Q19", [8, 16, 64], "f32", (0, 1)), ("D3Q19", [8, 16, 64], "f32", (0, -1)), ("D3Q19", [6, 8, 128], "f32", (1, 1)),
                   ("D3Q19", [6, 16, 64], "f32", (0, 1)), ("D3Q19", [7, 8, 64], "f32", (0, 1)), ("D3Q19", [4, 8, 64], "f32", (2, -1)),
                   ("D3Q19", [5, 8, 64], "f32", None), ("D3Q27", [6, 8, 64], "f32", (0, 1)), ("D3Q27", [5, 4, 64], "f32", None),
                   ("D3Q15", [6, 8, 64], "f32", (0, 1)), ("D3Q15", [6, 8, 32], "f64", (0, 1)), ("D3Q15", [5, 8, 32], "f64", None),
                   # round 3: D3Q19 fp64 with boundaries on 32 x 4 tiles (8 rows need 168.6 KB of LDS with the third
                   # slot of the downward populations)
                   pytest.param(*("D3Q19", [6, 8, 32], "f64", (0, 1)), marks=pytest.mark.experiments), pytest.param(*("D3Q19", [5, 4, 64], "f64", None), marks=pytest.mark.experiments), pytest.param(*("D3Q19", [4, 12, 32], "f64", (0, 1)), marks=pytest.mark.experiments)]


@pytest.mark.parametrize("lat,res,dt,abb", MASKED_TWO_STEP, ids=[f"{t[0]}-{'x'.join(map(str, t[1]))}-{t[2]}-{t[3]}" for t in map(_values, MASKED_TWO_STEP)])
@pytest.mark.parametrize("coll", ["bgk", "none"])
@pytest.mark.parametrize("seg", [0, 2, 3])
def test_masked_two_step_launch_is_bit_identical_to_two_masked_single_steps(lat, res, dt, abb, coll, seg):
    """lbm2m_kernel: bounce-back, equilibrium (table and per-node field) and an anti-bounce-back outlet at the
    last plane of the sweep axis (reference layout: logical axis 0, side +1 -- the Obstacle's) with its
    no-streaming bits: one two-step launch == two one-step launches, bit for bit, for every segment length.
    Outlets in the other directions are refused (lt_run then keeps the one-step kernel)."""
    dtype = TORCH_DT[dt]
    # seg 2: the outlet has the lowest index, so bounce-back / equilibrium nodes ON the outlet plane come after it
    f0, ncm, nsm, entries = _masked_case(lat, res, dtype, abb, 21, with_field=(seg == 3), abb_first=(seg == 2))
    plan = plan_for(lat, dtype, coll, res, entries)
    plan.set_masks(dev(ncm), dev(nsm))
    _check_masked_two_step(plan, f0, abb is not None and abb != (0, 1), res, seg)


MASKED_TWO_STEP_ROWS = [("D3Q19", [4, 8, 64], "f32", (2, 1)), ("D3Q19", [4, 8, 64], "f32", (2, -1)), ("D3Q19", [5, 16, 128], "f32", (2, 1)),
                        ("D3Q19", [6, 16, 128], "f32", (2, -1)), ("D3Q27", [5, 4, 64], "f32", (2, 1)), ("D3Q27", [4, 8, 128], "f32", (2, -1)),
                        ("D3Q15", [5, 8, 64], "f32", (2, 1)), ("D3Q15", [5, 8, 32], "f64", (2, 1)), ("D3Q15", [4, 16, 64], "f64", (2, -1)),
                        pytest.param(*("D3Q19", [5, 8, 32], "f64", (2, 1)), marks=pytest.mark.experiments), pytest.param(*("D3Q19", [4, 4, 64], "f64", (2, -1)), marks=pytest.mark.experiments)]


@pytest.mark.parametrize("lat,res,dt,abb", MASKED_TWO_STEP_ROWS, ids=[f"{t[0]}-{'x'.join(map(str, t[1]))}-{t[2]}-{t[3]}" for t in map(_values, MASKED_TWO_STEP_ROWS)])
@pytest.mark.parametrize("coll", ["bgk", "none"])
@pytest.mark.parametrize("seg", [0, 2, 3])
def test_masked_two_step_with_the_outlet_at_an_end_of_the_rows(lat, res, dt, abb, coll, seg):
    """lbm2m_kernel, AX = 0: the anti-bounce-back outlet's normal is the contiguous axis (reference layout: z; in the
    slab layout the Obstacle's x) and the face opposite it is an inlet of equilibrium nodes.  The node next to an
    outlet node is the neighbouring lane; the outlet's no-streaming bits are a per-thread constant.  Bit for bit
    two one-step launches; without the inlet face the plan is refused."""
    dtype = TORCH_DT[dt]
    f0, ncm, nsm, entries = _masked_case(lat, res, dtype, abb, 33, with_field=(seg == 3), abb_first=(seg == 2), inlet_face=True)
    plan = plan_for(lat, dtype, coll, res, entries)
    plan.set_masks(dev(ncm), dev(nsm))
    _check_masked_two_step(plan, f0, False, res, seg)
    if seg == 0 and coll == "bgk":
        face = [slice(None)] * 3
        face[abb[0]] = 0 if abb[1] == 1 else res[abb[0]] - 1
        hole = ncm.clone()                                  # one fluid node in the inlet face
        idx = [0, 0, 0]
        idx[abb[0]] = face[abb[0]]
        hole[tuple(idx)] = 0
        plan.set_masks(dev(hole), dev(nsm))
        _check_masked_two_step(plan, f0, True, res, seg)


MASKED_TWO_STEP_2D = [([12, 64], (0, 1)), ([9, 128], (0, 1)), ([8, 512], (0, 1)), ([10, 192], None), ([6, 64], (0, -1)),
                      ([6, 64], (1, 1)), ([4, 1024], (0, 1))]


@pytest.mark.parametrize("res,abb", MASKED_TWO_STEP_2D, ids=[f"{'x'.join(map(str, r))}-{a}" for r, a in MASKED_TWO_STEP_2D])
@pytest.mark.parametrize("coll,dt", [("bgk", "f64"), ("bgk", "f32"), ("none", "f32")])
@pytest.mark.parametrize("seg", [0, 2, 3])
def test_masked_two_step_launch_on_2d_lattices(res, abb, coll, dt, seg):
    """lbm2d2m_kernel: D2Q9 with bounce-back / equilibrium nodes (table and per-node field) and the 2-D Obstacle's
    outlet (last row of the sweep axis x): one launch == two masked one-step launches, bit for bit, strips of 64
    to 512 columns, every segment length, outlet before and after the other boundaries; other outlets refused."""
    dtype = TORCH_DT[dt]
    f0, ncm, nsm, entries = _masked_case("D2Q9", res, dtype, abb, 57, with_field=(seg == 3), abb_first=(seg == 2))
    plan = plan_for("D2Q9", dtype, coll, res, entries)
    plan.set_masks(dev(ncm), dev(nsm))
    plan.set_many_step(0)                                   # small grids: lt_run would otherwise name the many-step kernel
    _check_masked_two_step(plan, f0, abb is not None and abb != (0, 1), res, seg, name="lbm2d2m_kernel")


def _check_masked_two_step(plan, f0, refused, res, seg, name="lbm2m_kernel"):
    f = dev(f0)
    a, b, c = torch.empty_like(f), torch.empty_like(f), torch.full_like(f, float("nan"))
    plan.set_two_step(0)
    plan.stream_collide(f, a, 0.7)
    plan.stream_collide(a, b, 0.7)
    if seg and res[0] % seg:
        seg = 2 if res[0] % 2 == 0 else 0
    plan.set_two_step(1, seg)
    if refused:
        with pytest.raises(Exception, match="two steps per launch"):
            plan.stream_collide_twice(f, c, 0.7)
        return
    plan.stream_collide_twice(f, c, 0.7)
    torch.cuda.synchronize()
    assert name in plan.kernel_name()
    np.testing.assert_array_equal(c.cpu().numpy(), b.cpu().numpy())


# ---- the same in the SLAB layout (z slowest, two ghost planes): what a multi-GPU rank runs.  Round 4: these are the
# instantiations in which hipcc dropped register copies at the join of the equilibrium branch (csrc/Makefile header,
# DESIGN.md section 6); until then only the Obstacle on slabs exercised them.
def _masked_slab_case(lat, res, dtype, outlet, seed, with_field=False, abb_first=False):
    """Random bounce-back / equilibrium nodes in the memory order of a slab-layout plan ([z + 4, y, x]) plus,
    optionally, an anti-bounce-back outlet at x = nx - 1 (side +1) or x = 0 (side -1) with the face opposite it an
    inlet of equilibrium nodes -- the masks lettuce's Obstacle builds on a z-slab (obstacle.py:108-122,
    anti_bounce_back_outlet.py:93-103)."""
    L = orc.LATTICES[lat]
    nx, ny, nz = res
    shape = [nz + 4, ny, nx]
    g = torch.Generator().manual_seed(seed)
    e, w = orc.lattice_tensors(L, dtype)
    f0 = (torch.tensor(L.w, dtype=torch.float64).reshape(-1, 1, 1, 1)
          * (1 + 0.2 * torch.rand([L.q] + shape, generator=g, dtype=torch.float64))).to(dtype)
    bb_mask = torch.rand(shape, generator=g) < 0.2
    eq_mask = (torch.rand(shape, generator=g) < 0.15) & ~bb_mask
    units = orc.Units(10, 0.1)
    vel = torch.tensor([0.2, 0.1, -0.3], dtype=dtype)
    feq = orc.quadratic_equilibrium(units.pressure_pu_to_density_lu(torch.tensor(0.01, dtype=dtype)),
                                    units.velocity_to_lu(vel), e, w)
    entries = [{"kind": "bounce_back"}, {"kind": "equilibrium", "feq": feq.double().tolist()}]
    if with_field:
        field = (feq.reshape(-1, 1, 1, 1) * (1 + 0.1 * torch.rand([L.q] + shape, generator=g, dtype=torch.float64))).to(dtype)
        entries[1] = {"kind": "equilibrium", "field": field.cuda()}
    ncm = torch.zeros(shape, dtype=torch.uint8)
    nsm = torch.zeros([L.q] + shape, dtype=torch.uint8)
    out_mask = None
    if outlet is not None:
        xo, xi = (nx - 1, 0) if outlet == 1 else (0, nx - 1)
        eq_mask[:, :, xi] = True
        bb_mask[:, :, xi] = False
        out_mask = torch.zeros(shape, dtype=torch.bool)
        out_mask[:, :, xo] = True
        for q in range(L.q):
            if L.e[q][0] == -outlet:                       # the populations entering through the outlet are not streamed
                nsm[q, :, :, xo] = 1
        entry = {"kind": "abb_outlet", "axis": 0, "side": outlet}
        entries = [entry] + entries if abb_first else entries + [entry]
    for index, e_ in enumerate(entries, start=1):
        ncm[{"bounce_back": bb_mask, "equilibrium": eq_mask}.get(e_["kind"], out_mask)] = index
    return f0, ncm, nsm, entries


MASKED_SLAB = [("D3Q19", [64, 16, 10], "f32"), ("D3Q27", [64, 8, 9], "f32"), ("D3Q15", [128, 8, 8], "f32"),
               ("D3Q15", [32, 16, 7], "f64"), pytest.param(*("D3Q19", [32, 8, 8], "f64"), marks=pytest.mark.experiments)]


@pytest.mark.parametrize("lat,res,dt", MASKED_SLAB, ids=[f"{t[0]}-{'x'.join(map(str, t[1]))}-{t[2]}" for t in map(_values, MASKED_SLAB)])
@pytest.mark.parametrize("outlet", [None, 1, -1])
@pytest.mark.parametrize("coll", ["bgk", "none"])
@pytest.mark.parametrize("seg", [0, 2, 3])
def test_masked_two_step_in_the_slab_layout_is_bit_identical_to_two_masked_single_steps(lat, res, dt, outlet, coll, seg):
    """lbm2m_kernel<..., LAYOUT 1, ...> on a plan with two ghost planes: random bounce-back and equilibrium nodes
    (table / per-node field), with and without the Obstacle's outlet along the contiguous axis x (either end), BGK
    and streaming only, every segment length: one launch over the interior planes == two masked one-step launches,
    bit for bit; and the plan's first-use check has run and passed."""
    from lettuce_amd._native import Plan, LAYOUT_SLAB
    dtype = TORCH_DT[dt]
    f0, ncm, nsm, entries = _masked_slab_case(lat, res, dtype, outlet, 71, with_field=(seg == 3), abb_first=(seg == 2))
    plan = Plan(lat, dtype, coll, res, entries, layout=LAYOUT_SLAB, ghost_planes=2)
    plan.set_masks(dev(ncm), dev(nsm))
    f = dev(f0)
    a, b, c = torch.zeros_like(f), torch.zeros_like(f), torch.full_like(f, float("nan"))
    n2 = f.shape[1]
    plan.stream_collide_planes(f, a, 0.7, 1, n2 - 1)
    plan.stream_collide_planes(a, b, 0.7, 2, n2 - 2)
    plan.set_two_step(1, seg)
    assert plan.canary_status()["status"] == 0
    plan.stream_collide_twice_planes(f, c, 0.7, 2, n2 - 2)
    torch.cuda.synchronize()
    assert "lbm2m_kernel" in plan.kernel_name() and ", 1, " in plan.kernel_name()
    assert plan.canary_status() == {"status": 1, "mismatches": 0, "message": ""}
    np.testing.assert_array_equal(c[:, 2:n2 - 2].cpu().numpy(), b[:, 2:n2 - 2].cpu().numpy())


def test_first_use_check_of_the_masked_two_step_kernel_and_its_fallback():
    """Every plan with masks holds its two-step kernel against two one-step launches before it uses it
    (lt_plan_set_canary, include/lettuce_hip.h).  A plan whose check fails (forced here) keeps the one-step kernel:
    lt_run gives the one-step result and says why in lt_last_error, the explicit entry points refuse."""
    from lettuce_amd._native import NativeEngineError
    dtype = torch.float32
    res = [6, 8, 64]
    f0, ncm, nsm, entries = _masked_case("D3Q19", res, dtype, (0, 1), 5)
    tau = 0.7

    def plan_with(mode):
        plan = plan_for("D3Q19", dtype, "bgk", res, entries)
        plan.set_masks(dev(ncm), dev(nsm))
        plan.set_resident(0)
        plan.set_canary(mode)
        plan.set_two_step(1, 0)
        return plan

    one = plan_with(1)
    one.set_two_step(0)
    want = run_engine(one, f0, tau, 7)
    assert one.last_run_info()["two_step_launches"] == 0 and one.canary_status()["status"] == 0    # never needed

    checked = plan_with(1)
    np.testing.assert_array_equal(run_engine(checked, f0, tau, 7), want)
    assert checked.last_run_info()["two_step_launches"] == 3
    assert checked.canary_status() == {"status": 1, "mismatches": 0, "message": ""}
    checked.set_masks(dev(ncm), dev(nsm))                     # new masks: checked again at the next use
    assert checked.canary_status()["status"] == 0
    assert "lbm2m_kernel" in checked.kernel_name() and checked.canary_status()["status"] == 1

    trusted = plan_with(0)
    np.testing.assert_array_equal(run_engine(trusted, f0, tau, 7), want)
    assert trusted.last_run_info()["two_step_launches"] == 3 and trusted.canary_status()["status"] == 2

    failed = plan_with(2)
    np.testing.assert_array_equal(run_engine(failed, f0, tau, 7), want)       # the one-step kernel did the work
    info = failed.last_run_info()
    assert info["two_step_launches"] == 0 and info["single_step_launches"] == 6
    status = failed.canary_status()
    assert status["status"] == -1 and "first-use check" in status["message"] and "one-step kernel" in status["message"]
    assert "first-use check" in failed.lib.lt_last_error().decode()
    assert "lbm2m_kernel" not in failed.kernel_name() and "lbm_kernel" in failed.kernel_name()
    assert "first-use check" in failed.two_step_admitted()
    f = dev(f0)
    with pytest.raises(NativeEngineError, match="first-use check"):
        failed.stream_collide_twice(f, torch.empty_like(f), tau)
    failed.set_canary(1)                                       # checked for real now
    np.testing.assert_array_equal(run_engine(failed, f0, tau, 7), want)
    assert failed.last_run_info()["two_step_launches"] == 3 and failed.canary_status()["status"] == 1


def test_masked_two_step_is_refused_when_no_streaming_bits_lie_off_the_outlet_plane():
    """The admission test of the masked two-step kernel runs on the device when the masks are compiled."""
    f0, ncm, nsm, entries = _masked_case("D3Q19", [4, 8, 64], torch.float32, (0, 1), 3)
    nsm[5, 2, 3, 7] = 1                                    # a stray bit on a fluid node
    plan = plan_for("D3Q19", torch.float32, "bgk", [4, 8, 64], entries)
    plan.set_masks(dev(ncm), dev(nsm))
    plan.set_two_step(1)
    assert "lbm_kernel" in plan.kernel_name()
    a, b = dev(f0), torch.empty_like(dev(f0))
    out, _ = plan.run(a, b, 0.7, 5)                        # lt_run falls back to the one-step kernel
    assert plan.last_run_info()["two_step_launches"] == 0
    with pytest.raises(Exception, match="two steps per launch"):
        plan.stream_collide_twice(a, b, 0.7)


MASKED_GOLDEN = [("obstacle3d_d3q19_bgk_12x16x64_f32", "D3Q19", "f32", (1, 2, 3, 8)),
                 ("obstacle3d_d3q27_bgk_10x8x64_f32", "D3Q27", "f32", (1, 2, 3, 8)),
                 ("obstacle2d_d2q9_bgk_24x64_f64", "D2Q9", "f64", (1, 2, 3, 8))]


@pytest.mark.parametrize("name,lat,dt,snaps", MASKED_GOLDEN, ids=[t[0] for t in MASKED_GOLDEN])
def test_masked_two_step_on_obstacle_vectors_of_the_reference(name, lat, dt, snaps):
    """The reference's Obstacle (inlet + anti-bounce-back outlet along x = the sweep axis + sphere bounce-back)
    on grids the two-step tiles take: lt_run with pairing == lt_run without, exactly, and both match the
    reference's populations at the outlet's rounding level."""
    g = golden(name)
    plan = obstacle_plan(g, lat, "bgk", dt)
    plan.set_many_step(0)
    for n in snaps:
        plan.set_two_step(0)
        single = run_engine(plan, g["f0"], float(g["tau"]), n)
        plan.set_two_step(1)
        paired = run_engine(plan, g["f0"], float(g["tau"]), n)
        info = plan.last_run_info()
        assert info["two_step_launches"] == (n - 1) // 2, info
        np.testing.assert_array_equal(paired, single)
        assert_close(paired, g[f"f{n}"], dt, scale=10 if dt == "f64" else 1)


TWO_OUTLETS = [("two_outlets_d2q9_bgk_f64", "D2Q9", "f64"), ("two_outlets_d3q19_bgk_f64", "D3Q19", "f64"),
               ("two_outlets_d3q19_bgk_f32", "D3Q19", "f32"),
               # round 3: any number of outlets on at most two axes (+x, +y, -y; +x, +z, -z; +-y, +-z)
               ("three_outlets_d2q9_bgk_f64", "D2Q9", "f64"), ("three_outlets_d3q19_bgk_f32", "D3Q19", "f32"),
               ("four_outlets_d3q27_bgk_f64", "D3Q27", "f64"),
               # round 4: outlets on all three axes (+x +y +z; +x -y +y +z -z; -z +y +x -y), planes meeting in corners
               ("outlets_on_three_axes_d3q19_bgk_f64", "D3Q19", "f64"),
               ("outlets_on_three_axes_d3q27_bgk_f32", "D3Q27", "f32"),
               ("outlets_on_three_axes_d3q15_bgk_f64", "D3Q15", "f64")]


@pytest.mark.parametrize("name,lat,dt", TWO_OUTLETS, ids=[t[0] for t in TWO_OUTLETS])
def test_two_anti_bounce_back_outlets_against_the_reference(name, lat, dt):
    """Plans with several outlets whose planes meet in edges (VERDICT r01 item 8, r02 "missing" 5; reference: any list
    of boundaries, lettuce/_simulation.py:57-86, anti_bounce_back_outlet.py:22-103): on such an edge an outlet's
    neighbour has already been rewritten by the outlets with a lower index, so its state is rebuilt in full
    (neighbour_moments, DEPTH 1) -- enough for any number of outlets on at most two axes; with outlets on all three
    axes the planes meet in corners and the chain is one longer (DEPTH 2, reference layout; round 4)."""
    g = golden(name)
    L = orc.LATTICES[lat]
    dtype = TORCH_DT[dt]
    units = orc.tgv_units([int(r) for r in g["resolution"]], 100, 0.05)
    e, w = orc.lattice_tensors(L, dtype)
    feq_in = orc.quadratic_equilibrium(units.pressure_pu_to_density_lu(torch.tensor(0, dtype=dtype)),
                                       units.velocity_to_lu(torch.tensor(g["inlet_velocity_pu"], dtype=dtype)), e, w)
    entries = []
    for kind, direction in zip(g["boundary_order"].tolist(), g["boundary_direction"].tolist()):
        if kind == "AntiBounceBackOutlet":
            axis = [i for i, c in enumerate(direction) if c][0]
            entries.append({"kind": "abb_outlet", "axis": axis, "side": direction[axis]})
        elif kind == "BounceBackBoundary":
            entries.append({"kind": "bounce_back"})
        else:
            entries.append({"kind": "equilibrium", "feq": feq_in.double().tolist()})
    plan = plan_for(lat, dtype, "bgk", g["f0"].shape[1:], entries)
    plan.set_masks(dev(g["no_collision_mask"]), dev(unpack_nsm(g)))
    axes = {e["axis"] for e in entries if e["kind"] == "abb_outlet"}
    assert plan.kernel_name().endswith(f", {len(axes) - 1}>")       # the instantiation for that depth
    for n in (1, 2, 6):
        got = run_engine(plan, g["f0"], float(g["tau"]), n)
        assert_close(got, g[f"f{n}"], dt, scale=10 if dt == "f64" else 1)
    if L.d == 3 and len(axes) == 3:
        # on slabs outlets on all three axes are refused, with the reason
        from lettuce_amd._native import Plan, LAYOUT_SLAB
        with pytest.raises(Exception, match="all three axes"):
            Plan(lat, dtype, "bgk", [int(r) for r in g["resolution"]], entries, layout=LAYOUT_SLAB, ghost_planes=1)


BIT_IDENTICAL = [t for t in TGV if t[2] == "bgk"]


@pytest.mark.parametrize("name,lat,coll,dt,snaps", BIT_IDENTICAL, ids=[t[0] for t in BIT_IDENTICAL])
def test_periodic_bgk_is_bit_identical_to_the_reference(name, lat, coll, dt, snaps):
    """For periodic BGK flows the kernel reproduces the reference's floating-point arithmetic
    operation for operation -- torch.sum's cascade order for rho, the GEMM order for j and e.u,
    IEEE division by the (dtype-rounded) cs^2 constants, no fused multiply-adds in feq and in
    the relaxation -- so the populations equal the reference CPU path's bit for bit, in fp32
    and in fp64, after up to 100 steps.  (KBC and the ABB outlet stay at rounding level by design:
    one reciprocal instead of two divisions, neighbour moments from pre-collision populations.)"""
    g = golden(name)
    plan = plan_for(lat, TORCH_DT[dt], coll, g["f0"].shape[1:])
    for n in snaps:
        np.testing.assert_array_equal(run_engine(plan, g["f0"], float(g["tau"]), n), g[f"f{n}"])


def test_cfg1_populations_bit_identical_after_100_steps():
    g = golden("tgv2d_d2q9_bgk_128_f64")
    plan = plan_for("D2Q9", torch.float64, "bgk", [128, 128])
    np.testing.assert_array_equal(run_engine(plan, g["f0"], float(g["tau"]), 100), g["f100"])


# --------------------------------------------------------------------------- exact division on the device
@pytest.mark.parametrize("dt,which", [("f32", 0), ("f32", 1), ("f64", 0), ("f64", 1)])
def test_exact_division_emulation_on_the_device(dt, which):
    """div_cs on the GPU against IEEE division by the reference's rounded constant, bit for bit, on a stratified
    sample of bit patterns: every binade from 1e-30 |x / D| up to 1e30 (the range the host-side exhaustive check
    covers, tests/aux/exact_division_check.c) plus random mantissas -- which needs fp32 denormals to survive v_mul /
    v_fma (x r_lo is denormal near the lower end).  Below the cut-off the emulation may differ from the
    quotient; the result must still be finite and within 2 ulp."""
    from lettuce_amd._native import probe_div_cs
    ftype, itype = (np.float32, np.uint32) if dt == "f32" else (np.float64, np.uint64)
    d = ftype((2.0 if which == 0 else 1.0) * orc.CS2) if hasattr(orc, "CS2") else ftype((2.0 if which == 0 else 1.0) * (1 / np.sqrt(3.0)) ** 2)
    rng = np.random.default_rng(5)
    mant_bits, bias = (23, 127) if dt == "f32" else (52, 1023)
    lo, hi = (-100, 100) if dt == "f32" else (-1000, 1000)          # exponents: 1e-30 .. 1e30 / 1e-301 .. 1e301
    exps = np.arange(lo, hi + 1)
    mant = rng.integers(0, 1 << mant_bits, size=(len(exps), 4096), dtype=np.uint64)
    mant[:, :8] = [0, 1, 2, (1 << mant_bits) - 1, (1 << mant_bits) - 2, 1 << (mant_bits - 1), (1 << (mant_bits - 1)) - 1, 3]
    bits = ((exps[:, None] + bias).astype(np.uint64) << np.uint64(mant_bits)) | mant
    x = np.concatenate([bits.astype(itype).view(ftype).ravel(), -bits.astype(itype).view(ftype).ravel()])
    got = probe_div_cs(dev(x), which).cpu().numpy()
    want = (x / d).astype(ftype)
    np.testing.assert_array_equal(got.view(itype), want.view(itype))
    # below the cut-off: tiny arguments down to the smallest normals
    tiny = (ftype(1e-37) if dt == "f32" else ftype(1e-307)) * (1 + rng.random(4096).astype(ftype))
    g = probe_div_cs(dev(tiny), which).cpu().numpy()
    w = (tiny / d).astype(ftype)
    assert np.isfinite(g).all() and (np.abs(g - w) <= 2 * np.spacing(w)).all()


# --------------------------------------------------------------------------- KBC: one arithmetic in every kernel
@pytest.mark.parametrize("lat,res,dt", [("D3Q27", [8, 8, 64], "f32"), ("D3Q27", [6, 10, 12], "f64"), ("D2Q9", [64, 48], "f32"),
                                        ("D2Q9", [16, 24], "f64")])
def test_kbc_kernels_agree_bit_for_bit(lat, res, dt):
    """Round 4: collide_kbc is compiled without multiply-add contraction (two explicit fma_t), so the fused
    stream-collide kernel, collide followed by stream, lt_run's batches and -- on small 2-D grids -- the many-step
    kernel return the same bits (before, hipcc fused different products in every inlining context and the KBC kernels
    agreed at rounding level only; the reference's vectors keep their tolerance: its arithmetic is not reproduced
    operation for operation)."""
    dtype = TORCH_DT[dt]
    L = orc.LATTICES[lat]
    f0 = dev(_random_state(L, res, dtype, 3))
    plan = plan_for(lat, dtype, "kbc", res)
    plan.set_many_step(0)
    tau = 0.55
    a, b = f0.clone(), torch.empty_like(f0)
    for _ in range(4):                                   # four whole steps: collide, stream
        plan.collide(a, b, tau)
        plan.stream(b, a)
    fused = plan.run(f0.clone(), torch.empty_like(f0), tau, 4)[0]       # collide, 3 fused, stream
    assert torch.equal(fused, a)
    if lat == "D2Q9":
        many = plan_for(lat, dtype, "kbc", res)
        many.set_many_step(1)
        out = many.run(f0.clone(), torch.empty_like(f0), tau, 4)[0]
        if many.last_run_info()["many_step_launches"]:
            assert torch.equal(out, a)


# --------------------------------------------------------------------------- BGK in fast arithmetic (opt-in)
FAST = [("tgv3d_d3q19_bgk_16_f32", "D3Q19", "f32"), ("tgv3d_d3q19_bgk_32_f32", "D3Q19", "f32"),
        ("tgv3d_d3q19_bgk_16_f64", "D3Q19", "f64"), ("tgv3d_d3q27_bgk_16_f64", "D3Q27", "f64")]


@pytest.mark.experiments
@pytest.mark.parametrize("name,lat,dt", FAST, ids=[t[0] for t in FAST])
def test_fast_arithmetic_bgk_stays_inside_the_stated_tolerances(name, lat, dt):
    """lt_plan_set_arithmetic(plan, 1): the shorter BGK collision (one reciprocal of rho, cs^2 = 1/3, contracted
    multiply-adds, moments over opposite pairs) against the reference's own vectors, with the tolerances of SURVEY.md
    8(d): fp32 max |df| <= 1e-5 max |f| after 10 steps and the kinetic energy to 5e-5 over 100 steps (the
    reference's own fp32 / fp64 gap is 1.3e-5 at step 100); fp64 1e-12 after 100 steps, energy 1e-9.  The 1e-6 over
    10 steps is MISSED: 1.0-1.5e-6, depending on the build -- the reference's fp32 energy carries a drift of -1.1e-7
    per step from its rounded divisors, the fast form reproduces it only statistically (u = RN(j / rho (1 - 3e-8))
    inside one FMA) and v_rcp_f32's last bit moves the momentum by as much (a 1-ulp bias of the reciprocal is 3e-6 of
    energy after 10 steps in a host-side emulation).  One of the two reasons the arithmetic is not in the product
    library; asserted here at 2e-6."""
    g = golden(name)
    res = list(g["f0"].shape[1:])
    plan = plan_for(lat, TORCH_DT[dt], "bgk", res)
    plan.set_arithmetic("fast")
    assert ", 3, " in plan.kernel_name()
    scale = float(np.abs(g["f0"]).max())
    f10 = run_engine(plan, g["f0"], float(g["tau"]), 10)
    assert np.abs(f10 - g["f10"]).max() <= (1e-5 if dt == "f32" else 1e-13) * scale
    if "f100" in g:
        f100 = run_engine(plan, g["f0"], float(g["tau"]), 100)
        assert np.abs(f100 - g["f100"]).max() <= (1e-4 if dt == "f32" else 1e-12) * scale
    units = orc.tgv_units(res, float(g["reynolds"]), float(g["mach"]))
    to_pu = units.incompressible_energy_to_pu(1.0) * units.length_to_pu(1.0) ** 3
    cur, other = dev(g["f0"]), torch.empty_like(dev(g["f0"]))
    done = 0
    for step, want in zip(g["energy_steps"].tolist(), g["energy_pu"].tolist()):
        if step > done:
            cur, other = plan.run(cur, other, float(g["tau"]), step - done)
            done = step
        got = float(plan.kinetic_energy_lu(cur).cpu()) * to_pu
        tol = (2e-6 if step <= 10 else 5e-5) if dt == "f32" else 1e-9
        assert got == pytest.approx(want, rel=tol), (step, got, want)


@pytest.mark.experiments
def test_fast_arithmetic_in_the_two_step_kernel_and_where_it_is_refused():
    """the same arithmetic in lbm2_kernel<..., 3, ...> (two updates per launch): equal to the one-step kernel in fast
    arithmetic bit for bit, within tolerance of the reference's vectors; plans with boundaries, KBC and 2-D plans
    have no such kernel and say so."""
    from lettuce_amd._native import NativeEngineError
    g = golden("tgv3d_d3q19_bgk_8x16x64_f32")
    res = list(g["f0"].shape[1:])
    one = plan_for("D3Q19", torch.float32, "bgk", res)
    one.set_arithmetic("fast"); one.set_two_step(0)
    two = plan_for("D3Q19", torch.float32, "bgk", res)
    two.set_arithmetic(1); two.set_two_step(1)
    a, b = run_engine(one, g["f0"], float(g["tau"]), 10), run_engine(two, g["f0"], float(g["tau"]), 10)
    assert two.last_run_info()["two_step_launches"] == 4 and "lbm2_kernel" in two.kernel_name() and ", 3, " in two.kernel_name()
    np.testing.assert_array_equal(a, b)
    assert np.abs(b - g["f10"]).max() <= 1e-5 * float(np.abs(g["f10"]).max())
    assert not np.array_equal(b, g["f10"])                     # rounding level, not bit for bit: why it is opt-in
    two.set_arithmetic("exact")
    np.testing.assert_array_equal(run_engine(two, g["f0"], float(g["tau"]), 10), g["f10"])
    for lat, coll, shape, entries in (("D3Q27", "kbc", [8, 8, 8], []), ("D2Q9", "bgk", [16, 16], []),
                                      ("D3Q19", "bgk", [8, 8, 8], [{"kind": "bounce_back"}])):
        plan = plan_for(lat, torch.float32, coll, shape, entries)
        with pytest.raises(NativeEngineError, match="fast arithmetic"):
            plan.set_arithmetic("fast")


# --------------------------------------------------------------------------- two steps per launch
TWO_STEP_GOLDEN = [("tgv3d_d3q19_bgk_8x16x64_f32", "D3Q19", "f32", (1, 2, 3, 10)),
                   ("tgv3d_d3q19_bgk_8x8x32_f64", "D3Q19", "f64", (1, 2, 3, 10)),
                   ("tgv3d_d3q27_bgk_4x8x64_f32", "D3Q27", "f32", (2, 3, 10)),
                   ("tgv3d_d3q15_bgk_8x8x64_f32", "D3Q15", "f32", (2, 3, 10))]


@pytest.mark.parametrize("name,lat,dt,snaps", TWO_STEP_GOLDEN, ids=[t[0] for t in TWO_STEP_GOLDEN])
def test_two_step_kernel_reproduces_the_reference_vectors_bit_for_bit(name, lat, dt, snaps):
    """Direct vectors for lbm2_kernel: the reference's own CPU path stepped on grids the two-step kernel
    takes; lt_run with the pairing forced on must return the reference's populations exactly, and must
    really have gone through two-step launches (n steps = 1 collide + (n-1) fused + 1 stream)."""
    g = golden(name)
    plan = plan_for(lat, TORCH_DT[dt], "bgk", g["f0"].shape[1:])
    plan.set_two_step(1)
    for n in snaps:
        np.testing.assert_array_equal(run_engine(plan, g["f0"], float(g["tau"]), n), g[f"f{n}"])
        info = plan.last_run_info()
        assert info["two_step_launches"] == (n - 1) // 2 and info["single_step_launches"] == (n - 1) % 2, info


@pytest.mark.parametrize("res,seg", [([4, 8, 64], 0), ([8, 16, 64], 4), ([6, 24, 128], 3), ([1, 8, 64], 1),
                                     ([12, 40, 192], 0), ([5, 8, 64], 5)])
@pytest.mark.parametrize("coll", ["none", "bgk"])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_two_step_launch_is_bit_identical_to_two_single_steps(res, seg, coll, dt):
    """lt_stream_collide_twice (intermediate state in LDS, halo'd tiles, plane sweep with wrap) against
    two lt_stream_collide launches: every segment length incl. 1, tiles that wrap in both tiled
    axes, several tiles per axis; fp32 (64 x 8 tiles) and fp64 (32 x 8 tiles)."""
    plan = plan_for("D3Q19", TORCH_DT[dt], coll, res)
    plan.set_two_step(1, seg)
    torch.manual_seed(3)
    w = torch.rand(19, 1, 1, 1, device="cuda", dtype=TORCH_DT[dt]) * 0.05 + 0.02
    f = (w * (1 + 0.1 * torch.rand(plan.f_shape, device="cuda", dtype=TORCH_DT[dt]))).contiguous()
    a, b, c = torch.empty_like(f), torch.empty_like(f), torch.empty_like(f)
    plan.stream_collide(f, a, 0.6)
    plan.stream_collide(a, b, 0.6)
    plan.stream_collide_twice(f, c, 0.6)
    assert torch.equal(b, c)


@pytest.mark.parametrize("lat,dt", [("D3Q15", "f32"), ("D3Q15", "f64"), ("D3Q27", "f32")])
def test_two_step_launch_on_the_other_lattices(lat, dt):
    """the kernel is generic in the lattice; rows of 256 bytes, 8 or 4 rows per tile depending on LDS"""
    res = [5, 16, 128]
    plan = plan_for(lat, TORCH_DT[dt], "bgk", res)
    torch.manual_seed(23)
    L = orc.LATTICES[lat]
    w = torch.tensor(L.w, dtype=TORCH_DT[dt], device="cuda").reshape(L.q, 1, 1, 1)
    f = (w * (1 + 0.1 * torch.rand([L.q] + res, dtype=TORCH_DT[dt], device="cuda"))).contiguous()
    a, b, c = torch.empty_like(f), torch.empty_like(f), torch.empty_like(f)
    plan.stream_collide(f, a, 0.8)
    plan.stream_collide(a, b, 0.8)
    plan.stream_collide_twice(f, c, 0.8)
    assert torch.equal(b, c)
    sim = orc.OracleSimulation(L, f.cpu().clone(), "bgk", 0.8)
    plan.set_two_step(1)
    r, _ = plan.run(f.clone(), torch.empty_like(f), 0.8, 5)
    assert plan.last_run_info()["two_step_launches"] == 2
    sim.step(5)
    assert_close(r.cpu().numpy(), sim.f.numpy(), dt)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 8])
def test_lt_run_with_paired_steps_equals_lt_run_without(n):
    """odd and even numbers of fused steps: pairs through the two-step kernel plus a single one"""
    res = [8, 16, 64]
    L = orc.LATTICES["D3Q19"]
    torch.manual_seed(5)
    w = torch.tensor(L.w, dtype=torch.float32, device="cuda").reshape(19, 1, 1, 1)
    f0 = (w * (1 + 0.1 * torch.rand([19] + res, device="cuda"))).contiguous()
    outs = []
    for mode in (0, 1):
        plan = plan_for("D3Q19", torch.float32, "bgk", res)
        plan.set_two_step(mode)
        r, other = plan.run(f0.clone(), torch.empty_like(f0), 0.7, n)
        info = plan.last_run_info()
        assert info["two_step_launches"] == (mode * ((n - 1) // 2))
        assert info["single_step_launches"] == (n - 1) - 2 * info["two_step_launches"]
        outs.append((r.clone(), other.clone()))
    assert torch.equal(outs[0][0], outs[1][0])        # post-streaming populations
    assert torch.equal(outs[0][1], outs[1][1])        # post-collision populations of the last step
    ref = orc.OracleSimulation(L, f0.cpu().clone(), "bgk", 0.7)
    ref.step(n)
    assert_close(outs[1][0].cpu().numpy(), ref.f.numpy(), "f32")


def test_two_step_unsupported_combinations_fail_loudly():
    from lettuce_amd._native import NativeEngineError
    for lat, dt, coll, res in (("D3Q27", torch.float64, "bgk", [4, 8, 64]), ("D3Q19", torch.float64, "bgk", [4, 8, 48]),
                               ("D3Q19", torch.float32, "bgk", [4, 8, 96]), ("D3Q19", torch.float32, "bgk", [4, 6, 64]),
                               ("D3Q27", torch.float32, "kbc", [4, 8, 64])):
        plan = plan_for(lat, dt, coll, res)
        f = torch.rand(plan.f_shape, device="cuda", dtype=dt)
        with pytest.raises(NativeEngineError):
            plan.stream_collide_twice(f, torch.empty_like(f), 0.6)
        # lt_run falls back to single steps by itself
        plan.set_two_step(1)
        plan.run(f, torch.empty_like(f), 0.6, 4)
        assert plan.last_run_info()["two_step_launches"] == 0


def test_two_step_on_a_slab_with_two_ghost_planes_equals_two_single_steps():
    """lt_stream_collide_twice_planes on a slab-layout plan (no wrap along z, ragged last segment) and
    the two-step halo pack / unpack pair."""
    from lettuce_amd._native import Plan, LAYOUT_SLAB
    res = [64, 16, 10]                                   # local slab: nx, ny, nz
    plan = Plan("D3Q19", torch.float32, "bgk", res, [], layout=LAYOUT_SLAB, ghost_planes=2)
    plan.set_two_step(1, 4)                              # 10 planes in segments of 4, 4, 2
    torch.manual_seed(11)
    f = (0.05 + 0.01 * torch.rand(plan.f_shape, device="cuda")).contiguous()
    a, b, c = torch.zeros_like(f), torch.zeros_like(f), torch.zeros_like(f)
    n2 = f.shape[1]
    plan.stream_collide_planes(f, a, 0.6, 1, n2 - 1)
    plan.stream_collide_planes(a, b, 0.6, 2, n2 - 2)
    plan.stream_collide_twice_planes(f, c, 0.6, 2, n2 - 2)
    assert torch.equal(b[:, 2:n2 - 2], c[:, 2:n2 - 2])
    plan.stream_collide_twice_planes(f, c, 0.6, 3, 6)    # a sub-range
    assert torch.equal(b[:, 3:6], c[:, 3:6])
    # halo message: 9 in-plane + 5 + 5 crossing blocks; unpack(pack) moves interior planes to ghosts
    msg = torch.empty([19, 16, 64], device="cuda")
    g = f.clone()
    plan.pack_two_step(f, -1, msg)                       # my lower planes 2, 3 ...
    plan.unpack_two_step(g, +1, msg)                     # ... become the upper ghosts of the rank below
    e = np.array(orc.LATTICES["D3Q19"].e)
    inp, down = np.nonzero(e[:, 2] == 0)[0], np.nonzero(e[:, 2] == -1)[0]
    assert torch.equal(g[inp, n2 - 2], f[inp, 2]) and torch.equal(g[down, n2 - 2], f[down, 2])
    assert torch.equal(g[down, n2 - 1], f[down, 3])
    up = np.nonzero(e[:, 2] == 1)[0]
    assert torch.equal(g[up, n2 - 2], f[up, n2 - 2])      # untouched


def test_two_step_edge_launch_writes_the_same_halo_message_as_the_pack_kernel():
    from lettuce_amd._native import Plan, LAYOUT_SLAB
    plan = Plan("D3Q19", torch.float32, "bgk", [64, 16, 12], [], layout=LAYOUT_SLAB, ghost_planes=2)
    torch.manual_seed(13)
    f = (0.05 + 0.01 * torch.rand(plan.f_shape, device="cuda")).contiguous()
    n2 = f.shape[1]
    a, b = torch.zeros_like(f), torch.zeros_like(f)
    plan.stream_collide_twice_planes(f, a, 0.7, 2, n2 - 2)
    want_down, want_up = torch.empty([19, 16, 64], device="cuda"), torch.empty([19, 16, 64], device="cuda")
    plan.pack_two_step(a, -1, want_down)
    plan.pack_two_step(a, +1, want_up)
    got_down, got_up = torch.zeros_like(want_down), torch.zeros_like(want_up)
    plan.stream_collide_twice_planes_packed(f, b, 0.7, 2, 5, pack_lower=got_down)
    plan.stream_collide_twice_planes_packed(f, b, 0.7, n2 - 4, n2 - 2, pack_upper=got_up)
    assert torch.equal(b[:, 2:5], a[:, 2:5]) and torch.equal(b[:, n2 - 4:n2 - 2], a[:, n2 - 4:n2 - 2])
    assert torch.equal(got_down, want_down)
    assert torch.equal(got_up, want_up)


# --------------------------------------------------------------------------- many steps per launch (2-D)
@pytest.mark.parametrize("res", [[8, 8], [16, 24], [128, 128], [40, 8]])
@pytest.mark.parametrize("coll,dt", [("bgk", "f64"), ("bgk", "f32"), ("none", "f32"), ("kbc", "f64"), ("kbc", "f32")])
def test_many_steps_per_launch_equal_single_steps(res, coll, dt):
    """lt_stream_collide_many (K <= 8 steps in LDS, recomputed halo, neighbourhoods that wrap around
    tiny grids several times) against K lt_stream_collide launches, every K."""
    plan = plan_for("D2Q9", TORCH_DT[dt], coll, res)
    torch.manual_seed(17)
    w = torch.tensor(orc.LATTICES["D2Q9"].w, dtype=TORCH_DT[dt], device="cuda").reshape(9, 1, 1)
    f = (w * (1 + 0.1 * torch.rand([9] + res, dtype=TORCH_DT[dt], device="cuda"))).contiguous()
    a, b = f.clone(), torch.empty_like(f)
    for k in range(1, 9):
        plan.stream_collide(a, b, 0.7)
        a, b = b, a
        got = torch.empty_like(f)
        plan.stream_collide_many(f, got, 0.7, k)
        if coll == "kbc":      # KBC may contract multiply-adds differently in the two kernels: rounding level
            assert float((got - a).abs().max()) <= (1e-5 if dt == "f32" else 1e-13) * float(a.abs().max()), k
        else:
            assert torch.equal(got, a), k


MANY_MASKED = [([16, 24], None), ([24, 16], (0, 1)), ([24, 16], (0, -1)), ([16, 32], (1, 1)), ([8, 40], (1, -1)), ([8, 8], (0, 1))]


@pytest.mark.parametrize("res,abb", MANY_MASKED, ids=[f"{'x'.join(map(str, r))}-{a}" for r, a in MANY_MASKED])
@pytest.mark.parametrize("coll,dt", [("bgk", "f64"), ("bgk", "f32"), ("none", "f32")])
@pytest.mark.parametrize("variant", ["table", "field-outlet-first"])
def test_many_steps_per_launch_with_boundaries_equal_single_steps(res, abb, coll, dt, variant):
    """lbm_many_kernel<..., MASKED>: random bounce-back / equilibrium nodes (table or per-node field), no-streaming
    bits and an anti-bounce-back outlet along either axis, before or after the other boundaries in index order: K
    steps in one launch == K masked one-step launches, bit for bit, for every K (7 with an outlet: one more
    ring of nodes is recomputed so that the outlet's neighbour is valid)."""
    dtype = TORCH_DT[dt]
    f0, ncm, nsm, entries = _masked_case("D2Q9", res, dtype, abb, 41, with_field=(variant != "table"),
                                         abb_first=(variant != "table"))
    if abb is None:
        nsm[3, 2, 5] = 1                                   # stray no-streaming bits on ordinary nodes
        nsm[7, 6, 1] = 1
    plan = plan_for("D2Q9", dtype, coll, res, entries)
    plan.set_masks(dev(ncm), dev(nsm))
    f = dev(f0)
    a, b = f.clone(), torch.empty_like(f)
    kmax = 8 if abb is None else 7
    for k in range(1, kmax + 1):
        plan.stream_collide(a, b, 0.7)
        a, b = b, a
        got = torch.full_like(f, float("nan"))
        plan.stream_collide_many(f, got, 0.7, k)
        np.testing.assert_array_equal(got.cpu().numpy(), a.cpu().numpy(), err_msg=f"K = {k}")
    if abb is not None:
        with pytest.raises(Exception, match="n_steps"):
            plan.stream_collide_many(f, got, 0.7, 8)


MANY_GOLDEN = [("obstacle2d_d2q9_bgk_40x24_f64", "f64"), ("obstacle2d_d2q9_bgk_40x24_f32", "f32")]


@pytest.mark.parametrize("name,dt", MANY_GOLDEN, ids=[t[0] for t in MANY_GOLDEN])
def test_many_step_launches_on_the_obstacle_vectors_of_the_reference(name, dt):
    """The reference's 2-D Obstacle (inlet, outlet, cylinder) on a grid of multiples of 8: lt_run packs its fused
    steps into launches of up to 7, the result equals the step-by-step path exactly and the reference's
    populations at the outlet's rounding level."""
    g = golden(name)
    plan = obstacle_plan(g, "D2Q9", "bgk", dt)
    for n in (1, 2, 9, 20):
        plan.set_many_step(0)
        single = run_engine(plan, g["f0"], float(g["tau"]), n)
        plan.set_many_step(-1)                              # automatic: 40 x 24 nodes are well below the limit
        packed = run_engine(plan, g["f0"], float(g["tau"]), n)
        info = plan.last_run_info()
        assert info["many_step_launches"] == (0 if n < 3 else -(-(n - 1) // 7)), info
        np.testing.assert_array_equal(packed, single)
        assert_close(packed, g[f"f{n}"], dt, scale=10 if dt == "f64" else 1)
    assert "lbm_many_kernel" in plan.kernel_name() and "true" in plan.kernel_name()


def test_lt_run_uses_many_step_launches_on_small_2d_grids():
    """cfg1's shape: lt_run issues ceil((n-1)/8) launches for its fused steps and the populations are
    those of the step-by-step path (and bit-identical to the reference vectors, tested elsewhere)."""
    res = [128, 128]
    w = torch.tensor(orc.LATTICES["D2Q9"].w, dtype=torch.float64, device="cuda").reshape(9, 1, 1)
    torch.manual_seed(19)
    f0 = (w * (1 + 0.1 * torch.rand([9] + res, dtype=torch.float64, device="cuda"))).contiguous()
    outs = []
    for mode in (0, -1):
        plan = plan_for("D2Q9", torch.float64, "bgk", res)
        plan.set_many_step(mode)
        r, other = plan.run(f0.clone(), torch.empty_like(f0), 0.61, 20)
        info = plan.last_run_info()
        assert info["many_step_launches"] == (0 if mode == 0 else 3)       # 19 fused steps = 8 + 8 + 3
        outs.append((r.clone(), other.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("lat,dt,res", [("D3Q19", "f32", [256, 256, 256]), ("D3Q19", "f64", [384, 384, 96]),
                                        ("D3Q27", "f32", [256, 256, 128])],
                         ids=["cfg2-shape", "cfg5-shape", "d3q27"])
def test_two_step_path_full_size_long_run_is_bit_identical_to_the_one_step_path(lat, dt, res):
    """BASELINE-size grids, 201 steps through lt_run with and without paired steps: any race on the
    LDS planes or any addressing slip of the two-step kernel would show as a difference."""
    L = orc.LATTICES[lat]
    torch.manual_seed(0)
    w = torch.rand(L.q, 1, 1, 1, device="cuda", dtype=TORCH_DT[dt]) * 0.03 + 0.02
    f0 = (w * (1 + 0.05 * torch.rand([L.q] + res, device="cuda", dtype=TORCH_DT[dt]))).contiguous()
    outs = []
    for mode in (0, 1):
        plan = plan_for(lat, TORCH_DT[dt], "bgk", res)
        plan.set_two_step(mode)
        r, other = plan.run(f0.clone(), torch.empty_like(f0), 0.55, 201)
        assert plan.last_run_info()["two_step_launches"] == 100 * mode
        outs.append(r.clone())
        del plan, r, other
    assert torch.equal(outs[0], outs[1])
    assert bool(torch.isfinite(outs[1]).all())


def test_both_slab_edges_in_one_launch_with_and_without_messages():
    from lettuce_amd._native import Plan, LAYOUT_SLAB
    plan = Plan("D3Q19", torch.float32, "bgk", [64, 16, 14], [], layout=LAYOUT_SLAB, ghost_planes=2)
    torch.manual_seed(29)
    f = (0.05 + 0.01 * torch.rand(plan.f_shape, device="cuda")).contiguous()
    n2 = f.shape[1]
    ref = torch.zeros_like(f)
    plan.stream_collide_twice_planes(f, ref, 0.7, 2, n2 - 2)
    want_down, want_up = torch.empty([19, 16, 64], device="cuda"), torch.empty([19, 16, 64], device="cuda")
    plan.pack_two_step(ref, -1, want_down)
    plan.pack_two_step(ref, +1, want_up)
    for edge in (2, 3, 5):
        for packed in (False, True):
            out = torch.zeros_like(f)
            down, up = torch.zeros_like(want_down), torch.zeros_like(want_up)
            if packed:
                plan.stream_collide_twice_edges(f, out, 0.7, edge, pack_lower=down, pack_upper=up)
                assert torch.equal(down, want_down) and torch.equal(up, want_up)
            else:
                plan.stream_collide_twice_edges(f, out, 0.7, edge)
            assert torch.equal(out[:, 2:2 + edge], ref[:, 2:2 + edge])
            assert torch.equal(out[:, n2 - 2 - edge:n2 - 2], ref[:, n2 - 2 - edge:n2 - 2])
            assert float(out[:, 2 + edge:n2 - 2 - edge].abs().max()) == 0.0      # interior untouched


def test_two_step_kernel_randomised_shapes_lattices_dtypes_and_ranges():
    """40 random cases (fixed seed): lattice, dtype, grid, collision, periodic or slab layout with a
    random sub-range of output planes, segment length -- always bit-identical to two single steps."""
    import random
    from lettuce_amd._native import Plan, LAYOUT_SLAB
    rng = random.Random(2024)
    for _ in range(40):
        dt = rng.choice([torch.float32, torch.float64])
        lat = rng.choice(["D3Q19", "D3Q15"] + (["D3Q27"] if dt == torch.float32 else []))
        width = 64 if dt == torch.float32 else 32            # rows of 256 bytes; 8 or 4 rows per tile
        n0, n1, n2 = width * rng.randint(1, 3), 8 * rng.randint(1, 4), rng.randint(1, 14)
        coll = rng.choice(["none", "bgk"])
        slab = rng.random() < 0.4 and n2 >= 3
        plan = (Plan(lat, dt, coll, [n0, n1, n2], [], layout=LAYOUT_SLAB, ghost_planes=2) if slab
                else Plan(lat, dt, coll, [n2, n1, n0], []))
        seg = rng.choice([0, 1, 2, 3, 5])
        if seg and not slab and n2 % seg:
            seg = 0
        plan.set_two_step(1, seg)
        f = (torch.rand(plan.f_shape, device="cuda", dtype=dt) * 0.01 + 0.04).contiguous()
        a, b, c = torch.zeros_like(f), torch.zeros_like(f), torch.zeros_like(f)
        case = (lat, str(dt), [n0, n1, n2], coll, slab, seg)
        if slab:
            m = f.shape[1]
            plan.stream_collide_planes(f, a, 0.7, 1, m - 1)
            plan.stream_collide_planes(a, b, 0.7, 2, m - 2)
            lo = rng.randint(2, m - 3)
            hi = rng.randint(lo + 1, m - 2)
            plan.stream_collide_twice_planes(f, c, 0.7, lo, hi)
            assert torch.equal(b[:, lo:hi], c[:, lo:hi]), case
        else:
            plan.stream_collide(f, a, 0.7)
            plan.stream_collide(a, b, 0.7)
            plan.stream_collide_twice(f, c, 0.7)
            assert torch.equal(b, c), case


def test_two_step_slab_reads_only_the_exchanged_populations_of_the_ghost_planes():
    """Everything in the ghost planes that the two-step halo message does not carry is poisoned with
    NaN: the output planes must not change (the driver leaves those slots uninitialised)."""
    from lettuce_amd._native import Plan, LAYOUT_SLAB
    plan = Plan("D3Q19", torch.float32, "bgk", [64, 16, 9], [], layout=LAYOUT_SLAB, ghost_planes=2)
    torch.manual_seed(31)
    f = (0.05 + 0.01 * torch.rand(plan.f_shape, device="cuda")).contiguous()
    n2 = f.shape[1]
    ref = torch.zeros_like(f)
    plan.stream_collide_twice_planes(f, ref, 0.7, 2, n2 - 2)
    e = np.array(orc.LATTICES["D3Q19"].e)
    up, inp, down = (np.nonzero(e[:, 2] == v)[0] for v in (1, 0, -1))
    g = f.clone()
    g[down, 1] = float("nan"); g[np.concatenate([inp, down]), 0] = float("nan")          # lower ghosts
    g[up, n2 - 2] = float("nan"); g[np.concatenate([inp, up]), n2 - 1] = float("nan")    # upper ghosts
    out = torch.zeros_like(f)
    plan.stream_collide_twice_planes(g, out, 0.7, 2, n2 - 2)
    assert torch.equal(out[:, 2:n2 - 2], ref[:, 2:n2 - 2])


# --------------------------------------------------------------------------- round 3: padded populations
@pytest.mark.parametrize("name,lat,dt,snaps", TWO_STEP_GOLDEN, ids=[t[0] for t in TWO_STEP_GOLDEN])
@pytest.mark.parametrize("pad", [-1, 0, 64, 2368])
def test_resident_populations_reproduce_the_reference_vectors_bit_for_bit(name, lat, dt, snaps, pad):
    """the fused steps on the ENGINE's padded ping-pong buffers (lt_resident_load / _advance /
    _store) against the reference's own vectors on grids the two-step kernel takes -- whatever the distance between the
    populations (engine's choice, dense, 256 B, 9472 B), the caller's dense tensors see the reference's bits, and the
    steps really went through two-step launches."""
    g = golden(name)
    plan = plan_for(lat, TORCH_DT[dt], "bgk", g["f0"].shape[1:])
    plan.set_two_step(1)
    plan.set_resident(1, pad)
    on, stride = plan.resident_enabled()
    nodes = int(np.prod(g["f0"].shape[1:]))
    assert on and stride >= nodes and (pad < 0 or stride - nodes in range(pad, pad + 64))
    f0, out = dev(g["f0"]), torch.empty_like(dev(g["f0"]))
    done = 0
    for n in snaps:                                      # carry on from the resident state between the looks
        if done == 0:
            plan.resident_load(f0, float(g["tau"]))
            plan.resident_advance(float(g["tau"]), n - 1)
        else:
            plan.resident_advance(float(g["tau"]), n - done)
        info = plan.last_run_info()
        fused = n - 1 if done == 0 else n - done
        assert info["two_step_launches"] == fused // 2 and info["single_step_launches"] == fused % 2, info
        plan.resident_store(out)
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"f{n}"])
        done = n
    assert torch.equal(f0, dev(g["f0"]))                # the caller's populations were only read
    plan.resident_free()
    with pytest.raises(Exception, match="no resident populations"):
        plan.resident_advance(float(g["tau"]), 1)


def test_simulation_on_resident_populations_equals_the_dense_path():
    """lt.Simulation with the engine-owned buffers switched on (what `automatic` does beyond the caches) against the
    same steps on the caller's dense tensors: batches, looks in between, in-place edits and re-assignments of flow.f,
    odd and even counts -- flow.f is the same tensor of bits either way, and stays a plain [q, *res] tensor."""
    import lettuce_amd as lt
    c = lt.Context("cuda:0", torch.float32, use_native=True)

    def fresh(resident):
        flow = lt.TaylorGreenVortex(c, [16, 32, 64], 400, 0.1, lt.D3Q19())
        sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
        sim._native.plan.set_two_step(1)
        sim._native.plan.set_resident(1 if resident else 0)
        return flow, sim
    (fa, sa), (fb, sb) = fresh(True), fresh(False)
    for k in (4, 3, 3):
        sa(k); sb(k)
    assert fa._pending is not None                      # nobody has looked yet
    assert torch.equal(fa.f, fb.f) and fa.f.is_contiguous() and list(fa.f.shape) == [19, 16, 32, 64]
    sa(2); sb(2)                                        # carries on from what was shown
    assert sa._native.plan.last_run_info()["two_step_launches"] == 1
    assert torch.equal(fa.f, fb.f)
    fa.f[3] *= 1.01; fb.f[3] *= 1.01                    # in-place edit: the resident state is stale
    sa(3); sb(3)
    assert torch.equal(fa.f, fb.f)
    sa(5)                                               # a pass is pending ...
    fa.f = fb.f.clone()                                 # ... and dropped by an assignment
    assert fa._pending is None
    sa(2); sb(2)
    assert torch.equal(fa.f, fb.f)
    assert torch.equal(fa.f_next[0], fa.f_next[0])      # f_next is a plain tensor too
    rho = fa.rho()
    assert torch.isfinite(rho).all()


@pytest.mark.parametrize("lat,coll,dt", [("D3Q19", "bgk", "f32"), ("D3Q27", "kbc", "f32"), ("D3Q19", "bgk", "f64")])
def test_slab_kernels_with_a_population_stride_equal_the_dense_ones(lat, coll, dt):
    """lt_plan_set_population_stride: every slab-layout entry point (collide / stream / fused planes, the two-step
    launches, pack / unpack, the reductions) on padded tensors against the same calls on dense ones."""
    from lettuce_amd._native import Plan, LAYOUT_SLAB
    T = TORCH_DT[dt]
    res = [64, 16, 12]
    dense = Plan(lat, T, coll, res, [], layout=LAYOUT_SLAB, ghost_planes=2)
    padded = Plan(lat, T, coll, res, [], layout=LAYOUT_SLAB, ghost_planes=2)
    nodes = 64 * 16 * 16
    padded.set_population_stride(nodes + 64 * (3 if dt == "f32" else 5))
    torch.manual_seed(5)
    f = (0.05 + 0.01 * torch.rand(dense.f_shape, device="cuda", dtype=T)).contiguous()
    fp = padded.populations_like(f)
    assert not fp.is_contiguous() and torch.equal(fp, f)
    n2 = f.shape[1]
    a, ap = torch.zeros_like(f), padded.empty_populations().zero_()
    for name, args in (("collide_planes", (0.6, 2, n2 - 2)), ("stream_collide_planes", (0.6, 1, n2 - 1))):
        getattr(dense, name)(f, a, *args)
        getattr(padded, name)(fp, ap, *args)
        assert torch.equal(a, ap), name
    dense.stream_planes(f, a, 1, n2 - 1)
    padded.stream_planes(fp, ap, 1, n2 - 1)
    assert torch.equal(a, ap)
    if coll == "bgk":
        dense.stream_collide_twice_planes(f, a, 0.6, 2, n2 - 2)
        padded.stream_collide_twice_planes(fp, ap, 0.6, 2, n2 - 2)
        assert torch.equal(a, ap)
        q_msg = dense.two_step_message_blocks()
        m, mp = torch.zeros([q_msg, 16, 64], device="cuda", dtype=T), torch.zeros([q_msg, 16, 64], device="cuda", dtype=T)
        dense.pack_two_step(a, -1, m); padded.pack_two_step(ap, -1, mp)
        assert torch.equal(m, mp)
        dense.unpack_two_step(a, +1, m); padded.unpack_two_step(ap, +1, mp)
        assert torch.equal(a, ap)
    assert float(dense.kinetic_energy_lu(f)) == float(padded.kinetic_energy_lu(fp))
    assert float(dense.mass(f)) == float(padded.mass(fp))
    with pytest.raises(Exception, match="population stride"):
        padded.set_population_stride(nodes + 3)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("edge", [2, 3])
def test_direct_edge_launch_reads_the_received_messages_and_writes_the_outgoing_ones(dt, edge):
    """lt_stream_collide_twice_edges_direct: the planes beyond the cuts come from the halo
    messages as they arrived, the ghost planes of the field hold garbage; its output planes and both outgoing messages
    are those of the launch that reads unpacked ghost planes followed by the pack kernel -- on padded slab tensors."""
    from lettuce_amd._native import Plan, LAYOUT_SLAB
    T = TORCH_DT[dt]
    nx = 64 if dt == "f32" else 32
    plan = Plan("D3Q19", T, "bgk", [nx, 16, 12], [], layout=LAYOUT_SLAB, ghost_planes=2)
    plan.set_population_stride(nx * 16 * 16 + 128)
    torch.manual_seed(17)
    f = plan.populations_like(0.05 + 0.01 * torch.rand(plan.f_shape, device="cuda", dtype=T))
    n2 = f.shape[1]
    from_below = (0.05 + 0.01 * torch.rand([19, 16, nx], device="cuda", dtype=T)).contiguous()
    from_above = (0.05 + 0.01 * torch.rand([19, 16, nx], device="cuda", dtype=T)).contiguous()
    # the classic way: scatter the messages into the ghost planes, launch, gather the outgoing messages
    g = plan.populations_like(f)
    plan.unpack_two_step(g, -1, from_below)
    plan.unpack_two_step(g, +1, from_above)
    want = plan.empty_populations().zero_()
    plan.stream_collide_twice_planes(g, want, 0.7, 2, n2 - 2)
    want_down, want_up = torch.empty_like(from_below), torch.empty_like(from_below)
    plan.pack_two_step(want, -1, want_down)
    plan.pack_two_step(want, +1, want_up)
    # the direct way: ghost planes poisoned
    f[:, :2] = float("nan"); f[:, n2 - 2:] = float("nan")
    got = plan.empty_populations().zero_()
    got_down, got_up = torch.zeros_like(from_below), torch.zeros_like(from_below)
    plan.stream_collide_twice_edges_direct(f, got, 0.7, edge, from_below, from_above, got_down, got_up)
    if n2 - 4 > 2 * edge:
        plan.stream_collide_twice_planes(f, got, 0.7, 2 + edge, n2 - 2 - edge)
    assert torch.equal(got[:, 2:n2 - 2], want[:, 2:n2 - 2])
    assert torch.equal(got_down, want_down) and torch.equal(got_up, want_up)
    # without messages the same entry point reads the ghost planes of the field
    got2 = plan.empty_populations().zero_()
    plan.stream_collide_twice_edges_direct(g, got2, 0.7, edge, None, None, got_down.zero_(), got_up.zero_())
    assert torch.equal(got2[:, 2:2 + edge], want[:, 2:2 + edge]) and torch.equal(got_down, want_down)
    with pytest.raises(Exception, match="both received messages or"):
        plan.stream_collide_twice_edges_direct(f, got, 0.7, edge, from_below, None, got_down, got_up)


# --------------------------------------------------------------------------- round 3: two steps per launch, small 3-D grids
@pytest.mark.experiments
@pytest.mark.parametrize("lat,dt", [("D3Q19", "f32"), ("D3Q19", "f64"), ("D3Q27", "f32"), ("D3Q15", "f32"), ("D3Q15", "f64")])
@pytest.mark.parametrize("res", [[8, 8, 8], [16, 8, 24], [32, 32, 32], [40, 16, 8]])
@pytest.mark.parametrize("coll", ["bgk", "none"])
def test_two_steps_per_launch_on_small_3d_grids_equal_single_steps(lat, dt, res, coll):
    """lbm_many3d_kernel: the 10^3 neighbourhood of an 8^3 tile in LDS, two stream-collide steps
    per launch: bit for bit two launches of the one-step kernel, incl. grids smaller than the neighbourhood (8^3:
    every neighbour is a periodic image of the tile itself) and ragged tile counts."""
    T = TORCH_DT[dt]
    L = orc.LATTICES[lat]
    plan = plan_for(lat, T, coll, res)
    torch.manual_seed(7)
    w = torch.rand(L.q, 1, 1, 1, device="cuda", dtype=T) * 0.03 + 0.02
    f = (w * (1 + 0.05 * torch.rand([L.q] + res, device="cuda", dtype=T))).contiguous()
    a, b, c = torch.empty_like(f), torch.empty_like(f), torch.empty_like(f)
    plan.stream_collide(f, a, 0.7)
    plan.stream_collide(a, b, 0.7)
    plan.stream_collide_many(f, c, 0.7, 2)
    assert torch.equal(b, c)
    with pytest.raises(Exception, match="no kernel|n_steps"):
        plan.stream_collide_many(f, c, 0.7, 3)


@pytest.mark.experiments
@pytest.mark.parametrize("lat,dt", [("D3Q19", "f32"), ("D3Q19", "f64"), ("D3Q15", "f32")])
@pytest.mark.parametrize("res,planes", [([8, 8, 64], 0), ([6, 12, 128], 0), ([16, 16, 64], 4), ([5, 4, 64], 1), ([64, 64, 128], 0)])
@pytest.mark.parametrize("coll", ["bgk", "none"])
def test_three_steps_per_launch_equal_three_single_steps(lat, dt, res, planes, coll):
    """lbm3_kernel (round 3, threestep.hpp): three stream-collide steps per launch with BOTH intermediate states in LDS
    (tiles of 64 / 32 x 4 nodes and their one- and two-node halos, sixteen waves with one phase -- or two -- each, two
    barriers per plane): bit for bit three launches of the one-step kernel, for every segment length, on grids whose
    halos wrap around onto the tile itself.  (Opt-in entry point: measured slower per update than two steps per
    launch)"""
    T = TORCH_DT[dt]
    L = orc.LATTICES[lat]
    if dt == "f64":
        res = [res[0], res[1], max(32, res[2] // 2)]
    plan = plan_for(lat, T, coll, res)
    plan.set_two_step(1, planes)
    torch.manual_seed(11)
    w = torch.rand(L.q, 1, 1, 1, device="cuda", dtype=T) * 0.03 + 0.02
    f = (w * (1 + 0.05 * torch.rand([L.q] + res, device="cuda", dtype=T))).contiguous()
    a, b = f.clone(), torch.empty_like(f)
    for _ in range(3):
        plan.stream_collide(a, b, 0.7)
        a, b = b, a
    out = torch.empty_like(f)
    plan.stream_collide_thrice(f, out, 0.7)
    assert torch.equal(out, a)


@pytest.mark.experiments
def test_three_steps_per_launch_reproduce_the_reference_vectors():
    """... and chained: 3 x 3 steps through lbm3_kernel after the collide-only launch, then the streaming pass, against
    the reference's populations after 10 steps (periodic BGK: bit for bit)."""
    g = golden("tgv3d_d3q19_bgk_8x16x64_f32")
    plan = plan_for("D3Q19", torch.float32, "bgk", g["f0"].shape[1:])
    tau = float(g["tau"])
    a = torch.tensor(g["f0"], device="cuda")
    b = torch.empty_like(a)
    plan.collide(a, b, tau)
    for _ in range(3):
        plan.stream_collide_thrice(b, a, tau)
        a, b = b, a
    plan.stream(b, a)
    np.testing.assert_array_equal(a.cpu().numpy(), g["f10"])


@pytest.mark.experiments
def test_three_steps_per_launch_are_refused_where_they_do_not_apply():
    plan = plan_for("D3Q27", torch.float32, "bgk", [64, 8, 8])        # two levels of D3Q27 do not fit the LDS
    f = torch.rand(plan.f_shape, device="cuda") * 0.01 + 0.03
    with pytest.raises(Exception, match="no kernel"):
        plan.stream_collide_thrice(f, torch.empty_like(f), 0.7)
    plan = plan_for("D3Q19", torch.float32, "bgk", [8, 6, 64])        # middle extent % 4
    f = torch.rand(plan.f_shape, device="cuda") * 0.01 + 0.03
    with pytest.raises(Exception, match="no kernel"):
        plan.stream_collide_thrice(f, torch.empty_like(f), 0.7)


@pytest.mark.experiments
@pytest.mark.parametrize("name,dt,n", [("tgv3d_d3q19_bgk_32_f32", "f32", 10), ("tgv3d_d3q19_bgk_16_f64", "f64", 100)])
def test_small_3d_grids_run_two_steps_per_launch_and_reproduce_the_reference(name, dt, n):
    """lt_run on a launch-bound 3-D grid with lt_plan_set_many_step(plan, 1) pairs its fused steps into
    lbm_many3d_kernel launches and returns the reference's populations bit for bit (periodic BGK), for odd and even
    step counts.  (Not automatic: measured slower than one launch per step, api.hip many_step_wanted.)"""
    g = golden(name)
    plan = plan_for("D3Q19", TORCH_DT[dt], "bgk", g["f0"].shape[1:])
    assert "lbm_many3d_kernel" not in plan.kernel_name()
    plan.set_many_step(1)
    assert "lbm_many3d_kernel" in plan.kernel_name()
    np.testing.assert_array_equal(run_engine(plan, g["f0"], float(g["tau"]), n), g[f"f{n}"])
    info = plan.last_run_info()
    assert info["many_step_launches"] == (n - 1) // 2 and info["single_step_launches"] == (n - 1) % 2 and info["two_step_launches"] == 0, info
    np.testing.assert_array_equal(run_engine(plan, g["f0"], float(g["tau"]), n - 1 if f"f{n - 1}" in g else n), g[f"f{n - 1}"] if f"f{n - 1}" in g else g[f"f{n}"])
    plan.set_many_step(0)
    assert "lbm_many3d_kernel" not in plan.kernel_name()
    np.testing.assert_array_equal(run_engine(plan, g["f0"], float(g["tau"]), n), g[f"f{n}"])


@pytest.mark.parametrize("name,lat,dt,snaps", MASKED_GOLDEN[:2], ids=[t[0] for t in MASKED_GOLDEN[:2]])
@pytest.mark.parametrize("pad", [-1, 192])
def test_masked_two_step_on_resident_populations_equals_the_dense_path(name, lat, dt, snaps, pad):
    """Large Obstacle flows run their fused steps as masked two-step launches on the engine's padded buffers
    (automatic beyond the caches): forced on here on the reference's Obstacle vectors -- boundary tables, masks and
    per-node bytes stay dense while the populations are padded -- the caller's tensors must see exactly what lt_run
    gives on dense buffers, and the reference's populations at the outlet's rounding level."""
    g = golden(name)
    plan = obstacle_plan(g, lat, "bgk", dt)
    plan.set_many_step(0)
    plan.set_two_step(1)
    tau = float(g["tau"])
    f0, out = dev(g["f0"]), torch.empty_like(dev(g["f0"]))
    for n in snaps:
        plan.set_resident(0)
        dense = run_engine(plan, g["f0"], tau, n)
        plan.set_resident(1, pad)
        assert plan.resident_enabled()[0]
        plan.resident_load(f0, tau)
        plan.resident_advance(tau, n - 1)
        assert plan.last_run_info()["two_step_launches"] == (n - 1) // 2
        plan.resident_store(out)
        np.testing.assert_array_equal(out.cpu().numpy(), dense)
        assert_close(dense, g[f"f{n}"], dt)


@pytest.mark.parametrize("driver,dt,res", [("SlabSimulation", "f64", [20, 12, 10]), ("TwoStepSlabSimulation", "f32", [64, 8, 12]),
                                           ("TwoStepSlabSimulation", "f64", [32, 16, 8])])
def test_enstrophy_and_mass_of_a_slab_on_the_device(driver, dt, res):
    """lt_slab_velocity / lt_slab_enstrophy / lt_slab_mass_interior (one rank: its own neighbour) against the oracle's
    Enstrophy and Mass observables (observable_reporter.py:45-68, 140-158) on the same populations -- one and two
    ghost planes, dense and padded populations; fp64 partial sums against torch's pairwise fp32 sums in the tolerance."""
    import lettuce_amd as lt
    dtype = TORCH_DT[dt]
    ctx = lt.Context("cuda:0", dtype, use_native=True)
    slab = lt.ZSlab(res, 0, 1)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
    sim = getattr(lt, driver)(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
    mask = torch.zeros(res, dtype=torch.bool)
    mask[1:5, 0:3, res[2] - 2:] = True                      # touches the last global z plane and the y border
    mask_ext = mask[:, :, slab.z_indices()]
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", dtype)
    tol = 1e-11 if dt == "f64" else 2e-5
    for n in (0, 3):
        if n:
            sim(n); ref.step(n)
        f = sim.gather_f().cpu()
        want_e = float(orc.enstrophy_pu(f.double(), ref.lat, ref.units))
        want_m = float(orc.mass_observable(f.double(), mask))
        want_p = float(orc.mass_observable(f.double(), None))
        assert sim.enstrophy_pu() == pytest.approx(want_e, rel=tol)
        assert sim.mass_interior(mask_ext) == pytest.approx(want_m, rel=1e-12 if dt == "f64" else 1e-7)
        assert sim.mass_interior(None) == pytest.approx(want_p, rel=1e-12 if dt == "f64" else 1e-7)
    # what the entry points refuse: a single-domain plan
    from lettuce_amd._native import Plan, NativeEngineError
    dense = Plan("D3Q19", dtype, "bgk", res, [])
    with pytest.raises(NativeEngineError, match="slab layout"):
        dense._check(dense.lib.lt_slab_enstrophy(dense._handle, None, 1.0, 1.0, None, None))


@pytest.mark.parametrize("lat,dt,res,masked", [("D3Q19", "f32", [512, 512, 512], False), ("D3Q27", "f64", [320, 320, 256], False),
                                               ("D3Q19", "f32", [384, 384, 384], True), ("D3Q27", "f32", [384, 384, 320], True)],
                         ids=["d3q19-f32-512^3-10GiB", "d3q27-f64-5.7GB", "d3q19-f32-masked-4.3GB", "d3q27-f32-masked-5.1GB"])
def test_population_buffers_beyond_4_gib(lat, dt, res, masked):
    """Maximum sizes: population buffers of 5.7-10.2 GB, i.e. byte offsets beyond 2^32 within one tensor (32-bit node
    indices per population, 64-bit population offsets -- DESIGN section 3).  Size-independent properties: the fused
    kernels against collide + stream through the operator entry points, two updates per launch against one (where the
    lattice / dtype has the kernel) bit for bit, mass conserved, and the LAST population's last plane really moved
    (an offset that wrapped at 4 GiB would leave it or clobber a low address)."""
    L = orc.LATTICES[lat]
    dtype = TORCH_DT[dt]
    torch.manual_seed(5)
    w = torch.rand(L.q, 1, 1, 1, device="cuda", dtype=dtype) * 0.03 + 0.02
    f0 = torch.empty([L.q] + res, device="cuda", dtype=dtype)
    for q in range(L.q):                                  # plane-wise: no second buffer of the same size for rand()
        f0[q] = w[q] * (1 + 0.05 * torch.rand(res, device="cuda", dtype=dtype))
    assert f0.numel() * f0.element_size() > 4 * 2 ** 30
    entries = []
    if masked:
        entries = [{"kind": "bounce_back"}]
    plan = plan_for(lat, dtype, "bgk", res, entries)
    if masked:
        ncm = torch.zeros(res, dtype=torch.uint8, device="cuda")
        ncm[res[0] - 9:res[0] - 3, 5:11, res[2] - 8:res[2] - 2] = 1          # a block near the END of the buffer
        nsm = torch.zeros([L.q] + res, dtype=torch.uint8, device="cuda")
        plan.set_masks(ncm, nsm)
        del nsm
    mass0 = float(plan.mass(f0))
    tau, n = 0.6, 4
    # (1) n steps, one update per launch
    plan.set_two_step(0)
    one, other = plan.run(f0.clone(), torch.empty_like(f0), tau, n)
    assert plan.last_run_info()["two_step_launches"] == 0
    del other
    # (2) the same by the operator entry points: collide, stream, n times
    c, d = f0.clone(), torch.empty_like(f0)
    for _ in range(n):
        plan.collide(c, d, tau)
        plan.stream(d, c)
    del d
    assert torch.equal(one, c), "fused launches != collide + stream beyond 4 GiB"
    del c
    # (3) two updates per launch where the kernel exists
    plan2 = plan_for(lat, dtype, "bgk", res, entries)
    if masked:
        plan2.set_masks(ncm, torch.zeros([L.q] + res, dtype=torch.uint8, device="cuda"))
    plan2.set_two_step(1)
    two, other = plan2.run(f0.clone(), torch.empty_like(f0), tau, n)
    del other
    info = plan2.last_run_info()
    if lat == "D3Q27" and (dt == "f64" or masked):
        # no fp64 D3Q27 two-step kernel; with masks the D3Q27 kernel addresses the field with 32-bit offsets and leaves
        # fields of 4 GiB and more to the one-step kernel (its per-population instantiation was not faster)
        assert info["two_step_launches"] == 0
    else:
        # with masks: the instantiation with one buffer descriptor per population (fields of 4 GiB and more)
        if masked:
            assert plan2.kernel_name().startswith("lbm2m_kernel")
        assert info["two_step_launches"] >= 1, info
    assert torch.equal(two, one), "two updates per launch != one beyond 4 GiB"
    # (4) conservation (periodic, bounce-back) and the far end of the buffer
    assert float(plan.mass(two)) == pytest.approx(mass0, rel=1e-6 if dt == "f32" else 1e-13)
    assert not torch.equal(two[-1, -1], f0[-1, -1])
    assert bool(torch.isfinite(two[-1]).all()) and bool(torch.isfinite(two[0]).all())
