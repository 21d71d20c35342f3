"""The reference's invariant / integration tests restated for this package, over the same fixture
matrix (tests/conftest.py:66-84 of the reference): cpu (torch ops) always; cuda non-native and cuda
native (HIP engine) when a GPU is present (marked ``gpu``).  Sources: tests/collision/
test_collision_relaxes_shear_moments.py, test_collision_optimizes_pseudo_entropy.py,
tests/flow/test_initialize_fneq.py, test_flow.py, test_obstacle.py,
tests/reporter/test_generic_reporters.py, tests/boundary/test_bc_masks.py."""
import io

import numpy as np
import pytest
import torch

import lettuce_amd as lt
from test_host_api import UniformFlow

CONFIGS = [pytest.param(("cpu", torch.float64, False), id="cpu-f64"),
           pytest.param(("cpu", torch.float32, False), id="cpu-f32"),
           pytest.param(("cuda", torch.float64, False), id="cuda-f64", marks=pytest.mark.gpu),
           pytest.param(("cuda", torch.float64, True), id="cuda-f64-native", marks=pytest.mark.gpu),
           pytest.param(("cuda", torch.float32, True), id="cuda-f32-native", marks=pytest.mark.gpu)]
STENCILS = [lt.D1Q3, lt.D2Q9, lt.D3Q15, lt.D3Q19, lt.D3Q27]


def context(cfg):
    device, dtype, native = cfg
    return lt.Context(device=device, dtype=dtype, use_native=native)


def quiet(fn, *a, **k):
    with pytest.MonkeyPatch.context() as mp:
        mp.setattr("sys.stdout", io.StringIO())
        return fn(*a, **k)


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("stencil", STENCILS, ids=[s.__name__ for s in STENCILS])
@pytest.mark.parametrize("collision", [lt.BGKCollision, lt.KBCCollision])
def test_collision_relaxes_shear_moments(cfg, stencil, collision):
    if collision is lt.KBCCollision and stencil not in (lt.D2Q9, lt.D3Q27):
        pytest.skip("KBCCollision only implemented for D2Q9 and D3Q27")
    st = stencil()
    flow = UniformFlow(context(cfg), [16] * st.d, 100, 0.1, st)
    torch.manual_seed(4)
    flow.f = flow.f * (1 + 0.01 * torch.rand_like(flow.f))
    feq = flow.equilibrium(flow)
    pre, pre_eq = flow.shear_tensor(), flow.shear_tensor(feq)
    tau = 0.6 if collision is lt.BGKCollision else flow.units.relaxation_parameter_lu   # KBC takes the units' tau
    post = flow.shear_tensor(collision(tau)(flow))
    expect = pre - 1 / tau * (pre - pre_eq)
    assert post.cpu().numpy() == pytest.approx(expect.cpu().numpy(), abs=1e-5)


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("stencil", [lt.D2Q9, lt.D3Q27], ids=["D2Q9", "D3Q27"])
def test_kbc_pseudo_entropy_not_below_bgk(cfg, stencil):
    st = stencil()
    flow = UniformFlow(context(cfg), [16] * st.d, 100, 0.1, st)
    np.random.seed(1)
    flow.f = flow.context.convert_to_tensor(np.random.random([st.q] + [3] * st.d))
    tau = 0.5003
    f_kbc = lt.KBCCollision(tau)(flow)
    f_bgk = lt.BGKCollision(flow.units.relaxation_parameter_lu)(flow)   # same tau as KBC uses
    assert (flow.pseudo_entropy_local(f_bgk) <= flow.pseudo_entropy_local(f_kbc) + 1e-6).all()


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("case", ["tgv2d", "tgv3d", "shear2d"])
def test_initialize_fneq_keeps_moments_and_improves_tgv(cfg, case):
    ctx = context(cfg)
    make = {"tgv2d": lambda **k: lt.TaylorGreenVortex(ctx, [32, 32], 1000, 0.1, lt.D2Q9(), **k),
            "tgv3d": lambda **k: lt.TaylorGreenVortex(ctx, [16] * 3, 1000, 0.1, lt.D3Q27(), **k),
            "shear2d": lambda **k: lt.DoublyPeriodicShear2D(ctx, 32, 1000, 0.1, lt.D2Q9(), **k)}[case]
    with_neq, without = make(), make(initialize_fneq=False)
    tol = 1e-6
    for a, b in ((with_neq.rho(), without.rho()), (with_neq.u(), without.u()),
                 (with_neq.incompressible_energy(), without.incompressible_energy())):
        assert a.cpu().numpy() == pytest.approx(b.cpu().numpy(), rel=0.0, abs=tol)
    if case == "tgv2d":
        errors = []
        for flow in (with_neq, without):
            rep = lt.ErrorReporter(flow.analytic_solution, interval=1, out=None)
            lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [rep])(10)
            errors.append(np.mean(np.abs(rep.out), axis=0)[0])
        assert errors[0] < errors[1]


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("observable", [lt.Enstrophy, lt.MaximumVelocity, lt.IncompressibleKineticEnergy, lt.Mass])
@pytest.mark.parametrize("res", [[32] * 2, [16] * 3], ids=["2d", "3d"])
def test_observables_change_little_in_two_steps(cfg, observable, res):
    flow = lt.TaylorGreenVortex(context(cfg), res, 10000, 0.05)
    rep = quiet(lt.ObservableReporter, observable(flow), interval=1, out=None)
    lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [rep])(2)
    values = np.asarray(rep.out)
    assert values.shape[0] == 3 and values[1, 2] == pytest.approx(values[0, 2], rel=0.05)


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("name", sorted(lt.flow_by_name))
def test_registered_flows_run_one_step(cfg, name):
    flow_class, stencil = lt.flow_by_name[name]
    flow = flow_class(context(cfg), 16, 1, 0.05, stencil)
    sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
    sim(1)
    assert flow.i == 1 and torch.isfinite(flow.f).all()


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("stencil,res", [(lt.D2Q9, [16, 16]), (lt.D3Q27, [16, 16, 16])], ids=["2d", "3d"])
def test_obstacle_masks_and_two_steps(cfg, stencil, res):
    """tests/boundary/test_bc_masks.py + tests/flow/test_obstacle.py"""
    flow = lt.Obstacle(context(cfg), res, 100, 0.1, domain_length_x=2, stencil=stencil())
    mask = np.zeros(res, dtype=bool)
    mask[(slice(6, 9),) * len(res)] = True
    flow.mask = mask
    flow.initialize()
    sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
    assert sim.no_streaming_mask.any() and sim.no_collision_mask.any()
    assert int((sim.no_collision_mask == 2).sum()) == int(mask.sum())          # bounce-back index
    sim(2)
    assert torch.isfinite(flow.f).all()
