"""N > 1 path on CPU: world_size-2 (and 4) gloo processes run the z-slab driver with an
oracle-backed engine and must reproduce the single-domain oracle exactly; also checks that the
slab-aware initial condition equals the global one plane for plane."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, res, lattice, collision, steps, dtype_name, out_dir, driver="SlabSimulation",
            direct="1"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LT_SLAB_DIRECT"] = direct
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    dtype = getattr(torch, dtype_name)
    ctx = lt.Context("cpu", dtype, use_native=False)
    slab = lt.ZSlab(res)
    stencil = getattr(lt, lattice)()
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, stencil, slab=slab)
    coll = lt.BGKCollision(flow.units.relaxation_parameter_lu) if collision == "bgk" else lt.KBCCollision()
    engine = OracleSlabEngine(lattice, dtype, collision)
    sim = getattr(lt, driver)(flow, coll, slab, engine=engine)
    f0 = sim.gather_f()
    sim(steps)
    f1 = sim.gather_f()
    ke = sim.kinetic_energy_pu()
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), f0=f0.numpy(), f1=f1.numpy(), ke=ke)
    dist.barrier()
    dist.destroy_process_group()


CASES = [(2, [8, 6, 8], "D3Q19", "bgk", 5, "float64"),
         (4, [6, 8, 8], "D3Q27", "kbc", 3, "float64"),
         (2, [8, 8, 4], "D3Q19", "bgk", 4, "float32")]


@pytest.mark.parametrize("world,res,lattice,collision,steps,dtype_name", CASES,
                         ids=[f"{c[0]}ranks-{c[2]}-{c[3]}-{c[5]}" for c in CASES])
def test_slab_ranks_reproduce_single_domain(tmp_path, world, res, lattice, collision, steps, dtype_name):
    from oracle import lettuce_oracle as orc
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, res, lattice, collision, steps, dtype_name, str(tmp_path)),
             nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    dtype = getattr(torch, dtype_name)
    ref = orc.taylor_green(res, 400, 0.1, lattice, dtype, collision)
    tol = 1e-13 if dtype_name == "float64" else 1e-6
    # slab init == global init (torch's vectorised sin/cos may differ in the last bit between
    # tensor shapes, hence not assert_array_equal)
    np.testing.assert_allclose(got["f0"], ref.f.numpy(), rtol=0, atol=tol / 50)
    ref.step(steps)
    np.testing.assert_allclose(got["f1"], ref.f.numpy(), rtol=0, atol=tol)
    ke_ref = float(orc.kinetic_energy_pu(ref.f, ref.lat, ref.units))
    assert float(got["ke"]) == pytest.approx(ke_ref, rel=1e-12 if dtype_name == "float64" else 1e-5)


TWO_STEP_CASES = [(2, [8, 6, 16], 6), (2, [8, 6, 16], 5), (4, [8, 6, 16], 4), (3, [6, 4, 12], 7)]


@pytest.mark.parametrize("direct", ["1", "0"], ids=["direct", "edges-beside-interior"])
@pytest.mark.parametrize("world,res,steps", TWO_STEP_CASES, ids=[f"{c[0]}ranks-{c[2]}steps" for c in TWO_STEP_CASES])
def test_two_step_slab_ranks_reproduce_single_domain(tmp_path, world, res, steps, direct):
    """The two-step slab driver (two ghost planes, one 19-block message per direction per double
    step, odd and even numbers of fused steps) on 2-4 gloo ranks against the single-domain oracle, with both
    schedules of a double step: "direct" (edge launch fed from the receive buffers and writing the outgoing messages,
    then the planes in between) and the edge launches + pack / unpack of rounds 1-2."""
    from oracle import lettuce_oracle as orc
    port = 29100 + (os.getpid() % 2000) + world + steps + 7 * int(direct)
    mp.spawn(_worker, args=(world, port, res, "D3Q19", "bgk", steps, "float64", str(tmp_path),
                            "TwoStepSlabSimulation", direct), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", torch.float64, "bgk")
    np.testing.assert_allclose(got["f0"], ref.f.numpy(), rtol=0, atol=2e-15)
    ref.step(steps)
    np.testing.assert_allclose(got["f1"], ref.f.numpy(), rtol=0, atol=1e-13)
    assert float(got["ke"]) == pytest.approx(float(orc.kinetic_energy_pu(ref.f, ref.lat, ref.units)), rel=1e-12)


@pytest.mark.parametrize("direct", [True, False], ids=["direct", "edges-beside-interior"])
def test_two_step_slab_single_rank_and_batches(direct):
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    from oracle import lettuce_oracle as orc
    res = [6, 4, 5]
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    slab = lt.ZSlab(res, rank=0, world_size=1)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 100, 0.1, lt.D3Q19(), slab=slab)
    sim = lt.TwoStepSlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab,
                                   engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"), direct=direct)
    assert sim._direct_ok() == direct
    ref = orc.taylor_green(res, 100, 0.1, "D3Q19", torch.float64)
    sim(1); sim(2); sim(3); sim(4)
    ref.step(10)
    np.testing.assert_allclose(sim.gather_f().numpy(), ref.f.numpy(), rtol=0, atol=1e-13)
    # looking in between (the presentation pass scatters the messages the direct schedule left in the receive
    # buffers), an odd batch that carries on, then an even one
    sim(5); _ = sim.local_f(); sim(3); sim(4)
    ref.step(12)
    np.testing.assert_allclose(sim.gather_f().numpy(), ref.f.numpy(), rtol=0, atol=1e-13)


def test_single_rank_slab_self_exchange():
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    from oracle import lettuce_oracle as orc
    res = [6, 4, 5]
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    slab = lt.ZSlab(res, rank=0, world_size=1)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 100, 0.1, lt.D3Q19(), slab=slab)
    sim = lt.SlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab,
                            engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"))
    ref = orc.taylor_green(res, 100, 0.1, "D3Q19", torch.float64)
    np.testing.assert_allclose(sim.gather_f().numpy(), ref.f.numpy(), rtol=0, atol=2e-15)
    sim(3); sim(2)
    ref.step(5)
    np.testing.assert_allclose(sim.gather_f().numpy(), ref.f.numpy(), rtol=0, atol=1e-13)


def _obstacle_worker(rank, world, port, name, steps, out_dir, driver="SlabSimulation"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    from conftest import golden
    g = golden(name)
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    res = [int(r) for r in g["resolution"]]
    slab = lt.ZSlab(res)
    flow = lt.Obstacle(ctx, slab.extended_resolution, 100, 0.1, float(g["domain_length_x"]),
                       stencil=lt.D3Q27(), slab=slab)
    flow.mask = torch.tensor(g["obstacle_mask"])[:, :, slab.z_indices()]
    flow.initialize()
    sim = getattr(lt, driver)(flow, lt.KBCCollision(), slab, engine=OracleSlabEngine("D3Q27", torch.float64, "kbc"))
    f0 = sim.gather_f()
    sim(steps)
    f1 = sim.gather_f()
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), f0=f0.numpy(), f1=f1.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,driver", [(2, "SlabSimulation"), (3, "SlabSimulation"), (2, "TwoStepSlabSimulation"),
                                          (3, "TwoStepSlabSimulation")])
def test_slab_ranks_with_obstacle_boundaries(tmp_path, world, driver):
    """Obstacle (equilibrium inlet, ABB outlet along x, sphere bounce-back) on z-slabs: masks are
    built per rank on its planes of the global grid and the result equals the reference's -- with one
    exchange per step and with the two-step driver (two ghost planes, one 36-block message per double step:
    with boundaries it also carries the populations a no-streaming node of the ghost plane keeps)."""
    from conftest import golden
    name, steps = "obstacle3d_d3q27_kbc_f64", 8
    port = 29300 + (os.getpid() % 2000) + world + (10 if driver != "SlabSimulation" else 0)
    mp.spawn(_obstacle_worker, args=(world, port, name, steps, str(tmp_path), driver), nprocs=world, join=True)
    g, got = golden(name), np.load(tmp_path / "out.npz")
    np.testing.assert_allclose(got["f0"], g["f0"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(got["f1"], g["f8"], rtol=0, atol=1e-12)


def _reporter_worker(rank, world, port, res, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import io
    import contextlib
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    slab = lt.ZSlab(res)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
    out = []
    with contextlib.redirect_stdout(io.StringIO()):
        rep = lt.ObservableReporter(lt.SlabKineticEnergy(flow), interval=3, out=out)
        bad = lt.ObservableReporter(lt.IncompressibleKineticEnergy(flow), interval=3, out=[])
    coll = lt.BGKCollision(flow.units.relaxation_parameter_lu)
    refused = False
    try:
        lt.SlabSimulation(flow, coll, slab, reporter=[bad], engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"))
    except lt.LettuceException:
        refused = True
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
    sim = lt.SlabSimulation(flow, coll, slab, reporter=[rep], engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"))
    sim(steps)
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), series=np.array(out, dtype=np.float64), refused=refused,
                 flow_i=flow.i)
    dist.barrier()
    dist.destroy_process_group()


def test_slab_driver_with_an_observable_reporter(tmp_path):
    """Reporters on slabs: flow.i follows the driver, an ObservableReporter around the
    slab-aware kinetic energy logs the all-reduced series of the whole domain, and a reporter whose
    observable reads flow.f is refused with a LettuceException instead of failing on flow.f = None."""
    from oracle import lettuce_oracle as orc
    res, steps = [8, 6, 8], 7
    mp.spawn(_reporter_worker, args=(2, 29700 + os.getpid() % 2000, res, steps, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "out.npz")
    assert bool(got["refused"]) and int(got["flow_i"]) == steps
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", torch.float64)
    want = {0: float(orc.kinetic_energy_pu(ref.f, ref.lat, ref.units))}
    for i in range(1, steps + 1):
        ref.step(1)
        if i % 3 == 0:
            want[i] = float(orc.kinetic_energy_pu(ref.f, ref.lat, ref.units))
    series = got["series"]
    assert [int(r[0]) for r in series] == sorted(want)
    for row in series:
        assert row[2] == pytest.approx(want[int(row[0])], rel=1e-12)
        assert row[1] == pytest.approx(ref.units.time_to_pu(int(row[0])), rel=1e-12)


def test_slab_halo_narrower_than_the_ghost_planes_is_refused():
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    slab = lt.ZSlab([6, 4, 8], rank=0, world_size=1, halo=1)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 100, 0.1, lt.D3Q19(), slab=slab, initialize_fneq=False)
    with pytest.raises(lt.LettuceException, match="ghost planes"):
        lt.TwoStepSlabSimulation(flow, lt.BGKCollision(0.6), slab, engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"))


def _field_inlet_setup(lt, ctx, res, slab=None):
    """A periodic TGV with an equilibrium boundary whose velocity / pressure vary from node to node on a
    block of nodes (per-node arguments, equilibrium_boundary_pu.py:21-40), built on the whole grid or on
    a rank's extended slab (the arguments are then this rank's planes of the same global fields)."""
    torch.manual_seed(17)
    vel = 0.05 * torch.rand([3] + res, dtype=torch.float64)
    prs = 0.01 * torch.rand(res, dtype=torch.float64)
    mask = torch.zeros(res, dtype=torch.bool)
    mask[2:5, 1:4, :] = True                                    # crosses every cut along z
    if slab is not None:
        z = slab.z_indices()
        vel, prs, mask = vel[..., z], prs[..., z], mask[..., z]

    class Forced(lt.TaylorGreenVortex):
        extra = None

        @property
        def boundaries(self):
            return [] if self.extra is None else [self.extra]
    flow = Forced(ctx, slab.extended_resolution if slab is not None else res, 400, 0.1, lt.D3Q19(), slab=slab)
    flow.extra = lt.EquilibriumBoundaryPU(ctx, mask, vel, prs)
    return flow


def _field_worker(rank, world, port, res, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    slab = lt.ZSlab(res)
    flow = _field_inlet_setup(lt, ctx, res, slab)
    sim = lt.SlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab,
                            engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"))
    sim(steps)
    f1 = sim.gather_f()
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), f1=f1.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_ranks_with_per_node_equilibrium_boundary_arguments(tmp_path, world):
    """Per-node inlet fields on slabs: every rank slices the field to its planes;
    the result equals the single-domain run of the mirror's non-native (reference) path."""
    import lettuce_amd as lt
    res, steps = [8, 6, 12], 5
    mp.spawn(_field_worker, args=(world, 29800 + os.getpid() % 2000 + world, res, steps, str(tmp_path)),
             nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    flow = _field_inlet_setup(lt, ctx, res)
    lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])(steps)
    np.testing.assert_allclose(got["f1"], flow.f.numpy(), rtol=0, atol=1e-13)


def _z_channel(lt, ctx, res, slab=None):
    """Flow along +z: equilibrium inlet on the plane z = 0, anti-bounce-back outlet on z = nz - 1, a
    bounce-back block in between; on the whole grid or on a rank's extended slab."""
    zsel = slab.z_indices() if slab is not None else torch.arange(res[2])
    gz = zsel.reshape(1, 1, -1).expand(res[0], res[1], -1)
    block = torch.zeros(res, dtype=torch.bool)
    block[2:4, 1:3, 4:7] = True
    if slab is not None:
        block = block[..., zsel]

    class Channel(lt.TaylorGreenVortex):
        @property
        def boundaries(self):
            return [lt.EquilibriumBoundaryPU(self.context, gz == 0, [0.0, 0.0, 0.05]),
                    lt.AntiBounceBackOutlet([0, 0, 1], self), lt.BounceBackBoundary(block)]
    return Channel(ctx, slab.extended_resolution if slab is not None else res, 100, 0.05, lt.D3Q19(), slab=slab)


def _z_channel_two_step_worker(rank, world, port, res, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    slab = lt.ZSlab(res)
    flow = _z_channel(lt, ctx, res, slab)
    refused = ""
    try:
        lt.TwoStepSlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab,
                                 engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"))
    except lt.LettuceException as e:
        refused = str(e)
    with open(os.path.join(out_dir, f"refused{rank}.txt"), "w") as fh:
        fh.write(refused)
    dist.barrier()
    dist.destroy_process_group()


def test_two_step_slab_driver_refuses_an_outlet_along_z_on_every_rank(tmp_path):
    """The two-step launches take outlets along x only; the ranks agree (all-reduce) before any of them steps, so
    no rank is left waiting in an exchange for one that raised."""
    mp.spawn(_z_channel_two_step_worker, args=(2, 29950 + os.getpid() % 2000, [6, 5, 12], str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert "two-step slab driver cannot run this flow" in (tmp_path / f"refused{rank}.txt").read_text()


def _z_channel_worker(rank, world, port, res, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    slab = lt.ZSlab(res)
    flow = _z_channel(lt, ctx, res, slab)
    sim = lt.SlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab,
                            engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"))
    sim(steps)
    f1 = sim.gather_f()
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), f1=f1.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_slab_ranks_with_an_outlet_along_the_decomposed_axis(tmp_path, world):
    """An anti-bounce-back outlet along z on z-slabs: only the rank that holds the last
    plane of the global grid has the outlet; the result equals the single-domain run."""
    import lettuce_amd as lt
    res, steps = [6, 5, 12], 6
    mp.spawn(_z_channel_worker, args=(world, 29900 + os.getpid() % 2000 + world, res, steps, str(tmp_path)),
             nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    flow = _z_channel(lt, ctx, res)
    lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])(steps)
    np.testing.assert_allclose(got["f1"], flow.f.numpy(), rtol=0, atol=1e-13)


def _observables_worker(rank, world, port, res, steps, driver, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import io
    import contextlib
    import lettuce_amd as lt
    from slab_cpu_engine import OracleSlabEngine
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    slab = lt.ZSlab(res)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
    # a no-mass mask given like the flow's own masks: on the extended slab (here: a block that crosses the cut)
    mask = torch.zeros(res, dtype=torch.bool)
    mask[1:4, 2:5, res[2] // 2 - 2:res[2] // 2 + 1] = True
    mask = mask[:, :, slab.z_indices()]
    ens, mass = [], []
    with contextlib.redirect_stdout(io.StringIO()):
        reps = [lt.ObservableReporter(lt.SlabEnstrophy(flow), interval=2, out=ens),
                lt.ObservableReporter(lt.SlabMass(flow, no_mass_mask=mask), interval=2, out=mass)]
    sim = getattr(lt, driver)(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab, reporter=reps,
                              engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"))
    sim(steps)
    plain = sim.mass_interior(None)
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), ens=np.array(ens), mass=np.array(mass), plain=plain)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,driver", [(2, "SlabSimulation"), (3, "SlabSimulation"), (2, "TwoStepSlabSimulation")])
def test_enstrophy_and_mass_observables_on_slabs_need_no_gather(tmp_path, world, driver):
    """SlabEnstrophy / SlabMass (observable_reporter.py:45-68, 140-158) through the slab driver: the ranks swap three
    planes of the velocity field (6th-order differences), reduce their own nodes and all-reduce; the series equals
    the single-domain oracle's.  The mask crosses the cut, the first / last GLOBAL z plane is excluded once."""
    from oracle import lettuce_oracle as orc
    res, steps = [8, 6, 12], 4
    port = 29500 + (os.getpid() % 2000) + world + (20 if driver != "SlabSimulation" else 0)
    mp.spawn(_observables_worker, args=(world, port, res, steps, driver, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", torch.float64)
    mask = torch.zeros(res, dtype=torch.bool)
    mask[1:4, 2:5, res[2] // 2 - 2:res[2] // 2 + 1] = True
    want_e, want_m = {}, {}
    for i in range(steps + 1):
        if i % 2 == 0:
            want_e[i] = float(orc.enstrophy_pu(ref.f, ref.lat, ref.units))
            want_m[i] = float(orc.mass_observable(ref.f, mask))
        ref.step(1)
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", torch.float64)
    ref.step(steps)
    assert [int(r[0]) for r in got["ens"]] == sorted(want_e) == [int(r[0]) for r in got["mass"]]
    for row in got["ens"]:
        assert row[2] == pytest.approx(want_e[int(row[0])], rel=1e-11)
    for row in got["mass"]:
        assert row[2] == pytest.approx(want_m[int(row[0])], rel=1e-12)
    assert float(got["plain"]) == pytest.approx(float(orc.mass_observable(ref.f, None)), rel=1e-12)
