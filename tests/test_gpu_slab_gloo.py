"""Two ranks sharing the one GPU of the test box, gloo backend: the real HIP slab kernels and the
real point-to-point ghost-plane exchange code (only the transport differs from RCCL, which
cannot put two ranks on one device) must reproduce the single-domain oracle."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, res, steps, dtype_name, overlap, out_dir, driver="SlabSimulation", signalled=False,
            transport="rccl", copy_streams=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import lettuce_amd as lt
    ctx = lt.Context("cuda:0", getattr(torch, dtype_name), use_native=True)
    slab = lt.ZSlab(res)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
    kwargs = {"signalled": True} if signalled else {}
    if copy_streams is not None:
        kwargs["copy_streams"] = copy_streams
    sim = getattr(lt, driver)(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab,
                              overlap=overlap, transport=transport, **kwargs)
    sim(steps)
    if transport == "copy":
        assert sim._cw is not None and sim._cw.count > 0 and not sim._cw.timed_out()
        assert sim._cw.n_streams == (copy_streams or 1)
        if driver == "TwoStepSlabSimulation":
            assert sim._direct_ok()
    if signalled:
        assert sim._signalled_ok() and not sim.engine.wait_timed_out()
    f1 = sim.gather_f()
    ke = sim.kinetic_energy_pu()
    ens, mass = sim.enstrophy_pu(), sim.mass_interior(None)
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), f1=f1.cpu().numpy(), ke=ke, ens=ens, mass=mass)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, True), (2, False), (4, True)],
                         ids=["2ranks-overlap", "2ranks-serial", "4ranks-overlap"])
def test_ranks_sharing_one_gpu(tmp_path, world, overlap):
    """2 ranks: both neighbours are the same peer; 4 ranks: distinct lower / upper neighbours."""
    from oracle import lettuce_oracle as orc
    res, steps = [32, 16, 16], 6
    port = 29700 + (os.getpid() % 1000) + int(overlap) + 10 * world
    mp.spawn(_worker, args=(world, port, res, steps, "float64", overlap, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", torch.float64)
    ref.step(steps)
    np.testing.assert_allclose(got["f1"], ref.f.numpy(), rtol=0, atol=1e-13)
    assert float(got["ke"]) == pytest.approx(float(orc.kinetic_energy_pu(ref.f, ref.lat, ref.units)), rel=1e-11)
    # the slab observables: velocity planes swapped between the ranks (staged through the host for gloo), device sums
    assert float(got["ens"]) == pytest.approx(float(orc.enstrophy_pu(ref.f, ref.lat, ref.units)), rel=1e-11)
    assert float(got["mass"]) == pytest.approx(float(orc.mass_observable(ref.f, None)), rel=1e-12)


@pytest.mark.parametrize("world,overlap,steps,dtype_name,signalled",
                         [(2, True, 7, "float32", False), (2, False, 6, "float32", False), (3, True, 4, "float32", False),
                          (2, True, 5, "float64", False), (2, True, 8, "float32", True), (3, True, 5, "float64", True)],
                         ids=["2ranks-overlap-7steps", "2ranks-serial-6steps", "3ranks-overlap-4steps", "2ranks-fp64",
                              "2ranks-signalled", "3ranks-signalled-fp64"])
def test_two_step_slab_ranks_sharing_one_gpu(tmp_path, world, overlap, steps, dtype_name, signalled):
    """TwoStepSlabSimulation with the real kernels (lt_stream_collide_twice_planes on slabs with two
    ghost planes, lt_slab_pack/unpack_two_step) on 2-3 gloo ranks sharing the GPU, fp32, against the
    single-domain oracle; boundary / interior launches on two streams when overlapping."""
    from oracle import lettuce_oracle as orc
    res = [64, 16, 12 * world]
    port = 29600 + (os.getpid() % 1000) + int(overlap) + 10 * world
    port += 100 if signalled else 0
    mp.spawn(_worker, args=(world, port, res, steps, dtype_name, overlap, str(tmp_path), "TwoStepSlabSimulation", signalled),
             nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", getattr(torch, dtype_name))
    ref.step(steps)
    tol = 1e-5 if dtype_name == "float32" else 1e-13
    np.testing.assert_allclose(got["f1"], ref.f.numpy(), rtol=0, atol=tol * float(np.abs(ref.f.numpy()).max()))


@pytest.mark.parametrize("driver,world,steps,dtype_name,streams",
                         [("TwoStepSlabSimulation", 2, 9, "float32", None), ("TwoStepSlabSimulation", 3, 6, "float64", None),
                          ("TwoStepSlabSimulation", 4, 8, "float32", None), ("SlabSimulation", 2, 5, "float64", None),
                          ("SlabSimulation", 3, 4, "float32", None), ("TwoStepSlabSimulation", 3, 7, "float32", 2)],
                         ids=["two-step-2ranks", "two-step-3ranks-fp64", "two-step-4ranks", "single-step-2ranks-fp64",
                              "single-step-3ranks", "two-step-3ranks-a-stream-per-direction"])
def test_copy_transport_between_processes_sharing_one_gpu(tmp_path, driver, world, steps, dtype_name, streams):
    """transport="copy" as the ranks of a node use it: every process allocates its receive window with lt_ipc_alloc,
    the 64-byte handles travel through the process group, every rank maps its neighbours' windows (hipIpcOpenMemHandle)
    and the halo messages move by device-to-device copies without compute units, each followed by a counter the
    receiver's polling wave waits for -- real kernels, real inter-process mapping, 2-4 ranks on the one GPU of the
    box, odd and even step counts, against the single-domain oracle; one case with a stream per direction
    (``copy_streams=2``, bench.py's second copy candidate at N > 1)."""
    from oracle import lettuce_oracle as orc
    res = [64, 16, 12 * world]
    port = 29300 + (os.getpid() % 500) + 10 * world + (5 if driver == "SlabSimulation" else 0)
    mp.spawn(_worker, args=(world, port + (3 if streams else 0), res, steps, dtype_name, True, str(tmp_path), driver, False,
                            "copy", streams), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", getattr(torch, dtype_name))
    ref.step(steps)
    tol = 1e-5 if dtype_name == "float32" else 1e-13
    np.testing.assert_allclose(got["f1"], ref.f.numpy(), rtol=0, atol=tol * float(np.abs(ref.f.numpy()).max()))


def _bench_loop_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import contextlib
    import io
    import bench
    import lettuce_amd as lt
    args = bench.parse(["--gpus", str(world), "--steps", "6", "--warmup", "3", "--batches", "2", "--candidate-budget", "120",
                        "--first-candidate-budget", "120"])
    bench.MIN_BATCH_S = 0.0
    ctx = lt.Context("cuda:0", torch.float32, use_native=True)
    res = [64, 16, 12 * world]
    slab = lt.ZSlab(res)

    def build(driver, transport):
        flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
        coll = lt.BGKCollision(flow.units.relaxation_parameter_lu)
        if driver == "two-step":
            return lt.TwoStepSlabSimulation(flow, coll, slab, transport=transport.split("-")[0], direct=True,
                                            copy_streams=2 if transport == "copy-2streams" else None)
        return lt.SlabSimulation(flow, coll, slab, transport=transport)

    wanted = [("single-step", "rccl"), ("two-step", "rccl"), ("two-step", "copy"), ("two-step", "copy-2streams")]
    ranks = bench.Ranks(dist, world, rank, 0, torch.device("cpu"))      # the loop's own collectives through gloo on the host
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        bench.candidate_loop(args, ranks, wanted, build, "TGV3D D3Q19 fp32 on ranks sharing one GPU", res,
                             res[0] * res[1] * slab.nz_local, 152, "f32", probe_steps=7,
                             cpu_baseline_row={"value": 1.0, "unit": "MLUPS", "cores": 1, "kind": "port", "sample": "stub"}
                             if rank == 0 else None)
    if rank == 0:
        with open(os.path.join(out_dir, "line.json"), "w") as fh:
            fh.write(out.getvalue())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_candidate_loop_with_the_real_kernels_on_ranks_sharing_one_gpu(tmp_path, world):
    """bench.py's N > 1 loop as the driver will run it -- reference candidate, two-step driver over the process group,
    two-step driver over the copy transport with one stream and with a stream per direction -- with the real kernels, 2-3 processes that share the GPU of the box, the
    copy transport's windows mapped between the processes (HIP IPC), odd probe length: every candidate must be
    bit-identical to the reference after the probe and after the timed batches, nothing may fail, and the line must
    carry what the judge asks of an N > 1 line."""
    import json
    port = 29100 + (os.getpid() % 500) + 7 * world
    mp.spawn(_bench_loop_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    lines = [ln for ln in open(tmp_path / "line.json").read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    t = line["config"]["transport"]
    assert line["n_gpus"] == world and line["value"] > 0 and t["failures"] == {}, t
    assert set(t["warmup_ms_per_step"]) == {"single-step/rccl", "two-step/rccl", "two-step/copy", "two-step/copy-2streams"}
    for name in ("two-step/rccl", "two-step/copy", "two-step/copy-2streams"):
        assert "bit-identical" in t["checks"][name] and "after the timed batches too" in t["checks"][name], t["checks"]
    assert t["ranks_seen"]["ranks"] == world and len(t["rank_checksums"]) == world
    assert line["cpu_baseline"]["kind"] == "port" and "traffic" in line["roofline"]


def _two_step_identity_worker(rank, port, res, steps, transport, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if transport.startswith("rccl"):
        os.environ["LT_SLAB_FORCE_P2P"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    import lettuce_amd as lt
    ctx = lt.Context("cuda:0", torch.float32, use_native=True)
    outs = []
    timed_out = False
    for driver in ("SlabSimulation", "TwoStepSlabSimulation"):
        slab = lt.ZSlab(res)
        flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
        kwargs = {"transport": transport.split("-")[0]}
        if transport.endswith("-signalled"):
            if driver == "SlabSimulation":
                kwargs = {"transport": "rccl"}
            else:
                kwargs["signalled"] = True
        sim = getattr(lt, driver)(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab, **kwargs)
        sim(steps)
        outs.append(sim.gather_f().clone())
        if kwargs.get("signalled"):
            assert sim._signalled_ok()
            timed_out = sim.engine.wait_timed_out()
    np.savez(os.path.join(out_dir, "out.npz"), same=bool(torch.equal(outs[0], outs[1])), timed_out=timed_out)
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["rccl", "window", "rccl-signalled", "copy"])
def test_two_step_slab_is_bit_identical_to_the_single_step_slab(tmp_path, transport):
    """Same kernels' arithmetic, different schedule and halo: the two drivers must agree bit for bit
    (single rank exchanging with itself through RCCL / through its own peer window).  "rccl-signalled": one
    launch per double step whose edge workgroups run first, write the messages and release the communication
    stream through a device counter (lt_stream_collide_twice_slab + lt_slab_wait_messages); its polling wave
    must never have given up."""
    mp.spawn(_two_step_identity_worker, args=(29950 + os.getpid() % 1000, [64, 32, 16], 9, transport, str(tmp_path)),
             nprocs=1, join=True)
    got = np.load(tmp_path / "out.npz")
    assert bool(got["same"]) and not bool(got["timed_out"])


def _golden_slab_worker(rank, port, res, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    import lettuce_amd as lt
    ctx = lt.Context("cuda:0", torch.float32, use_native=True)
    from conftest import golden
    f0 = torch.as_tensor(golden("tgv3d_d3q19_bgk_64x8x12_f32")["f0"]).cuda()
    out = {}
    for n in steps:
        slab = lt.ZSlab(res)
        flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
        out.setdefault("f0_device", flow.f[..., slab.halo:-slab.halo].cpu().numpy())
        # start from the reference's own initial populations (periodic extension by the slab's halo planes)
        flow.f = torch.cat([f0[..., -slab.halo:], f0, f0[..., :slab.halo]], dim=-1).contiguous()
        sim = lt.TwoStepSlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
        sim(n)
        out[f"f{n}"] = sim.gather_f().cpu().numpy()
    np.savez(os.path.join(out_dir, "out.npz"), **out)
    dist.destroy_process_group()


def test_two_step_slab_driver_reproduces_the_reference_vectors_on_one_rank(tmp_path):
    """TwoStepSlabSimulation (slab layout, two ghost planes, lt_stream_collide_twice_planes + halo messages to
    itself) against vectors of the reference's CPU path on a grid its tiles take: bit-identical."""
    from conftest import golden
    g = golden("tgv3d_d3q19_bgk_64x8x12_f32")
    mp.spawn(_golden_slab_worker, args=(29850 + os.getpid() % 1000, [64, 8, 12], (2, 9, 10), str(tmp_path)),
             nprocs=1, join=True)
    got = np.load(tmp_path / "out.npz")
    np.testing.assert_allclose(got["f0_device"], g["f0"], rtol=0, atol=1e-6)     # device-side initial condition
    for n in (2, 9, 10):
        np.testing.assert_array_equal(got[f"f{n}"], g[f"f{n}"])


def _rccl_worker(rank, port, res, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LT_SLAB_FORCE_P2P"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    import lettuce_amd as lt
    ctx = lt.Context("cuda:0", torch.float32, use_native=True)
    slab = lt.ZSlab(res)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
    sim = lt.SlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
    assert sim._force_p2p and not sim._host_transport
    sim(steps)
    np.savez(os.path.join(out_dir, "out.npz"), f1=sim.gather_f().cpu().numpy(), ke=sim.kinetic_energy_pu())
    dist.destroy_process_group()


def test_rccl_point_to_point_path_with_self_exchange(tmp_path):
    """The RCCL (backend 'nccl') send/recv path of the slab driver -- batch_isend_irecv on the
    communication stream, overlapped with the interior kernel -- exercised on one GPU by letting
    the single rank send its ghost planes to itself through the process group."""
    from oracle import lettuce_oracle as orc
    res, steps = [64, 32, 16], 8
    mp.spawn(_rccl_worker, args=(29800 + os.getpid() % 1000, res, steps, str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "out.npz")
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", torch.float32)
    ref.step(steps)
    np.testing.assert_allclose(got["f1"], ref.f.numpy(), rtol=0, atol=1e-5 * float(np.abs(ref.f.numpy()).max()))


def _window_worker(rank, port, res, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    import lettuce_amd as lt
    ctx = lt.Context("cuda:0", torch.float64, use_native=True)
    slab = lt.ZSlab(res)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 400, 0.1, lt.D3Q19(), slab=slab)
    sim = lt.SlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab,
                            transport="window")
    assert sim._window is not None
    sim(3)              # odd and even batch lengths: the window parity runs through both values
    sim(steps - 3)
    np.savez(os.path.join(out_dir, "out.npz"), f1=sim.gather_f().cpu().numpy(), ke=sim.kinetic_energy_pu(),
             exchanges=sim._window.count)
    dist.destroy_process_group()


def test_window_transport_with_self_exchange(tmp_path):
    """The one-sided transport (stores into the neighbour's receive window + signal pads, torch
    symmetric memory) on one GPU: the single rank is its own lower and upper neighbour, so the
    boundary-plane launch stores into its own window through the peer mapping."""
    from oracle import lettuce_oracle as orc
    res, steps = [64, 32, 16], 9
    mp.spawn(_window_worker, args=(29900 + os.getpid() % 1000, res, steps, str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "out.npz")
    assert int(got["exchanges"]) == steps      # one exchange per step (1 after collide + n-1 fused, per batch)
    ref = orc.taylor_green(res, 400, 0.1, "D3Q19", torch.float64)
    ref.step(steps)
    np.testing.assert_allclose(got["f1"], ref.f.numpy(), rtol=0, atol=1e-13)


def _obstacle_worker(rank, world, port, name, steps, dtype_name, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import lettuce_amd as lt
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import golden
    g = golden(name)
    ctx = lt.Context("cuda:0", getattr(torch, dtype_name), use_native=True)
    res = [int(r) for r in g["resolution"]]
    slab = lt.ZSlab(res)
    flow = lt.Obstacle(ctx, slab.extended_resolution, 100, 0.1, float(g["domain_length_x"]),
                       stencil=lt.D3Q27(), slab=slab)
    flow.mask = torch.tensor(g["obstacle_mask"])[:, :, slab.z_indices()]
    flow.initialize()
    sim = lt.SlabSimulation(flow, lt.KBCCollision(), slab)
    sim(steps)
    f1 = sim.gather_f()
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), f1=f1.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name,dtype_name,atol", [(1, "obstacle3d_d3q27_kbc_f64", "float64", 1e-11),
                                                        (2, "obstacle3d_d3q27_kbc_f64", "float64", 1e-11),
                                                        (3, "obstacle3d_d3q27_kbc_f32", "float32", 1e-5)])
def test_obstacle_on_slabs_with_the_hip_engine(tmp_path, world, name, dtype_name, atol):
    """Inlet + ABB outlet + sphere bounce-back on z-slabs (masked slab-layout kernels, ghost
    exchange of bounce-back nodes' populations included) against the reference vectors."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import golden
    port = 29900 + (os.getpid() % 1000) + world
    mp.spawn(_obstacle_worker, args=(world, port, name, 8, dtype_name, str(tmp_path)), nprocs=world, join=True)
    g, got = golden(name), np.load(tmp_path / "out.npz")
    np.testing.assert_allclose(got["f1"], g["f8"], rtol=0, atol=atol * float(np.abs(g["f8"]).max()))


def _obstacle_two_step_worker(rank, world, port, name, lattice, snaps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import lettuce_amd as lt
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import golden
    g = golden(name)
    ctx = lt.Context("cuda:0", torch.float32, use_native=True)
    res = [int(r) for r in g["resolution"]]
    out = {}
    for n in snaps:
        for driver in ("SlabSimulation", "TwoStepSlabSimulation"):
            slab = lt.ZSlab(res)
            flow = lt.Obstacle(ctx, slab.extended_resolution, 100, 0.1, float(g["domain_length_x"]),
                               stencil=getattr(lt, lattice)(), slab=slab)
            flow.mask = torch.tensor(g["obstacle_mask"])[:, :, slab.z_indices()]
            flow.initialize()
            sim = getattr(lt, driver)(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
            sim(n)
            f = sim.gather_f()
            if rank == 0:
                out[f"{driver}_{n}"] = f.cpu().numpy()
            if driver == "TwoStepSlabSimulation" and rank == 0:
                out["kernel"] = sim.engine.kernel_name()
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name,lattice", [(1, "obstacle3d_d3q19_bgk_64x8x16_f32", "D3Q19"),
                                                (2, "obstacle3d_d3q19_bgk_64x8x16_f32", "D3Q19"),
                                                (4, "obstacle3d_d3q27_bgk_64x8x16_f32", "D3Q27")])
def test_obstacle_on_slabs_with_two_updates_per_launch(tmp_path, world, name, lattice):
    """VERDICT r01 item 4, second half: boundary flows on slabs through the two-step kernel.  The reference's
    Obstacle on z-slabs in the slab layout (x contiguous, so the outlet's normal is the contiguous axis:
    lbm2m_kernel AX = 0 with two ghost planes), one halo message per double step that also carries what the
    no-streaming nodes of the ghost planes keep: bit-identical to the one-exchange-per-step slab driver for
    odd and even step counts, and equal to the reference's populations at the outlet's rounding level."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import golden
    snaps = (1, 2, 3, 8)
    port = 29400 + (os.getpid() % 1000) + world
    mp.spawn(_obstacle_two_step_worker, args=(world, port, name, lattice, snaps, str(tmp_path)), nprocs=world, join=True)
    g, got = golden(name), np.load(tmp_path / "out.npz")
    assert "lbm2m_kernel" in str(got["kernel"]) and ", 0>" in str(got["kernel"])
    for n in snaps:
        np.testing.assert_array_equal(got[f"TwoStepSlabSimulation_{n}"], got[f"SlabSimulation_{n}"])
        np.testing.assert_allclose(got[f"TwoStepSlabSimulation_{n}"], g[f"f{n}"], rtol=0,
                                   atol=1e-5 * float(np.abs(g[f"f{n}"]).max()))
