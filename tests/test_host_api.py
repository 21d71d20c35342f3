"""Host-side mirror of the reference interface (CPU, no GPU needed): the non-native mode of
lettuce_amd must give the reference's results (golden vectors), keep its invariants
(tests/collision/*, tests/stencil/*, tests/boundary/* of the reference) and its API contract
(SURVEY.md Appendix C)."""
import io
import os
import re
from copy import copy

import numpy as np
import pytest
import torch

import lettuce_amd as lt
from conftest import golden, unpack_nsm, TORCH_DT, ROOT

STENCILS = [lt.D1Q3, lt.D2Q9, lt.D3Q15, lt.D3Q19, lt.D3Q27]


def ctx(dt="f64"):
    return lt.Context("cpu", TORCH_DT[dt], use_native=False)


class UniformFlow(lt.ExtFlow):
    """The reference's TestFlow (tests/conftest.py:195-232): u = 1.01, p = 0.01 everywhere."""

    def __init__(self, context, resolution, reynolds_number, mach_number, stencil=None, equilibrium=None):
        self._boundaries = []
        super().__init__(context, resolution, reynolds_number, mach_number, stencil, equilibrium)

    def make_resolution(self, resolution, stencil=None):
        if isinstance(resolution, int):
            return [resolution] if stencil is None else [resolution] * stencil.d
        return resolution

    def make_units(self, reynolds_number, mach_number, resolution):
        return lt.UnitConversion(reynolds_number, mach_number, characteristic_length_lu=resolution[0])

    def initial_pu(self):
        return (0.01 * np.ones([1] + self.resolution), 1.01 * np.ones([self.stencil.d] + self.resolution))

    @property
    def boundaries(self):
        return self._boundaries

    @boundaries.setter
    def boundaries(self, b):
        self._boundaries = b


# ------------------------------------------------------------------ stencils (tests/stencil/*)
@pytest.mark.parametrize("cls", STENCILS, ids=[c.__name__ for c in STENCILS])
def test_stencil_properties_and_tables(cls):
    from oracle import lettuce_oracle as orc
    s = cls()
    e, w = np.array(s.e), np.array(s.w)
    assert w.sum() == pytest.approx(1.0)
    assert (e.sum(axis=0) == 0).all() and (e[0] == 0).all()
    assert (e[s.opposite] == -e).all()
    assert s.cs == 1 / np.sqrt(3.0)
    ref = orc.LATTICES[cls.__name__]
    assert [list(v) for v in ref.e] == s.e and list(ref.opposite) == s.opposite
    assert list(ref.w) == pytest.approx(s.w, rel=1e-16)
    assert (s.d, s.q) == (ref.d, ref.q)


def test_csrc_lattice_tables_match_python():
    """lettuce_amd/csrc/lattice.hpp holds the same velocity order as the Python stencils."""
    text = open(os.path.join(ROOT, "lettuce_amd", "csrc", "lattice.hpp")).read()
    for cls in (lt.D1Q3, lt.D2Q9, lt.D3Q15, lt.D3Q19, lt.D3Q27):
        s = cls()
        block = text.split(f"struct {cls.__name__} ")[1].split("static constexpr double W")[0]
        nums = [int(v) for v in re.findall(r"-?\d+", block.split("E[")[1].split("=", 1)[1])]
        table = np.array(nums).reshape(s.q, 3)[:, :s.d]
        assert table.tolist() == s.e
        opp = text.split(f"struct {cls.__name__} ")[1].split("OPP[")[1].split("=", 1)[1].split("}")[0]
        assert [int(v) for v in re.findall(r"\d+", opp)] == s.opposite


# ------------------------------------------------------------------ golden vectors, non-native
TGV = [("tgv2d_d2q9_bgk_32_f64", lt.D2Q9, "bgk", "f64", 10), ("tgv3d_d3q19_bgk_16_f32", lt.D3Q19, "bgk", "f32", 10),
       ("tgv3d_d3q27_kbc_16_f64", lt.D3Q27, "kbc", "f64", 10), ("tgv3d_d3q19_bgk_ragged_f64", lt.D3Q19, "bgk", "f64", 7)]


@pytest.mark.parametrize("name,stencil,coll,dt,n", TGV, ids=[t[0] for t in TGV])
def test_taylor_green_matches_reference(name, stencil, coll, dt, n):
    g = golden(name)
    res = [int(r) for r in g["resolution"]]
    flow = lt.TaylorGreenVortex(ctx(dt), res, float(g["reynolds"]), float(g["mach"]), stencil())
    tol = 2e-14 if dt == "f64" else 2e-6
    np.testing.assert_allclose(flow.f.numpy(), g["f0"], rtol=0, atol=tol)
    collision = lt.BGKCollision(flow.units.relaxation_parameter_lu) if coll == "bgk" else lt.KBCCollision(tau=0.9)
    out = []
    with pytest.MonkeyPatch.context() as mp:
        mp.setattr("sys.stdout", io.StringIO())
        rep = lt.ObservableReporter(lt.IncompressibleKineticEnergy(flow), interval=n, out=out)
    sim = lt.Simulation(flow, collision, [rep])
    mlups = sim(n)
    assert mlups > 0 and flow.i == n
    np.testing.assert_allclose(flow.f.numpy(), g[f"f{n}"], rtol=0, atol=tol * 5)
    assert [row[0] for row in out] == [0, n]
    ref = dict(zip(g["energy_steps"].tolist(), g["energy_pu"].tolist()))
    assert out[0][2] == pytest.approx(ref[0], rel=1e-12 if dt == "f64" else 2e-6)
    assert out[1][2] == pytest.approx(ref[n], rel=1e-12 if dt == "f64" else 2e-6)
    if coll == "kbc":    # the constructor's tau is ignored (kbc_collision.py:97-99)
        assert collision.tau == flow.units.relaxation_parameter_lu


@pytest.mark.parametrize("name,stencil,coll", [("obstacle2d_d2q9_bgk_f64", lt.D2Q9, "bgk"),
                                                ("obstacle3d_d3q27_kbc_f64", lt.D3Q27, "kbc")])
def test_obstacle_flow_matches_reference(name, stencil, coll):
    g = golden(name)
    res = [int(r) for r in g["resolution"]]
    flow = lt.Obstacle(ctx(), res, 100, 0.1, float(g["domain_length_x"]), stencil=stencil())
    flow.mask = g["obstacle_mask"]
    flow.initialize()
    np.testing.assert_allclose(flow.f.numpy(), g["f0"], rtol=0, atol=1e-15)
    collision = lt.BGKCollision(flow.units.relaxation_parameter_lu) if coll == "bgk" else lt.KBCCollision()
    sim = lt.Simulation(flow, collision, [])
    assert [type(b).__name__ for b in sim.boundaries[1:]] == list(g["boundary_order"])
    np.testing.assert_array_equal(sim.no_collision_mask.numpy(), g["no_collision_mask"])
    np.testing.assert_array_equal(sim.no_streaming_mask.numpy(), unpack_nsm(g))
    sim(2)
    np.testing.assert_allclose(flow.f.numpy(), g["f2"], rtol=0, atol=1e-13)


def test_shear_flows():
    flow2 = lt.DoublyPeriodicShear2D(ctx(), 16, 100, 0.05)
    assert float(flow2.initial_pu()[1][0].min()) == 1.0          # the reference's swapped branches
    g = golden("shear3d_d3q19_bgk_f64")
    flow = lt.DoublyPeriodicShear3D(ctx(), 16, 1000, 0.1)
    np.testing.assert_allclose(flow.f.numpy(), g["f0"], rtol=0, atol=1e-14)
    lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])(5)
    np.testing.assert_allclose(flow.f.numpy(), g["f5"], rtol=0, atol=1e-13)


# ------------------------------------------------------------------ invariants (tests/collision/*)
@pytest.mark.parametrize("cls", STENCILS, ids=[c.__name__ for c in STENCILS])
@pytest.mark.parametrize("collision", ["bgk", "kbc"])
def test_collision_conserves_mass_and_momentum(cls, collision):
    s = cls()
    if collision == "kbc" and cls not in (lt.D2Q9, lt.D3Q27):
        pytest.skip("KBC exists for D2Q9 / D3Q27")
    flow = UniformFlow(ctx(), [16] * s.d, 1, 0.01, s)
    torch.manual_seed(1)
    flow.f = flow.f * (1 + 0.05 * torch.rand_like(flow.f))
    coll = lt.BGKCollision(0.51) if collision == "bgk" else lt.KBCCollision(0.51)
    rho0, j0 = flow.rho(), flow.j()
    flow.f = coll(flow)
    assert torch.allclose(flow.rho(), rho0, atol=1e-12)
    assert torch.allclose(flow.j(), j0, atol=1e-12)


def test_bgk_fixpoint_two_applications_at_tau_half():
    flow = UniformFlow(ctx(), [16, 16], 1, 0.01, lt.D2Q9())
    torch.manual_seed(2)
    flow.f = torch.rand_like(flow.f) + 1
    f0 = copy(flow.f)
    coll = lt.BGKCollision(0.5)
    flow.f = coll(flow)
    flow.f = coll(flow)
    assert torch.allclose(flow.f, f0, atol=1e-12)


def test_equilibrium_moments():
    flow = UniformFlow(ctx(), [8, 8, 8], 1, 0.01, lt.D3Q27())
    feq = flow.equilibrium(flow)
    assert torch.allclose(flow.rho(feq), flow.rho(), atol=1e-13)
    assert torch.allclose(flow.j(feq), flow.j(), atol=1e-13)


def test_bounce_back_and_abb_masks():
    flow = UniformFlow(ctx(), [8, 6, 4], 1, 0.01, lt.D3Q19())
    f_old = copy(flow.f)
    out = lt.BounceBackBoundary(torch.ones(flow.resolution, dtype=torch.bool))(flow)
    assert torch.equal(out, f_old[flow.stencil.opposite])
    abb = lt.AntiBounceBackOutlet([0, -1, 0], flow)
    ncm = abb.make_no_collision_mask(flow.resolution, flow.context)
    nsm = abb.make_no_streaming_mask([19] + flow.resolution, flow.context)
    assert ncm[:, 0, :].all() and not ncm[:, 1:, :].any()
    moving_in = [q for q in range(19) if flow.stencil.e[q][1] == 1]
    assert nsm[moving_in][:, :, 0, :].all() and int(nsm.sum()) == len(moving_in) * 8 * 4
    with pytest.raises(AssertionError):
        lt.AntiBounceBackOutlet([1, 1, 0], flow)


# ------------------------------------------------------------------ API contract
def test_public_namespace_and_signatures():
    for name in ["Context", "Stencil", "TorchStencil", "UnitConversion", "Flow", "Equilibrium", "Boundary",
                 "Collision", "Reporter", "Simulation", "ExtFlow", "TaylorGreenVortex", "TaylorGreenVortex2D",
                 "TaylorGreenVortex3D", "Obstacle", "DoublyPeriodicShear2D", "QuadraticEquilibrium",
                 "BGKCollision", "KBCCollision", "KBCCollision2D", "KBCCollision3D", "NoCollision",
                 "BounceBackBoundary", "EquilibriumBoundaryPU", "AntiBounceBackOutlet", "Observable",
                 "ObservableReporter", "IncompressibleKineticEnergy", "MaximumVelocity", "Mass", "Enstrophy",
                 "LettuceException", "torch_gradient", "append_axes", "D1Q3", "D2Q9", "D3Q15", "D3Q19", "D3Q27"]:
        assert hasattr(lt, name), name
    c = lt.Context("cpu")
    assert c.dtype == torch.float32 and c.use_native is False
    with pytest.raises(AssertionError):
        lt.Context("cpu", use_native=True)
    assert lt.Context("cpu").convert_to_tensor(np.array([True])).dtype == torch.bool
    u = lt.UnitConversion(100, 0.05, characteristic_length_lu=128, characteristic_length_pu=2 * np.pi)
    assert u.relaxation_parameter_lu == pytest.approx(0.610851251684408, rel=1e-14)
    assert u.convert_time_to_pu(u.convert_time_to_lu(3.0)) == pytest.approx(3.0)


def test_reporter_cadence_and_deprecated_step():
    """sim(3); sim(2) -> reporters see i = 0..5."""
    flow = lt.TaylorGreenVortex(ctx(), [8, 8], 10, 0.05, lt.D2Q9())

    class Seen(lt.Reporter):
        def __init__(self):
            super().__init__(1)
            self.i = []

        def __call__(self, simulation):
            self.i.append(simulation.flow.i)
    rep = Seen()
    sim = lt.Simulation(flow, lt.BGKCollision(0.8), [rep])
    sim(3)
    with pytest.warns(DeprecationWarning):
        sim.step(2)
    assert rep.i == [0, 1, 2, 3, 4, 5]
    assert sim._steps_to_next_report(100) == 1
    rep.interval = 4
    assert sim._steps_to_next_report(100) == 1        # an unknown Reporter subclass is called after every step
    rep.batchable = True                              # ... unless it declares that it only acts on its interval
    assert sim._steps_to_next_report(100) == 3        # i = 5 -> next multiple of 4 is 8
    sim.reporter.clear()
    assert sim._steps_to_next_report(100) == 100


def test_checkpoint_roundtrip(tmp_path):
    flow = lt.TaylorGreenVortex(ctx(), [8, 8], 10, 0.05, lt.D2Q9())
    f0 = copy(flow.f)
    flow.dump(tmp_path / "f.pkl")
    lt.Simulation(flow, lt.BGKCollision(0.8), [])(3)
    assert not torch.equal(flow.f, f0)
    flow.load(tmp_path / "f.pkl")
    assert torch.equal(flow.f, f0)


def test_torch_gradient_orders():
    x = torch.linspace(0, 2 * np.pi * (1 - 1 / 64), 64, dtype=torch.float64)
    g = torch.meshgrid(x, x, indexing="ij")
    f = torch.sin(g[0]) * torch.cos(2 * g[1])
    dx = float(x[1] - x[0])
    for order, tol in ((2, 2e-2), (4, 2e-4), (6, 2e-6)):
        grad = lt.torch_gradient(f, dx=dx, order=order)
        assert torch.allclose(grad[0], torch.cos(g[0]) * torch.cos(2 * g[1]), atol=tol)
        assert torch.allclose(grad[1], -2 * torch.sin(g[0]) * torch.sin(2 * g[1]), atol=4 * tol)


# ------------------------------------------------------------------ C ABI (no compute calls)
def test_library_exports_every_declared_symbol(engine_library):
    from lettuce_amd import _native
    header = open(os.path.join(ROOT, "include", "lettuce_hip.h")).read()
    # entry points of the kernels that lost their A/B are declared under LT_EXPERIMENTS and exist only in a library
    # built with `make EXPERIMENTS=1` (lt_build_flags() & 1)
    experiments = "".join(re.findall(r"#ifdef LT_EXPERIMENTS\n(.*?)#endif", header, re.S))
    declared = set(re.findall(r"\b(lt_[a-z_]+)\s*\(", re.sub(r"#ifdef LT_EXPERIMENTS\n.*?#endif", "", header, flags=re.S)))
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    assert set(re.findall(r"\b(lt_[a-z_]+)\s*\(", experiments)) == set(_native.EXPERIMENT_SYMBOLS)
    lib = _native.load_library()
    for name in declared | (set(_native.EXPERIMENT_SYMBOLS) if lib.lt_build_flags() & 1 else set()):
        assert hasattr(lib, name), name
    assert not (lib.lt_build_flags() & 1) or _native.experiments_built()
    assert lib.lt_abi_version() == 2
    # argument checks that run before any HIP call
    bad = _native._PlanDesc()
    bad.abi_version = 99
    handle = _native.ctypes.c_void_p()
    assert lib.lt_plan_create(_native.ctypes.byref(bad), _native.ctypes.byref(handle)) == 1
    assert b"ABI version" in lib.lt_last_error()
    bad.abi_version, bad.stencil, bad.dims, bad.collision = 2, 1, 3, 2     # KBC on D3Q19
    bad.shape[0] = bad.shape[1] = bad.shape[2] = 8
    assert lib.lt_plan_create(_native.ctypes.byref(bad), _native.ctypes.byref(handle)) == 2
    assert b"KBC" in lib.lt_last_error()
    # empty and oversized grids are refused with a message (no HIP call has happened yet)
    bad.collision = 1
    bad.shape[2] = 0
    assert lib.lt_plan_create(_native.ctypes.byref(bad), _native.ctypes.byref(handle)) == 1
    assert b"shape[2]" in lib.lt_last_error()
    bad.shape[0], bad.shape[1], bad.shape[2] = 2048, 2048, 512            # 2^31 nodes
    assert lib.lt_plan_create(_native.ctypes.byref(bad), _native.ctypes.byref(handle)) == 2
    assert b"2^31" in lib.lt_last_error()
    assert lib.lt_plan_destroy(None) == 0


def test_two_step_admission_by_descriptor_knows_the_4_gib_limit(engine_library):
    """a plan with boundaries whose field reaches 4 GiB must not be put on the two-step path (its
    kernel addresses the field with 32-bit offsets); the rule lt_run / lt_plan_two_step_admitted apply is a
    descriptor-only query, checked here without a device: D3Q27 fp32 at 344^3 and D3Q27 fp64 at 272^3 are refused with
    masks, D3Q19 slabs of 4 GiB too; 256^3 is fine either way; Obstacle D3Q19 fp32 at 384^3 (4.0 GiB) is fine since
    round 4 (a descriptor per population)."""
    from lettuce_amd import _native
    lib = _native.load_library()
    c = _native.ctypes

    def limits(stencil, dtype, shape, masked, layout=0, ghosts=0):
        d = _native._PlanDesc()
        d.abi_version, d.stencil, d.dtype, d.collision = _native.LT_ABI_VERSION, _native.STENCIL_IDS[stencil], _native.DTYPE_IDS[dtype], 1
        d.layout, d.ghost_planes, d.dims = layout, ghosts, len(shape)
        for a, n in enumerate(shape):
            d.shape[a] = n
        w, r, ok = c.c_int32(), c.c_int32(), c.c_int32()
        assert lib.lt_two_step_limits(c.byref(d), int(masked), c.byref(w), c.byref(r), c.byref(ok)) == 0, lib.lt_last_error()
        return w.value, r.value, bool(ok.value)
    assert limits("D3Q19", torch.float32, [256] * 3, True) == (64, 8, True)
    # 19 * 384^3 * 4 B = 4.01 GiB: D3Q15 / D3Q19 BGK in the reference layout has the instantiation with a descriptor per
    # population (round 4), where one POPULATION must stay below 4 GiB; other collisions, D3Q27 and slabs keep the field rule
    assert limits("D3Q19", torch.float32, [384] * 3, True) == (64, 8, True)
    assert limits("D3Q19", torch.float32, [1024, 1024, 1020], True) == (64, 8, True)       # 3.98 GiB per population
    assert limits("D3Q19", torch.float32, [1024, 1024, 1024], True) == (64, 8, False)
    assert limits("D3Q27", torch.float32, [344, 344, 344], True)[2] is False               # 27 * 344^3 * 4 B = 4.09 GiB
    assert limits("D3Q19", torch.float32, [384] * 3, False) == (64, 8, True)
    assert limits("D3Q19", torch.float32, [376, 384, 384], True)[2]                # 3.93 GiB
    assert limits("D3Q27", torch.float32, [256] * 3, True) == (64, 4, True)
    assert limits("D3Q27", torch.float64, [272] * 3, True)[1:] == (0, False)       # no two-step kernel, and 4.05 GiB
    assert limits("D3Q19", torch.float64, [256] * 3, False) == (32, 8, True)
    # the slab layout counts its ghost planes
    assert limits("D3Q19", torch.float32, [512, 512, 212], True, layout=1, ghosts=2) == (64, 8, False)
    assert limits("D3Q19", torch.float32, [512, 512, 208], True, layout=1, ghosts=2) == (64, 8, True)
    assert limits("D2Q9", torch.float32, [4096, 4096], True) == (512, 1, True)
    bad = _native._PlanDesc()
    bad.abi_version, bad.stencil, bad.dims = 2, 1, 2
    assert lib.lt_two_step_limits(c.byref(bad), 0, None, None, None) == 1 and b"3-dimensional" in lib.lt_last_error()


def test_every_plan_method_that_launches_runs_on_the_plans_device():
    """a Plan method that hands torch's CURRENT stream to the engine must make the plan's GPU the
    current device first (``_on_device``), or a plan on cuda:1 launches on cuda:0's stream with cuda:1's pointers."""
    import inspect
    from lettuce_amd import _native
    missing = []
    for name, fn in vars(_native.Plan).items():
        if not callable(fn) or name.startswith("__"):
            continue
        src = inspect.getsource(inspect.unwrap(fn))
        if "_stream_handle()" in src or "torch.cuda.current_stream()" in src or ".record()" in src:
            if not hasattr(fn, "__wrapped__"):
                missing.append(name)
    assert missing == [], missing


def test_native_context_never_falls_back(monkeypatch, engine_library):
    """use_native with no engine library must raise, not run torch ops."""
    from lettuce_amd import _native
    flow = lt.TaylorGreenVortex(ctx(), [8, 8], 10, 0.05, lt.D2Q9())
    flow.context.use_native = True                    # as if a GPU context had been requested
    monkeypatch.setattr(_native, "_LIB", None)
    monkeypatch.setattr(_native, "library_path", lambda: "/nonexistent/liblettuce_hip.so")
    with pytest.raises(_native.NativeEngineError, match="not found"):
        lt.Simulation(flow, lt.BGKCollision(0.8), [])
    # and a component without kernels is refused loudly
    monkeypatch.undo()
    class D2Q5(lt.Stencil):                     # a lattice the engine has no kernels for
        def __init__(self):
            self.e = [[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]]
            self.w = [1 / 3] + [1 / 6] * 4
            self.opposite = [0, 3, 4, 1, 2]
    flow3 = UniformFlow(ctx(), [4, 4], 1, 0.01, D2Q5())
    flow3.context.use_native = True
    with pytest.raises(lt.LettuceException, match="D2Q5"):
        lt.Simulation(flow3, lt.BGKCollision(0.8), [])


def test_cli_benchmark_and_convergence_on_cpu(capsys):
    from lettuce_amd.cli import main
    assert main(["--no-cuda", "-p", "double", "benchmark", "-s", "3", "-r", "32", "-f", "taylor2D"]) == 0
    assert "MLUPS" in capsys.readouterr().out


def test_three_instruction_division_by_the_cs2_constants_is_the_ieee_quotient(tmp_path):
    """div_cs (kernels.hpp) replaces x / (2 cs^2) and x / cs^2 by a multiplication, an exact FMA
    remainder and an FMA correction; the C check compares that sequence with the IEEE division
    on every 61st fp32 bit pattern and on 4e7 random fp64 arguments (exhaustive fp32 run: stride 1)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = tmp_path / "exact_division_check"
    src = os.path.join(ROOT, "tests", "aux", "exact_division_check.c")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", src, "-o", str(exe), "-lm", "-lpthread"],
                   check=True)
    # f32two: the two-instruction fp32 form the kernels use since round 3 (x r_hi + RN(x r_lo))
    for args in (["f32two", "61"], ["f32", "61"], ["f64", "40000000"]):
        out = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.strip() == "mismatches 0", (args, out.stdout, out.stderr)


def test_masked_two_step_isa_has_no_dropped_register_copies(tmp_path):
    """Round 3's wrong result was hipcc dropping three register copies at the join of the equilibrium branch of
    lbm2m_kernel<float, D3Q27, slab, BGK, 64 x 4, AX = 0> (SIOptimizeVGPRLiveRange marks their source <undef>; the ISA
    then reads "; kill: def $vgprA killed $vgprB").  The product build switches that pass off (csrc/Makefile); this
    compiles that one kernel with the Makefile's flags -- device side only, ~10 s -- and looks for the pattern
    (`make -C lettuce_amd/csrc isa-check` does the same for every unit)."""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if hipcc is None:
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "lettuce_amd", "csrc")
    makefile = open(os.path.join(csrc, "Makefile")).read()
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*?)(?<!\\)$", makefile, re.M | re.S).group(1).replace("\\\n", " ").split()
    assert "-amdgpu-opt-vgpr-liverange=false" in flags, flags
    flags = [f.replace("$(ARCH)", "gfx950").replace("-I../../include", "-I" + os.path.join(ROOT, "include")) for f in flags]
    out = tmp_path / "masked_two_step.s"
    subprocess.run([hipcc] + flags + ["-I" + csrc, "-S", "--cuda-device-only", os.path.join(ROOT, "tests", "aux", "masked_two_step_isa.hip"),
                    "-o", str(out)], check=True, capture_output=True, timeout=600)
    isa = out.read_text()
    assert "lbm2m_kernel" in isa and "ds_bpermute_b32" in isa          # the kernel is there, outlet along the rows
    dropped = re.findall(r"kill: def \$[sv]gpr\d+ killed \$[sv]gpr\d+ ", isa)
    assert dropped == [], dropped


def test_a_plan_with_a_population_stride_refuses_dense_tensors():
    """with a population stride set, the engine addresses population q at q * stride; a dense
    clone() / empty_like() of a padded tensor handed to ANY entry point would be written past its end.  The checks
    of the binding run before the library is touched, so a CPU-side stand-in for the tensors shows them."""
    from lettuce_amd import _native

    class FakePlan(_native.Plan):
        def __init__(self):                                   # no library, no device
            self.dtype, self.q, self.d = torch.float32, 19, 3
            self.resolution, self.layout, self.ghost_planes = [8, 4, 6], _native.LAYOUT_SLAB, 2
            self.pop_stride, self._handle, self._const = 8 * 4 * 10 + 64, None, {"blocks": 19, ("crossing", 1): [5, 7, 10, 11, 14]}

    class OnDevice:                                           # a meta tensor that claims to live on the GPU
        def __init__(self, t):
            self.t = t
        device = torch.device("cuda", 0)
        def __getattr__(self, name):
            return getattr(self.t, name)
        def __getitem__(self, i):
            return OnDevice(self.t[i])

    plan = FakePlan()
    shape = plan.f_shape
    dense = OnDevice(torch.empty(shape, device="meta"))
    inner = torch.empty(shape[1:], device="meta").stride()
    padded = OnDevice(torch.empty(plan.q * plan.pop_stride, device="meta").as_strided(shape, (plan.pop_stride,) + tuple(inner)))
    plan._tensor_ok(padded, shape)
    plan._populations_ok(padded, padded)
    with pytest.raises(_native.NativeEngineError, match="elements between populations"):
        plan._tensor_ok(dense, shape)
    with pytest.raises(_native.NativeEngineError, match="elements between populations"):
        plan._populations_ok(padded, dense)
    plan._tensor_ok(OnDevice(torch.empty(shape[1:], device="meta")), shape[1:])      # per-node fields stay dense
    small = OnDevice(torch.empty([18, 4, 8], device="meta"))
    with pytest.raises(_native.NativeEngineError, match="halo message buffer"):
        plan._message_ok(small, plan.two_step_message_blocks())
    plan._message_ok(OnDevice(torch.empty([19, 4, 8], device="meta")), plan.two_step_message_blocks())
    # every slab entry point of the binding checks its population arguments
    import inspect
    unchecked = [name for name in ("collide_planes", "stream_planes", "stream_collide_planes", "stream_collide_plane_pair",
                                   "stream_collide_plane_pair_packed", "stream_collide_twice_planes",
                                   "stream_collide_twice_planes_packed", "stream_collide_twice_edges",
                                   "stream_collide_twice_edges_direct", "stream_collide_twice_slab", "pack", "unpack",
                                   "pack_two_step", "unpack_two_step")
                 if "_populations_ok(" not in inspect.getsource(inspect.unwrap(getattr(_native.Plan, name)))]
    assert unchecked == [], unchecked


def test_bench_finds_the_committed_pmc_traffic_for_the_fused_kernel(monkeypatch):
    """bench.py fills roofline.traffic from profiles/traffic.json by kernel name -- the name the engine
    reports (lt_plan_kernel_name) and the one rocprofv3 prints differ in case and suffix -- and only when
    the table was measured on a build of the present kernel sources (its source_hash)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
        table = json.load(fh)
    name = "lbm2_kernel<float, lt::d3q19, 0, 1, 64, 8, 1, 0, 1>"
    # a table from other kernel sources is not used
    monkeypatch.setattr(bench, "source_hash", lambda: "not-the-hash-of-the-table")
    assert bench.traffic_from_profile(name) is None
    # the dominant kernel of the default bench run: two lattice updates per launch, so the HBM bytes
    # of a launch are about half the algorithmic bytes of the two updates
    monkeypatch.setattr(bench, "source_hash", lambda: table.get("source_hash"))
    traffic = bench.traffic_from_profile(name)
    assert traffic is not None
    assert 0.45 < traffic / (2 * 152 * 256 ** 3) < 0.6
    # names that gained trailing template parameters since the profile was taken still match
    assert bench.traffic_from_profile(name.replace(", 1>", ", 1, 0>")) == traffic
    assert bench.traffic_from_profile("lbm_kernel<float, lt::d3q19, 0, 1, true, true, false, 1, 0, 0, false>") is None
    # round 3: the other BASELINE workloads have rows of their own (bench.py's other_configs)
    kbc = bench.traffic_from_profile("lbm_kernel_occ4<float, lt::d3q27, 0, 2, true, true, true, 1, 0, 3, false>",
                                     "obstacle3d_d3q27_kbc_f32_256")
    assert kbc is not None and 0.95 < kbc / (217 * 256 ** 3) < 1.1              # one pass: 217 B per node
    fp64 = bench.traffic_from_profile("lbm2_kernel<double, lt::d3q19, 0, 1, 32, 8, 1, 0, 1>", "shear3d_d3q19_bgk_f64_384x384x96")
    assert fp64 is not None and 0.5 < fp64 / (2 * 304 * 384 * 384 * 96) < 0.6
    assert bench.traffic_from_profile(name, "obstacle3d_d3q27_kbc_f32_256") is None


def test_only_reporters_that_opt_in_are_batched():
    """The reference calls every reporter after every step (lettuce/_simulation.py:203-205); the engine fuses
    the steps between two calls only for reporters that declare batchable = True (the library's do)."""
    flow = lt.TaylorGreenVortex(ctx(), [8, 8], 100, 0.05, lt.D2Q9())
    calls = []

    class EveryStep(lt.Reporter):
        def __call__(self, simulation):
            calls.append(simulation.flow.i)

    with pytest.MonkeyPatch.context() as mp:
        mp.setattr("sys.stdout", io.StringIO())
        lib = lt.ObservableReporter(lt.IncompressibleKineticEnergy(flow), interval=5, out=[])
    sim = lt.Simulation(flow, lt.BGKCollision(0.6), [lib])
    assert sim._steps_to_next_report(100) == 5
    sim.reporter = [lib, EveryStep(interval=5)]          # an unknown subclass: no batching, whatever its interval
    assert sim._steps_to_next_report(100) == 1
    EveryStep.batchable = True
    assert sim._steps_to_next_report(100) == 5
