"""One BASELINE workload through lt.Simulation for the rocprofv3 passes of tools/gpu_round3.sh (kernel trace and
PMC counters need a process that runs ONE workload: the counters are summed per kernel name).
usage: profile_workload.py cfg2|cfg4|cfg4bgk|obst19|cfg5 [steps]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt

which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda:0")
if which == "cfg2":
    ctx = lt.Context(dev, torch.float32, True)
    edge = int(os.environ.get("LT_PROFILE_EDGE", "256"))          # other sizes: tools/pmc_tlb.sh
    shape = [int(v) for v in os.environ["LT_PROFILE_RES"].split(",")] if os.environ.get("LT_PROFILE_RES") else [edge] * 3
    flow = lt.TaylorGreenVortex(ctx, shape, 1600, 0.1, lt.D3Q19())
    sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
    q, esize, key = 19, 4, "tgv3d_d3q19_bgk_f32_" + "x".join(str(v) for v in shape)
elif which in ("cfg4", "cfg4bgk", "obst19"):
    ctx = lt.Context(dev, torch.float32, True)
    stencil = lt.D3Q19() if which == "obst19" else lt.D3Q27()
    flow = lt.Obstacle(ctx, [256] * 3, 100, 0.1, domain_length_x=4, stencil=stencil)
    x, y, z = flow.grid
    flow.mask = ((x - 1) ** 2 + (y - 2) ** 2 + (z - 2) ** 2) < 0.5 ** 2
    flow.initialize()
    coll = lt.KBCCollision() if which == "cfg4" else lt.BGKCollision(flow.units.relaxation_parameter_lu)
    sim = lt.Simulation(flow, coll, [])
    q, esize = stencil.q, 4
    key = {"cfg4": "obstacle3d_d3q27_kbc_f32_256", "cfg4bgk": "obstacle3d_d3q27_bgk_f32_256",
           "obst19": "obstacle3d_d3q19_bgk_f32_256"}[which]
elif which == "cfg5":
    ctx = lt.Context(dev, torch.float64, True)
    flow = lt.DoublyPeriodicShear3D(ctx, [384, 384, 96], 10000, 0.1)
    sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
    q, esize, key = 19, 8, "shear3d_d3q19_bgk_f64_384x384x96"
elif which in ("slab", "slab5"):
    # one rank of the multi-GPU path by itself (no transport: the halo messages stay where the edge launch wrote them):
    # the per-GPU slab of cfg3 / cfg5 through the two-step slab driver -- edge launch + sweep per double step
    res, dt, q, esize = (([512, 512, 64], torch.float32, 19, 4) if which == "slab" else ([384, 384, 96], torch.float64, 19, 8))
    ctx = lt.Context(dev, dt, True)
    slab = lt.ZSlab(res, 0, 1)
    if which == "slab":
        flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab)
        key = "slab_tgv3d_d3q19_bgk_f32_512x512x64"
    else:
        flow = lt.DoublyPeriodicShear3D(ctx, slab.extended_resolution, 10000, 0.1, slab=slab)
        key = "slab_shear3d_d3q19_bgk_f64_384x384x96"
    sim = lt.TwoStepSlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
    flow.resolution = res                       # nodes of the slab (the extended slab was only needed for the set-up)
else:
    raise SystemExit(f"unknown workload {which}")
sim(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
sim(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_ = sim.f if which.startswith("slab") else flow.f
torch.cuda.synchronize()
n = 1
for r in flow.resolution:
    n *= r
print(json.dumps({"workload": key, "which": which, "nodes": n, "q": q, "esize": esize, "steps": steps,
                  "kernel": (sim.engine if which.startswith("slab") else sim._native.plan).kernel_name(),
                  "mlups_wall": round(steps * n / dt / 1e6, 1),
                  "last_run": None if which.startswith("slab") else sim._native.plan.last_run_info()}), flush=True)
