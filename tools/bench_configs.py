"""MLUPS of the other BASELINE.json configurations through the lettuce-style API (single GPU):
cfg1 TGV2D D2Q9 128^2 fp64, cfg4 Obstacle3D D3Q27 256^3 KBC fp32 (inlet + ABB outlet + sphere
bounce-back), cfg5's per-GPU slab (periodic shear D3Q19 384x384x96 fp64).  One JSON line each.
usage: bench_configs.py [res=N] [cfg1 cfg4 cfg4bgk cfg4bgk1 obst19 obst19_1 cfg5]   (res: edge of the Obstacle cube, 256)"""
import sys, os, json, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt

def timed(sim, warm, steps):
    sim(warm); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sim._native.fused_events = (e0, e1)
    t0 = time.perf_counter(); sim(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # the events bracket the dominant kind of fused launch of the batch: ms per lattice update
    info = sim._native.plan.last_run_info()
    updates = 2 * info["two_step_launches"] if info["two_step_launches"] else (
        info["single_step_launches"] or 8 * info["many_step_launches"])
    fused_ms = e0.elapsed_time(e1) / max(1, updates)
    sim._native.fused_events = None
    timed.updates_per_launch = 2 if info["two_step_launches"] else (8 if info["many_step_launches"] else 1)
    return dt, fused_ms

def residency_ab(sim, steps=40):
    """fused-kernel ms for a few residency caps, interleaved (RESIDENCY_AB=1)"""
    if not os.environ.get("RESIDENCY_AB"):
        return
    out = {}
    for r in range(3):
        for cap in (0, 5, 4, 3, -1):
            sim._native.plan.set_residency(cap)
            out.setdefault(cap, []).append(timed(sim, 2, steps)[1])
    sim._native.plan.set_residency(-1)
    print(json.dumps({"residency_ab_ms": {k: round(sorted(v)[1], 4) for k, v in out.items()}}), flush=True)

def report(name, flow, sim, dt, fused_ms, steps, bytes_per_node):
    n = 1
    for r in flow.resolution: n *= r
    gbs = bytes_per_node * n / (fused_ms * 1e-3) / 1e9           # algorithmic bytes of lattice updates per second
    upl = getattr(timed, "updates_per_launch", 1)
    hbm = gbs / upl                                               # populations read once + written once per launch
    print(json.dumps({"config": name, "resolution": flow.resolution, "steps": steps,
                      "mlups_wall": round(steps * n / dt / 1e6, 1), "ms_per_update": round(fused_ms, 5),
                      "updates_per_launch": upl, "hbm_GBps_required": round(hbm, 1),
                      "frac_of_8TBs": round(hbm / 8000, 4), "algorithmic_update_GBps": round(gbs, 1),
                      "bytes_per_node_and_update": bytes_per_node, "kernel": sim._native.plan.kernel_name()}), flush=True)

def main():
    edge = ([int(a[4:]) for a in sys.argv[1:] if a.startswith("res=")] or [256])[0]
    which = [a for a in sys.argv[1:] if not a.startswith("res=")] or ["cfg1", "cfg4", "cfg5", "cfg4bgk"]
    if "cfg1" in which:
        ctx = lt.Context("cuda:0", torch.float64, True)
        flow = lt.TaylorGreenVortex(ctx, [128, 128], 100, 0.05, lt.D2Q9())
        sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
        dt, ms = timed(sim, 100, 1000)
        report("cfg1 TGV2D D2Q9 128^2 BGK fp64 (launch-bound)", flow, sim, dt, ms, 1000, 144)
        e = float(lt.IncompressibleKineticEnergy(flow)())
        print(json.dumps({"cfg1_energy_after_1100_steps": e}), flush=True)
    for tag, coll, stencil, two_step in (("cfg4", "kbc", lt.D3Q27, -1), ("cfg4bgk", "bgk", lt.D3Q27, -1),
                                         ("cfg4bgk1", "bgk", lt.D3Q27, 0), ("obst19", "bgk", lt.D3Q19, -1),
                                         ("obst19_1", "bgk", lt.D3Q19, 0)):
        if tag not in which: continue
        ctx = lt.Context("cuda:0", torch.float32, True)
        flow = lt.Obstacle(ctx, [edge] * 3, 100, 0.1, domain_length_x=4, stencil=stencil())
        x, y, z = flow.grid
        flow.mask = ((x - 1) ** 2 + (y - 2) ** 2 + (z - 2) ** 2) < 0.5 ** 2
        flow.initialize()
        collision = lt.KBCCollision() if coll == "kbc" else lt.BGKCollision(flow.units.relaxation_parameter_lu)
        sim = lt.Simulation(flow, collision, [])
        sim._native.batch(1)                       # compiles the masks; then the pairing can be chosen
        sim._native.plan.set_two_step(two_step)
        dt, ms = timed(sim, 10, 100)
        q = stencil().q
        report(f"{tag}: Obstacle3D D3Q{q} {edge}^3 {coll.upper()} fp32, inlet+ABB outlet+sphere BB"
               + (" (one update per launch forced)" if two_step == 0 else ""), flow, sim, dt, ms, 100, 8 * q + 1)
        residency_ab(sim)
        u = flow.u()
        print(json.dumps({"finite": bool(torch.isfinite(flow.f).all()), "umax_lu": float(u.abs().max())}), flush=True)
        del sim, flow
        torch.cuda.empty_cache()
    if "cfg5" in which:
        ctx = lt.Context("cuda:0", torch.float64, True)
        flow = lt.DoublyPeriodicShear3D(ctx, [384, 384, 96], 10000, 0.1)
        sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
        dt, ms = timed(sim, 10, 100)
        report("cfg5 per-GPU slab: periodic shear D3Q19 384x384x96 BGK fp64", flow, sim, dt, ms, 100, 304)
        residency_ab(sim)

main()
