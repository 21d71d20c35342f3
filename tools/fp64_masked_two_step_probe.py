"""Obstacle3D D3Q19 256^3 BGK fp64 (inlet + anti-bounce-back outlet + sphere): the masked one-step kernel against the
masked two-step kernel on 32 x 4 tiles (round 3: 8 rows do not fit the LDS with the outlet's third downward slot)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt


def build(two_step):
    ctx = lt.Context("cuda:0", torch.float64, True)
    flow = lt.Obstacle(ctx, [256, 256, 256], 100, 0.1, domain_length_x=4, stencil=lt.D3Q19())
    x, y, z = flow.grid
    flow.mask = ((x - 1) ** 2 + (y - 2) ** 2 + (z - 2) ** 2) < 0.5 ** 2
    flow.initialize()
    sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
    sim._native.batch(1)
    sim._native.plan.set_two_step(two_step)
    return flow, sim


def timed(sim, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sim._native.fused_events = (e0, e1)
    sim(steps); torch.cuda.synchronize()
    info = sim._native.plan.last_run_info()
    sim._native.fused_events = None
    launches = info["two_step_launches"] or info["single_step_launches"]
    return e0.elapsed_time(e1) / launches / (2 if info["two_step_launches"] else 1), info


fa, sa = build(0)
fb, sb = build(1)
print(json.dumps({"admitted": sb._native.plan.two_step_admitted(), "kernel_two": sb._native.plan.kernel_name(),
                  "kernel_one": sa._native.plan.kernel_name(), "resident": sb._native.plan.resident_enabled()}), flush=True)
sa(21); sb(21)
print(json.dumps({"bit_identical_after_22_steps": bool(torch.equal(fa.f, fb.f)), "info": sb._native.plan.last_run_info()}), flush=True)
for rep in range(3):
    ma, _ = timed(sa, 40)
    mb, info = timed(sb, 40)
    print(json.dumps({"ms_per_update_one_step": round(ma, 4), "ms_per_update_two_step": round(mb, 4),
                      "glups_one": round(256 ** 3 / ma / 1e6, 2), "glups_two": round(256 ** 3 / mb / 1e6, 2), "info": info}), flush=True)
