"""cfg4 (Obstacle3D D3Q27 256^3 KBC fp32) with two lattice updates per launch (lbm2m_kernel with
the KBC collision, opt-in through lt_plan_set_two_step(plan, 1, 0)) against the one-step masked kernel: time per
update and agreement of the populations after the same steps."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt


def build(n):
    ctx = lt.Context("cuda:0", torch.float32, True)
    flow = lt.Obstacle(ctx, [n, n, n], 100, 0.1, domain_length_x=4, stencil=lt.D3Q27())
    x, y, z = flow.grid
    flow.mask = ((x - 1) ** 2 + (y - 2) ** 2 + (z - 2) ** 2) < 0.5 ** 2
    flow.initialize()
    sim = lt.Simulation(flow, lt.KBCCollision(), [])
    sim._native.batch(1)
    return flow, sim


def timed(sim, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sim._native.fused_events = (e0, e1)
    sim(steps); torch.cuda.synchronize()
    info = sim._native.plan.last_run_info()
    sim._native.fused_events = None
    launches = info["two_step_launches"] or info["single_step_launches"]
    per = 2 if info["two_step_launches"] else 1
    return e0.elapsed_time(e1) / launches / per, info


for n in (64, 256):
    fa, sa = build(n)
    fb, sb = build(n)
    sb._native.plan.set_two_step(1, 0)
    why = sb._native.plan.two_step_admitted()
    sa(41); sb(41)
    d = float((fa.f - fb.f).abs().max())
    print(json.dumps({"n": n, "two_step_admitted": why is None, "why": why, "kernel_one": sa._native.plan.kernel_name(),
                      "kernel_two": sb._native.plan.kernel_name(), "info_two": sb._native.plan.last_run_info(),
                      "max_abs_diff_after_42_steps": d, "max_f": float(fa.f.abs().max()),
                      "finite": bool(torch.isfinite(fb.f).all())}), flush=True)
    if n == 256:
        for rep in range(3):
            ma, _ = timed(sa, 60)
            mb, info = timed(sb, 60)
            print(json.dumps({"n": n, "ms_per_update_one_step": round(ma, 4), "ms_per_update_two_step": round(mb, 4),
                              "glups_one": round(n ** 3 / ma / 1e6, 2), "glups_two": round(n ** 3 / mb / 1e6, 2), "info": info}), flush=True)
    del fa, sa, fb, sb
    torch.cuda.empty_cache()
