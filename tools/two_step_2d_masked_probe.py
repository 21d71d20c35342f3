"""Large 2-D Obstacle (inlet, outlet, cylinder): masked one-step kernel against lbm2d2m_kernel.  One JSON line per grid."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt

for dtype in (torch.float32, torch.float64):
    for res in ([4096, 2048], [8192, 4096], [2048, 1024]):
        out = {"flow": "Obstacle2D D2Q9 BGK (inlet, outlet, cylinder)", "dtype": str(dtype), "res": res}
        finals = []
        for mode in (0, 1):
            ctx = lt.Context("cuda:0", dtype, True)
            flow = lt.Obstacle(ctx, res, 100, 0.05, domain_length_x=4, stencil=lt.D2Q9())
            x, y = flow.grid
            flow.mask = ((x - 1) ** 2 + (y - 1) ** 2) < 0.3 ** 2
            flow.initialize()
            sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
            sim._native.batch(1)
            sim._native.plan.set_two_step(mode)
            sim(21); torch.cuda.synchronize()
            t0 = time.perf_counter(); sim(200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out["one-step" if mode == 0 else "two-step"] = {
                "ms_per_step": round(dt / 200 * 1e3, 4), "glups": round(200 * res[0] * res[1] / dt / 1e9, 2),
                "kernel": sim._native.plan.kernel_name()}
            finals.append(flow.f.clone())
            del sim, flow
        out["bit_identical"] = bool(torch.equal(finals[0], finals[1]))
        out["finite"] = bool(torch.isfinite(finals[0]).all())
        print(json.dumps(out), flush=True)
        del finals
        torch.cuda.empty_cache()
