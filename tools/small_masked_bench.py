"""Small 2-D boundary flows (launch-bound): lt_run with up to 7 steps per launch (lbm_many_kernel<..., MASKED>)
against one launch per step.  One JSON line per grid."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt

for dtype in (torch.float64, torch.float32):
    for res in ([64, 32], [128, 64], [128, 128], [256, 128], [256, 256]):
        out = {"flow": "Obstacle2D D2Q9 BGK (inlet, outlet, cylinder)", "dtype": str(dtype), "res": res}
        finals = []
        for mode in (0, -1, 1):
            ctx = lt.Context("cuda:0", dtype, True)
            flow = lt.Obstacle(ctx, res, 20, 0.05, domain_length_x=4, stencil=lt.D2Q9())
            x, y = flow.grid
            flow.mask = ((x - 1) ** 2 + (y - 1) ** 2) < 0.3 ** 2
            flow.initialize()
            sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
            sim._native.batch(1)
            sim._native.plan.set_many_step(mode)
            sim(200); torch.cuda.synchronize()
            t0 = time.perf_counter(); sim(5000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out[{0: "one launch per step", -1: "automatic", 1: "forced many-step"}[mode]] = {
                "us_per_step": round(dt / 5000 * 1e6, 3), "mlups": round(5000 * res[0] * res[1] / dt / 1e6, 1),
                "kernel": sim._native.plan.kernel_name(), "launches": sim._native.plan.last_run_info()}
            finals.append(flow.f.clone())
        out["bit_identical"] = bool(torch.equal(finals[0], finals[1]))
        out["finite"] = bool(torch.isfinite(finals[0]).all())
        print(json.dumps(out), flush=True)
