"""Launch-bound grids through lt.Simulation: microseconds per step with one step per launch and with the
several-steps-per-launch kernels (2-D: up to 8, lbm_many_kernel; 3-D: 2, lbm_many3d_kernel; round 3)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt
for res, st, dt in (([128, 128], lt.D2Q9, torch.float64), ([16, 16, 16], lt.D3Q19, torch.float32), ([32, 32, 32], lt.D3Q19, torch.float32),
                    ([32, 32, 32], lt.D3Q19, torch.float64), ([32, 32, 32], lt.D3Q27, torch.float32),
                    ([48, 48, 48], lt.D3Q19, torch.float32), ([64, 64, 64], lt.D3Q19, torch.float32),
                    ([96, 96, 96], lt.D3Q19, torch.float32), ([128, 128, 128], lt.D3Q19, torch.float32)):
    ctx = lt.Context("cuda:0", dt, True)
    for mode in (0, 1):
        flow = lt.TaylorGreenVortex(ctx, res, 100, 0.05, st())
        sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
        sim._native.plan.set_many_step(mode)
        sim._native.plan.set_two_step(0)
        sim(200)
        t = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); sim(2000); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
        n = 1
        for r in res: n *= r
        print(json.dumps({"res": res, "stencil": st.__name__, "dtype": str(dt).split(".")[1], "several_steps_per_launch": bool(mode),
                          "kernel": sim._native.plan.kernel_name(), "us_per_step": round(min(t) / 2000 * 1e6, 2),
                          "mlups": round(2000 * n / min(t) / 1e6, 1), "last_run": sim._native.plan.last_run_info()}), flush=True)
