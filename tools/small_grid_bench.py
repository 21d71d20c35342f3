"""Launch-bound grids: eager launches vs hipGraph replay inside lt_run (cfg1 shape and others)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt
for res, st, dt in (([128, 128], lt.D2Q9, torch.float64), ([256, 256], lt.D2Q9, torch.float64),
                    ([32, 32, 32], lt.D3Q19, torch.float32), ([64, 64, 64], lt.D3Q19, torch.float32)):
    ctx = lt.Context("cuda:0", dt, True)
    for mode in (0, 1):
        flow = lt.TaylorGreenVortex(ctx, res, 100, 0.05, st())
        sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
        sim._native.plan.set_graph_mode(mode)
        sim(200)
        t = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); sim(2000); t.append(time.perf_counter() - t0)
        n = 1
        for r in res: n *= r
        print(json.dumps({"res": res, "graph": bool(mode), "us_per_step": round(min(t) / 2000 * 1e6, 2),
                          "mlups": round(2000 * n / min(t) / 1e6, 1)}), flush=True)
