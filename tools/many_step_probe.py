"""Small 2-D grids: ms per step of lt_run with and without the many-steps-per-launch kernel."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

for dt in (torch.float64, torch.float32):
    for n in (64, 128, 256, 512, 1024):
        out = {}
        for mode in (0, 1):
            plan = Plan("D2Q9", dt, "bgk", [n, n], [], device=torch.device("cuda:0"))
            plan.set_many_step(mode)
            a = torch.rand(plan.f_shape, device="cuda", dtype=dt) * 0.01 + 0.1
            b = torch.empty_like(a)
            plan.run(a, b, 0.6, 50); torch.cuda.synchronize()
            ts = []
            for r in range(5):
                t0 = time.perf_counter(); plan.run(a, b, 0.6, 1001); torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 1001 * 1e6)
            out["many" if mode else "single"] = round(sorted(ts)[2], 3)
        print(json.dumps({"grid": n, "dtype": str(dt)[6:], "us_per_step": out,
                          "glups_many": round(n * n / out["many"] / 1e3, 2), "glups_single": round(n * n / out["single"] / 1e3, 2)}), flush=True)
