"""Two-step kernel on large single-GPU grids: segment length (planes per workgroup) against the automatic choice."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

for res in ([512, 512, 512], [256, 512, 512], [384, 384, 384]):
    plan = Plan("D3Q19", torch.float32, "bgk", res, [], device=torch.device("cuda:0"))
    f = torch.rand(plan.f_shape, device="cuda") * 0.01 + 0.05
    g = torch.empty_like(f)
    out = {"res": res}
    n2 = res[0]
    for seg in (0, n2, n2 // 2, n2 // 4, n2 // 8):
        plan.set_two_step(1, seg)
        vals = []
        for r in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a, b = f, g
            for it in range(6):
                if it == 2:
                    e0.record()
                plan.stream_collide_twice(a, b, 0.6); a, b = b, a
            e1.record(); torch.cuda.synchronize()
            vals.append(e0.elapsed_time(e1) / 4)
        n = res[0] * res[1] * res[2]
        out[f"seg{seg}"] = {"ms_per_launch": round(sorted(vals)[1], 4), "glups": round(2 * n / sorted(vals)[1] / 1e6, 1)}
    print(json.dumps(out), flush=True)
    del plan, f, g
    torch.cuda.empty_cache()
