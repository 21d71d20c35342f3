import sys, os, json
sys.path.insert(0, "/root/repo")
import torch
from lettuce_amd._native import Plan
def ev(): return torch.cuda.Event(enable_timing=True)
for res in ([256,256,256], [384,384,96]):
    plan = Plan("D3Q19", torch.float64, "bgk", res, [], device=torch.device("cuda:0"))
    f = torch.rand(plan.f_shape, device="cuda", dtype=torch.float64) * 0.01 + 0.05
    a, b, c = torch.empty_like(f), torch.empty_like(f), torch.empty_like(f)
    plan.stream_collide(f, a, 0.6); plan.stream_collide(a, b, 0.6)
    plan.stream_collide_twice(f, c, 0.6); torch.cuda.synchronize()
    same = bool(torch.equal(b, c))
    out = {}
    for r in range(5):
        for label in ("single", "twice"):
            e0, e1 = ev(), ev()
            x, y = f, a
            for it in range(8):
                if it == 2: e0.record()
                if label == "single":
                    plan.stream_collide(x, y, 0.6); plan.stream_collide(y, x, 0.6)
                else:
                    plan.stream_collide_twice(x, y, 0.6); x, y = y, x
            e1.record(); torch.cuda.synchronize()
            out.setdefault(label, []).append(e0.elapsed_time(e1) / 12)
    print(json.dumps({"res": res, "bit_identical": same, "ms_per_step": {k: round(sorted(v)[2], 4) for k, v in out.items()}}), flush=True)
    del f, a, b, c, plan
