import sys, json
sys.path.insert(0, "/root/repo")
import torch
from lettuce_amd._native import Plan
torch.manual_seed(0)
for lat, dt, res in (("D3Q19", torch.float32, [256]*3), ("D3Q19", torch.float64, [384, 384, 96]), ("D3Q27", torch.float32, [256, 256, 128])):
    q = int(lat[3:])
    w = torch.rand(q, 1, 1, 1, device="cuda", dtype=dt) * 0.03 + 0.02
    f0 = (w * (1 + 0.05 * torch.rand([q] + res, device="cuda", dtype=dt))).contiguous()
    outs = []
    for mode in (0, 1):
        plan = Plan(lat, dt, "bgk", res, [], device=torch.device("cuda:0"))
        plan.set_two_step(mode)
        r, o = plan.run(f0.clone(), torch.empty_like(f0), 0.55, 301)
        outs.append(r.clone()); info = plan.last_run_info()
        del plan, r, o
    print(json.dumps({"case": f"{lat} {str(dt)[6:]} {res}", "bit_identical_after_301_steps": bool(torch.equal(outs[0], outs[1])),
                      "finite": bool(torch.isfinite(outs[1]).all()), "last": info}), flush=True)
    del outs, f0
