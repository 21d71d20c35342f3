"""Does capping the workgroups resident per CU (lt_plan_set_residency) change the fused kernel's
HBM rate?  Interleaved A/B in one process, HIP events.  Development probe."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan


def ev():
    return torch.cuda.Event(enable_timing=True)


def time_fused(plan, a, b, tau, iters=20):
    e0, e1 = ev(), ev()
    for _ in range(2):
        plan.stream_collide(a, b, tau); a, b = b, a
    e0.record()
    for _ in range(iters):
        plan.stream_collide(a, b, tau); a, b = b, a
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


only = os.environ.get("CASES")
cases = [("D3Q19", torch.float32, "bgk", [256] * 3), ("D3Q27", torch.float32, "bgk", [256] * 3),
         ("D3Q19", torch.float64, "bgk", [256] * 3), ("D3Q27", torch.float32, "kbc", [256] * 3)]
lds_list = [int(x) for x in os.environ.get("RESIDENCY", "0,8,6,5,4,3,2,-1").split(",")]
rounds = int(os.environ.get("ROUNDS", 3))
if only:
    cases = [cases[int(i)] for i in only.split(",")]
for name, dt, coll, res in cases:
    plan = Plan(name, dt, coll, res, [], device=torch.device("cuda:0"))
    a = torch.rand(plan.f_shape, device="cuda", dtype=dt) * 0.01 + 0.05
    b = torch.empty_like(a)
    out = {}
    for r in range(rounds):
        for lds in lds_list:
            plan.set_residency(lds)
            out.setdefault(lds, []).append(time_fused(plan, a, b, 0.6))
    print(json.dumps({"case": f"{name} {coll} {str(dt)[6:]}", "ms_by_workgroups_per_cu": {k: round(sorted(v)[len(v) // 2], 4) for k, v in out.items()}}), flush=True)
    del a, b, plan
