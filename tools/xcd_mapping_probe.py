"""lbm2_kernel at 256^3: block -> tile numbering (0 = an eighth of the grid per XCD, 3 = none, 4 = an eighth of every
segment layer per XCD) by segment length, interleaved repetitions.  Dev tool."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

res = [256] * 3
plan = Plan("D3Q19", torch.float32, "bgk", res, [], device=torch.device("cuda:0"))
f = torch.rand(plan.f_shape, device="cuda") * 0.01 + 0.05
g = torch.empty_like(f)
out = {}
for r in range(5):
    for seg in (128, 64, 32):
        for pol in (0, 3, 4):
            plan.set_two_step(1, seg); plan.set_shift_policy(pol)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a, b = f, g
            for it in range(12):
                if it == 2:
                    e0.record()
                plan.stream_collide_twice(a, b, 0.6); a, b = b, a
            e1.record(); torch.cuda.synchronize()
            out.setdefault(f"seg{seg} policy{pol}", []).append(e0.elapsed_time(e1) / 10)
print(json.dumps({k: round(sorted(v)[len(v) // 2], 4) for k, v in out.items()}))
