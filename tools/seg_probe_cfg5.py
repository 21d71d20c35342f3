"""cfg5's per-GPU slab shape (384 x 384 x 96, D3Q19 fp64): segment length of the two-step kernel against the automatic choice."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

res = [384, 384, 96]          # reference layout: a2 = x = 384 planes of 384 x 96
plan = Plan("D3Q19", torch.float64, "bgk", res, [], device=torch.device("cuda:0"))
f = torch.rand(plan.f_shape, device="cuda", dtype=torch.float64) * 0.01 + 0.05
g = torch.empty_like(f)
out = {"res": res}
for rep in range(3):
    for seg in (0, 384, 192, 128, 96, 64, 48, 32, 24, 16):
        plan.set_two_step(1, seg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a, b = f, g
        for it in range(8):
            if it == 2:
                e0.record()
            plan.stream_collide_twice(a, b, 0.6); a, b = b, a
        e1.record(); torch.cuda.synchronize()
        out.setdefault(f"seg{seg}", []).append(e0.elapsed_time(e1) / 6 / 2)
print(json.dumps({k: (round(sorted(v)[1], 4) if isinstance(v, list) and len(v) == 3 else v) for k, v in out.items()}))
