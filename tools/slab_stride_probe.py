"""Two-step kernel on a 512 x 512 slab: how does the population stride (planes incl. ghosts) matter?"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan, LAYOUT_SLAB

out = {}
for rnd in range(3):
    for nz in (56, 60, 61, 62, 63, 64, 65, 66, 68, 72):
        plan = Plan("D3Q19", torch.float32, "bgk", [512, 512, nz], [], layout=LAYOUT_SLAB, ghost_planes=2)
        a = torch.rand(plan.f_shape, device="cuda") * 0.01 + 0.05
        b = torch.empty_like(a)
        n2 = a.shape[1]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(12):
            if it == 2:
                e0.record()
            plan.stream_collide_twice_planes(a, b, 0.6, 2, n2 - 2); a, b = b, a
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out.setdefault(nz, []).append(ms / (2 * nz * 512 * 512) * 1e6)       # ns per 1000 updates... keep relative
        del a, b, plan
print(json.dumps({"ps_per_update_by_nz": {k: round(sorted(v)[1] * 1e3, 3) for k, v in out.items()},
                  "glups_by_nz": {k: round(1e-3 / sorted(v)[1] * 1e3, 2) for k, v in out.items()}}))
