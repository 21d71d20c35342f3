"""Two-step kernel on the other 3-D lattices / dtypes: bit-identity and ms per update."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan, NativeEngineError

def ev(): return torch.cuda.Event(enable_timing=True)
for lat, dt in (("D3Q27", torch.float32), ("D3Q15", torch.float32), ("D3Q15", torch.float64), ("D3Q19", torch.float64), ("D3Q19", torch.float32)):
    res = [256, 256, 256]
    plan = Plan(lat, dt, "bgk", res, [], device=torch.device("cuda:0"))
    f = torch.rand(plan.f_shape, device="cuda", dtype=dt) * 0.01 + 0.05
    a, b, c = torch.empty_like(f), torch.empty_like(f), torch.empty_like(f)
    plan.stream_collide(f, a, 0.6); plan.stream_collide(a, b, 0.6)
    try:
        plan.stream_collide_twice(f, c, 0.6)
    except NativeEngineError as exc:
        print(json.dumps({"lattice": lat, "dtype": str(dt)[6:], "unsupported": str(exc)[:80]})); continue
    torch.cuda.synchronize()
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
    print(json.dumps({"lattice": lat, "dtype": str(dt)[6:], "bit_identical": same, "kernel": plan.kernel_name(),
                      "ms_per_update": {k: round(sorted(v)[2], 4) for k, v in out.items()}}), flush=True)
    del f, a, b, c, plan
