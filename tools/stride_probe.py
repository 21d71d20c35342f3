"""Does a power-of-two population stride (256^3) cost bandwidth?  Fused kernel at nearby shapes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan
shapes = [[256, 256, 256], [257, 256, 256], [258, 256, 256], [264, 256, 256], [272, 256, 256], [256, 256, 264], [256, 264, 256], [250, 250, 250], [288, 288, 200]]
for dt in (torch.float32, torch.float64):
    res = {}
    for rnd in range(3):
        for rs in shapes:
            n = rs[0] * rs[1] * rs[2]
            plan = Plan("D3Q19", dt, "bgk", rs)
            a = torch.full([19] + rs, 1.0 / 19, dtype=dt, device="cuda") * (1 + 0.01 * torch.rand([19] + rs, dtype=dt, device="cuda"))
            b = torch.empty_like(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            plan.stream_collide(a, b, 0.6); a, b = b, a
            e0.record()
            for _ in range(20):
                plan.stream_collide(a, b, 0.6); a, b = b, a
            e1.record(); torch.cuda.synchronize()
            res.setdefault(tuple(rs), []).append(e0.elapsed_time(e1) / 20)
            del a, b, plan
    es = 4 if dt == torch.float32 else 8
    for rs, v in res.items():
        m = sorted(v)[1]; n = rs[0] * rs[1] * rs[2]
        print(json.dumps({"dtype": "f32" if es == 4 else "f64", "res": list(rs), "ms": round(m, 4), "mlups": round(n / m / 1e3, 1),
                          "GBps": round(2 * 19 * es * n / m / 1e6, 1)}), flush=True)
