"""Fused-kernel rate of every lattice x dtype x collision (periodic, large grids).  One JSON line each."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

def main():
    cases = []
    for lat, res in (("D2Q9", [4096, 4096]), ("D3Q19", [256] * 3), ("D3Q27", [256] * 3)):
        for dt in (torch.float32, torch.float64):
            for coll in ("none", "bgk", "kbc"):
                if coll == "kbc" and lat == "D3Q19": continue
                cases.append((lat, res, dt, coll))
    for lat, res, dt, coll in cases:
        q = int(lat.split("Q")[1]); n = 1
        for r in res: n *= r
        es = 4 if dt == torch.float32 else 8
        plan = Plan(lat, dt, coll, res)
        a = torch.full([q] + res, 1.0 / q, dtype=dt, device="cuda") * (1 + 0.01 * torch.rand([q] + res, dtype=dt, device="cuda"))
        b = torch.empty_like(a)
        ms = []
        for r in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            plan.stream_collide(a, b, 0.6); a, b = b, a
            e0.record()
            for _ in range(20):
                plan.stream_collide(a, b, 0.6); a, b = b, a
            e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 20)
        m = sorted(ms)[1]
        gbs = 2 * q * es * n / m / 1e6
        print(json.dumps({"lattice": lat, "dtype": "f32" if es == 4 else "f64", "collision": coll, "res": res,
                          "ms": round(m, 4), "mlups": round(n / m / 1e3, 1), "GBps": round(gbs, 1),
                          "frac_8TBs": round(gbs / 8000, 3)}), flush=True)
        del a, b, plan
main()
