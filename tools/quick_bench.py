"""Kernel-level timing of the fused stream-collide kernel (variants interleaved in one
process, HIP events on the launch stream).  Development tool, not the judged bench."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

def time_variant(plan, a, b, tau, iters):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        plan.stream_collide(a, b, tau); a, b = b, a
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(iters):
        plan.stream_collide(a, b, tau); a, b = b, a
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / iters

def main():
    cases = [("D3Q19", torch.float32, "bgk", [256] * 3), ("D3Q19", torch.float64, "bgk", [256] * 3),
             ("D3Q27", torch.float32, "bgk", [256] * 3), ("D3Q27", torch.float32, "kbc", [256] * 3),
             ("D3Q19", torch.float32, "bgk", [512, 512, 64]), ("D3Q19", torch.float64, "bgk", [384, 384, 96]),
             ("D2Q9", torch.float64, "bgk", [4096, 4096])]
    rounds = int(os.environ.get("ROUNDS", 3))
    for lat, dt, coll, res in cases:
        q = int(lat.split("Q")[1])
        plan = Plan(lat, dt, coll, res)
        n = 1
        for r in res: n *= r
        w = 1.0 / q
        a = torch.full([q] + res, w, dtype=dt, device="cuda") * (1 + 0.01 * torch.rand([q] + res, dtype=dt, device="cuda"))
        b = torch.empty_like(a)
        esize = 4 if dt == torch.float32 else 8
        policies = (0, 1, 2) if coll == "bgk" else (0,)
        best = {}
        for rnd in range(rounds):
            for pol in policies:
                plan.set_shift_policy(pol)
                ms = time_variant(plan, a, b, 0.6, 20)
                best.setdefault(pol, []).append(ms)
        for pol in policies:
            ms = sorted(best[pol])[len(best[pol]) // 2]
            mlups = n / ms / 1e3
            gbs = mlups * 1e6 * 2 * q * esize / 1e9
            print(json.dumps({"lattice": lat, "dtype": str(dt), "coll": coll, "res": res, "shift": pol,
                              "ms": round(ms, 4), "min_ms": round(min(best[pol]), 4), "mlups": round(mlups, 1), "GBps": round(gbs, 1),
                              "frac_8TBs": round(gbs / 8000, 3), "kernel": plan.kernel_name()}), flush=True)
        # copy ceiling for reference: out-of-place copy of the same bytes
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); ev0.record()
        for _ in range(10): b.copy_(a)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 10
        print(json.dumps({"copy_same_bytes_ms": round(ms, 4), "GBps": round(2 * a.numel() * esize / ms / 1e6, 1)}), flush=True)
        del a, b, plan

if __name__ == "__main__":
    main()
