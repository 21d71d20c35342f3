"""Kernel-level timing of the fused stream-collide kernel and of a plain copy (variants
interleaved in one process, HIP events on the launch stream).  Development tool."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan, probe_copy

def ev():
    return torch.cuda.Event(enable_timing=True)

def time_fused(plan, a, b, tau, iters):
    e0, e1 = ev(), ev()
    for _ in range(2):
        plan.stream_collide(a, b, tau); a, b = b, a
    e0.record()
    for _ in range(iters):
        plan.stream_collide(a, b, tau); a, b = b, a
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def time_copy(a, b, pol, cap, iters):
    e0, e1 = ev(), ev()
    probe_copy(b, a, pol, cap)
    e0.record()
    for _ in range(iters):
        probe_copy(b, a, pol, cap)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def med(x):
    x = sorted(x); return x[len(x) // 2]

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    rounds = int(os.environ.get("ROUNDS", 3))
    if which in ("all", "copy"):
        a = torch.rand(19 * 256 ** 3, device="cuda"); b = torch.empty_like(a)
        res = {}
        for r in range(rounds):
            for pol in (0, 1, 2, 3):
                for cap in (0, 2048, 4096, 8192):
                    res.setdefault((pol, cap), []).append(time_copy(a, b, pol, cap, 10))
        for (pol, cap), v in sorted(res.items()):
            print(json.dumps({"copy_policy": pol, "max_blocks": cap, "ms": round(med(v), 4),
                              "GBps": round(2 * a.numel() * 4 / med(v) / 1e6, 1)}), flush=True)
        e0, e1 = ev(), ev(); e0.record()
        for _ in range(10): b.copy_(a)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"torch_copy_GBps": round(2 * a.numel() * 4 / (e0.elapsed_time(e1) / 10) / 1e6, 1)}), flush=True)
        del a, b
    if which in ("all", "fused"):
        cases = [("D3Q19", torch.float32, [256] * 3, "bgk"), ("D3Q19", torch.float64, [256] * 3, "bgk"),
                 ("D3Q27", torch.float32, [256] * 3, "bgk"), ("D3Q27", torch.float32, [256] * 3, "kbc"),
                 ("D3Q19", torch.float64, [384, 384, 96], "bgk"), ("D2Q9", torch.float64, [4096, 4096], "bgk")]
        for lat, dt, rs, coll in cases:
            q = int(lat.split("Q")[1]); n = 1
            for r_ in rs: n *= r_
            es = 4 if dt == torch.float32 else 8
            plan = Plan(lat, dt, coll, rs)
            a = torch.full([q] + rs, 1.0 / q, dtype=dt, device="cuda") * (1 + 0.01 * torch.rand([q] + rs, dtype=dt, device="cuda"))
            b = torch.empty_like(a)
            # (wide, shift, cache policy, max blocks)
            if coll != "bgk":
                ms = med([time_fused(plan, a, b, 0.6, 20) for _ in range(rounds)])
                print(json.dumps({"lattice": lat, "dtype": str(dt), "coll": coll, "res": rs, "ms": round(ms, 4),
                                  "mlups": round(n / ms / 1e3, 1), "GBps": round(n / ms / 1e3 * 2 * q * es / 1e3, 1)}), flush=True)
                del a, b, plan
                continue
            variants = [(0, 0, -1, 0), (0, 0, 0, 0), (0, 0, 3, 0), (0, 0, 3, 8192), (1, 0, 0, 0), (1, 1, 0, 0),
                        (1, 2, 0, 0), (1, 0, 2, 0), (1, 2, 3, 0)]
            res = {}
            for r in range(rounds):
                for wide, shift, tune, cap in variants:
                    plan.set_shift_policy(shift); plan.set_tuning(tune, bool(wide))
                    res.setdefault((wide, shift, tune, cap), []).append(time_fused(plan, a, b, 0.6, 20))
            for (wide, shift, tune, cap), v in res.items():
                ms = med(v); mlups = n / ms / 1e3
                print(json.dumps({"lattice": lat, "dtype": str(dt), "res": rs, "wide16B": wide, "shift": shift, "nt": tune, "max_blocks": cap,
                                  "ms": round(ms, 4), "min_ms": round(min(v), 4), "mlups": round(mlups, 1),
                                  "GBps": round(mlups * 2 * q * es / 1e3, 1)}), flush=True)
            del a, b, plan

if __name__ == "__main__":
    main()
