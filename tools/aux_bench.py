"""Rates of the observable / field kernels (lt_kinetic_energy, lt_mass, lt_max_velocity,
lt_macroscopic, lt_equilibrium) at 256^3, HIP events.  Development tool."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan


def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, dt in (("D3Q19", torch.float32), ("D3Q19", torch.float64), ("D3Q27", torch.float32)):
    res = [256] * 3
    plan = Plan(name, dt, "bgk", res, [], device=torch.device("cuda:0"))
    q, es, n = plan.q, (4 if dt == torch.float32 else 8), 256 ** 3
    f = torch.rand(plan.f_shape, device="cuda", dtype=dt) * 0.01 + 0.05
    rho, u = plan.macroscopic(f)
    out = {}
    for label, fn, nbytes in (
            ("kinetic_energy", lambda: plan.kinetic_energy_lu(f), q * es * n),
            ("mass", lambda: plan.mass(f), q * es * n),
            ("max_velocity", lambda: plan.max_velocity_lu(f), q * es * n),
            ("macroscopic", lambda: plan.macroscopic(f), (q + 4) * es * n),
            ("equilibrium", lambda: plan.equilibrium(rho, u), (q + 4) * es * n)):
        ms = timeit(fn)
        out[label] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1)}
    print(json.dumps({"case": f"{name} {str(dt)[6:]} 256^3", **out}), flush=True)
    del f, rho, u, plan
