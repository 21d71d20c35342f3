"""lbm3_kernel (three updates per launch, both intermediate states in LDS) against three launches of the one-step kernel
(bit-identity) and against the two-step kernel (time per update): dense and padded (resident-like) buffers."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan
dev = torch.device("cuda:0")


def field(plan, dtype, seed=1):
    torch.manual_seed(seed)
    w = torch.rand(plan.q, 1, 1, 1, device=dev, dtype=dtype) * 0.05 + 0.02
    return (w * (1 + 0.1 * torch.rand(plan.f_shape, device=dev, dtype=dtype))).contiguous()


def check(stencil, dtype, coll, res, seg=0):
    plan = Plan(stencil, dtype, coll, res, [], device=dev)
    plan.set_two_step(1, seg)
    f = field(plan, dtype)
    a, b = f.clone(), torch.empty_like(f)
    for _ in range(3):
        plan.stream_collide(a, b, 0.6)
        a, b = b, a
    out = torch.empty_like(f)
    plan.stream_collide_thrice(f, out, 0.6)
    torch.cuda.synchronize()
    same = bool(torch.equal(out, a))
    print(json.dumps({"check": [stencil, str(dtype).split(".")[1], coll, res, seg], "bit_identical": same,
                      "max_abs_diff": float((out - a).abs().max())}), flush=True)
    return same


def timed(fn, reps=10):
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


ok = True
for args in (("D3Q19", torch.float32, "none", [8, 8, 64]), ("D3Q19", torch.float32, "bgk", [8, 8, 64]),
             ("D3Q19", torch.float32, "bgk", [6, 12, 128]), ("D3Q19", torch.float32, "bgk", [16, 16, 64], 4),
             ("D3Q19", torch.float64, "bgk", [8, 8, 32]), ("D3Q15", torch.float32, "bgk", [8, 8, 64]),
             ("D3Q19", torch.float32, "bgk", [64, 64, 128])):
    ok = check(*args) and ok
if not ok:
    sys.exit(1)
for stencil, dtype, res in (("D3Q19", torch.float32, [256, 256, 256]), ("D3Q19", torch.float64, [384, 384, 96])):
    plan = Plan(stencil, dtype, "bgk", res, [], device=dev)
    plan.set_two_step(1, 0)
    nodes = res[0] * res[1] * res[2]
    for pad in (0, 32832):
        if pad:
            plan.set_population_stride(-(-(nodes + pad) // 64) * 64)
        f = plan.empty_populations(); f.uniform_(0.04, 0.06)
        g = plan.empty_populations(); g.zero_()
        state = {"a": f, "b": g}

        def run(fn):
            def step():
                fn(state["a"], state["b"], 0.6)
                state["a"], state["b"] = state["b"], state["a"]
            return step
        t3 = timed(run(plan.stream_collide_thrice))
        t2 = timed(run(plan.stream_collide_twice))
        t1 = timed(run(plan.stream_collide))
        print(json.dumps({"grid": res, "dtype": str(dtype).split(".")[1], "pad": pad,
                          "three_step_ms_per_launch": round(t3, 4), "two_step_ms_per_launch": round(t2, 4),
                          "one_step_ms_per_launch": round(t1, 4),
                          "ms_per_update": [round(t3 / 3, 4), round(t2 / 2, 4), round(t1, 4)],
                          "glups": [round(3 * nodes / t3 / 1e6, 1), round(2 * nodes / t2 / 1e6, 1), round(nodes / t1 / 1e6, 1)]}), flush=True)
        del f, g
        torch.cuda.empty_cache()
