"""A/B of builds of the engine library IN ONE PROCESS ON THE SAME BUFFERS: the two-step kernel at 256^3 (padded
buffers) launched alternately from each library.  Separate processes get different physical pages, and that alone moves
a launch by +- 3 % (round 3: the position of a run in a sequence of bench.py runs decided its time, reproducibly).
usage: same_buffer_ab.py [slab | cfg5] other1.so [other2.so ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd._native as nat

dev = torch.device("cuda:0")


SLAB = len(sys.argv) > 1 and sys.argv[1] == "slab"      # the per-rank slab of cfg3 (512 x 512 x 64 + ghost planes) instead
if SLAB:
    del sys.argv[1]
CFG5 = len(sys.argv) > 1 and sys.argv[1] == "cfg5"      # ... or cfg5's per-GPU shape in fp64 (384 x 384 x 96)
if CFG5:
    del sys.argv[1]


def plan_from(path):
    nat._LIB = None
    if path:
        os.environ["LT_ENGINE_LIBRARY"] = path
    else:
        os.environ.pop("LT_ENGINE_LIBRARY", None)
    if SLAB:
        plan = nat.Plan("D3Q19", torch.float32, "bgk", [512, 512, 64], [], layout=nat.LAYOUT_SLAB, ghost_planes=2, device=dev)
        plan.set_two_step(1, 64)
        plan.set_population_stride(-(-(512 * 512 * 68 + 32832) // 64) * 64)
        return plan
    if CFG5:
        plan = nat.Plan("D3Q19", torch.float64, "bgk", [384, 384, 96], [], device=dev)
        plan.set_two_step(1, 0)
        plan.set_population_stride(-(-(384 * 384 * 96 + 32832) // 64) * 64)
        return plan
    plan = nat.Plan("D3Q19", torch.float32, "bgk", [256, 256, 256], [], device=dev)
    if os.environ.get("LT_AB_ARITH"):                # "fast": A/B of variants of the fast BGK arithmetic
        plan.set_arithmetic(os.environ["LT_AB_ARITH"])
    plan.set_two_step(1, 0)
    plan.set_population_stride(-(-(256 ** 3 + 32832) // 64) * 64)
    return plan


libs = [""] + sys.argv[1:]
plans = [plan_from(p) for p in libs]
for trial in range(2):                       # two sets of buffers
    f = plans[0].empty_populations(); f.uniform_(0.04, 0.06)
    g = plans[0].empty_populations(); g.zero_()
    ref = None
    for name, plan in zip(libs, plans):
        plan.stream_collide_twice(f, g, 0.6)
        torch.cuda.synchronize()
        if ref is None:
            ref = g.clone()
        elif not torch.equal(g, ref):
            print(json.dumps({"lib": name, "MISMATCH": True, "max_abs_diff": float((g - ref).abs().max())}))
    del ref
    times = {name: [] for name in libs}
    for rep in range(5):
        for name, plan in zip(libs, plans):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            plan.stream_collide_twice(f, g, 0.6)
            e0.record()
            for _ in range(10):
                plan.stream_collide_twice(f, g, 0.6)
                plan.stream_collide_twice(g, f, 0.6)
            e1.record(); torch.cuda.synchronize()
            times[name].append(round(e0.elapsed_time(e1) / 20, 4))
    print(json.dumps({"buffers": trial, "ms_per_launch": {os.path.basename(k) or "product": v for k, v in times.items()}}), flush=True)
    del f, g
    torch.cuda.empty_cache()
    junk = torch.empty(3 * 1024 ** 3 // 4, device=dev)      # shift where the next buffers land
