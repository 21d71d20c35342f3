"""The A/B variants of the two-step kernel (lt_plan_set_shift_policy: 0 product, 1 two nodes per thread in both phases,
2 two output nodes per thread, 3 no XCD-aware numbering, 4 the round-1 numbering) on the SAME buffers in one process,
launches alternating (cfg2: 256^3 D3Q19 fp32, padded buffers); also segment lengths."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan
dev = torch.device("cuda:0")
plan = Plan("D3Q19", torch.float32, "bgk", [256, 256, 256], [], device=dev)
plan.set_population_stride(-(-(256 ** 3 + 32832) // 64) * 64)
f = plan.empty_populations(); f.uniform_(0.04, 0.06)
g = plan.empty_populations(); g.zero_()
cases = [("product", 0, 0), ("two nodes per thread", 1, 0), ("two output nodes per thread", 2, 0), ("no XCD numbering", 3, 0),
         ("round-1 numbering", 4, 0), ("64 planes per workgroup", 0, 64), ("256 planes per workgroup", 0, 256), ("32 planes", 0, 32)]
times = {c[0]: [] for c in cases}
ref = None
for name, policy, seg in cases:
    plan.set_two_step(1, seg); plan.set_shift_policy(policy)
    plan.stream_collide_twice(f, g, 0.6); torch.cuda.synchronize()
    if ref is None:
        ref = g.clone()
    elif not torch.equal(g, ref):
        print(json.dumps({"case": name, "MISMATCH": True}))
for rep in range(5):
    for name, policy, seg in cases:
        plan.set_two_step(1, seg); plan.set_shift_policy(policy)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        plan.stream_collide_twice(f, g, 0.6)
        e0.record()
        for _ in range(10):
            plan.stream_collide_twice(f, g, 0.6)
            plan.stream_collide_twice(g, f, 0.6)
        e1.record(); torch.cuda.synchronize()
        times[name].append(round(e0.elapsed_time(e1) / 20, 4))
for name, t in times.items():
    print(json.dumps({"case": name, "ms_per_launch": t[1:]}), flush=True)
