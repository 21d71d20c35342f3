"""Two-step slab step broken into its launches (512 x 512 x 64 slab, one stream, HIP events)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan, LAYOUT_SLAB

plan = Plan("D3Q19", torch.float32, "bgk", [512, 512, 64], [], layout=LAYOUT_SLAB, ghost_planes=2)
a = torch.rand(plan.f_shape, device="cuda") * 0.01 + 0.05
b = torch.empty_like(a)
n2 = a.shape[1]
msg = torch.empty([19, 512, 512], device="cuda")


def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


for _ in range(200):                                   # clocks up
    plan.stream_collide_twice_planes(a, b, 0.6, 2, n2 - 2)
torch.cuda.synchronize()
out = {
    "all_64_planes": timed(lambda: plan.stream_collide_twice_planes(a, b, 0.6, 2, n2 - 2)),
    "interior_60": timed(lambda: plan.stream_collide_twice_planes(a, b, 0.6, 4, n2 - 4)),
    "two_edges": timed(lambda: (plan.stream_collide_twice_planes(a, b, 0.6, 2, 4),
                                plan.stream_collide_twice_planes(a, b, 0.6, n2 - 4, n2 - 2))),
    "two_edges_packed": timed(lambda: (plan.stream_collide_twice_planes_packed(a, b, 0.6, 2, 4, pack_lower=msg),
                                       plan.stream_collide_twice_planes_packed(a, b, 0.6, n2 - 4, n2 - 2, pack_upper=msg))),
    "pack_x2": timed(lambda: (plan.pack_two_step(b, -1, msg), plan.pack_two_step(b, +1, msg))),
    "unpack_x2": timed(lambda: (plan.unpack_two_step(b, -1, msg), plan.unpack_two_step(b, +1, msg))),
    "edges+interior": timed(lambda: (plan.stream_collide_twice_planes(a, b, 0.6, 2, 4),
                                     plan.stream_collide_twice_planes(a, b, 0.6, n2 - 4, n2 - 2),
                                     plan.stream_collide_twice_planes(a, b, 0.6, 4, n2 - 4))),
}
plan.set_two_step(1, 64)
out["all_64_planes_seg64"] = timed(lambda: plan.stream_collide_twice_planes(a, b, 0.6, 2, n2 - 2))
for seg in (15, 30, 60):
    plan.set_two_step(1, seg)
    out[f"interior_60_seg{seg}"] = timed(lambda: plan.stream_collide_twice_planes(a, b, 0.6, 4, n2 - 4))
print(json.dumps(out))
