"""The edge launch of the direct slab schedule alone (lt_stream_collide_twice_edges_direct, 512 x 512 x 64): time per
launch by tile shape (shift policy 5 = 32 x 8 tiles, two workgroups per CU) and the sweep in between for scale."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan, LAYOUT_SLAB

dev = torch.device("cuda:0")
plan = Plan("D3Q19", torch.float32, "bgk", [512, 512, 64], [], layout=LAYOUT_SLAB, ghost_planes=2, device=dev)
nodes = 512 * 512 * 68
plan.set_population_stride(-(-(nodes + 32832) // 64) * 64)
f = plan.empty_populations(); f.uniform_(0.05, 0.06)
g = plan.empty_populations(); g.zero_()
msg = [torch.rand([19, 512, 512], device=dev) * 0.01 + 0.05 for _ in range(4)]


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, reps=20):
    best = 1e9
    for _ in range(3):
        fn(); e0, e1 = ev(), ev(); e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


ref = None
for policy in (0, 5):
    plan.set_shift_policy(policy)
    g.zero_()
    plan.stream_collide_twice_edges_direct(f, g, 0.6, 2, msg[0], msg[1], msg[2], msg[3])
    torch.cuda.synchronize()
    state = (g[:, 2:4].clone(), g[:, 64:66].clone(), msg[2].clone(), msg[3].clone())
    if ref is None:
        ref = state
    same = all(torch.equal(a, b) for a, b in zip(state, ref))
    ms = timed(lambda: plan.stream_collide_twice_edges_direct(f, g, 0.6, 2, msg[0], msg[1], msg[2], msg[3]))
    print(json.dumps({"edge_launch": "64 x 8 tiles" if policy == 0 else "32 x 8 tiles", "us_per_launch": round(ms * 1e3, 2),
                      "same_planes_and_messages_as_64x8": same}), flush=True)
plan.set_shift_policy(0)
plan.set_two_step(1, 0)
ms = timed(lambda: plan.stream_collide_twice_planes(f, g, 0.6, 4, 64))
print(json.dumps({"sweep_in_between_us": round(ms * 1e3, 2)}), flush=True)
