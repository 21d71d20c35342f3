"""lt_stream_collide_twice: bit-identity with two single steps, and timing against them."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan


def ev():
    return torch.cuda.Event(enable_timing=True)


def check(res, coll, seg):
    plan = Plan("D3Q19", torch.float32, coll, res, [], device=torch.device("cuda:0"))
    plan.set_two_step(1, seg)
    torch.manual_seed(3)
    w = torch.rand(19, 1, 1, 1, device="cuda") * 0.05 + 0.02
    f = (w * (1 + 0.1 * torch.rand(plan.f_shape, device="cuda"))).contiguous()
    a, b = torch.empty_like(f), torch.empty_like(f)
    plan.stream_collide(f, a, 0.6); plan.stream_collide(a, b, 0.6)
    c = torch.empty_like(f)
    plan.stream_collide_twice(f, c, 0.6)
    torch.cuda.synchronize()
    same = bool(torch.equal(b, c))
    print(json.dumps({"check": res, "collision": coll, "seg": seg, "bit_identical": same,
                      "max_abs_diff": float((b - c).abs().max())}), flush=True)
    return same


ok = True
for res, seg in (([4, 8, 64], 0), ([8, 16, 64], 4), ([6, 24, 128], 3), ([1, 8, 64], 1), ([12, 40, 192], 0)):   # [x, y, z], z contiguous
    for coll in ("none", "bgk"):
        ok &= check(res, coll, seg)
if not ok:
    sys.exit(1)

res = [256] * 3
plan = Plan("D3Q19", torch.float32, "bgk", res, [], device=torch.device("cuda:0"))
f = torch.rand(plan.f_shape, device="cuda") * 0.01 + 0.05
g = torch.empty_like(f)
out = {}
variants = [("single", -1, 0)] + [(f"v{v}_seg{seg}", seg, v) for v in (0, 2) for seg in (64, 128)]
for r in range(5):
    for label, seg, variant in variants:
        if seg > 0:
            plan.set_two_step(1, seg)
            plan.set_shift_policy(variant)
        e0, e1 = ev(), ev()
        a, b = f, g
        for it in range(12):
            if it == 2:
                e0.record()
            if seg < 0:
                plan.stream_collide(a, b, 0.6); a, b = b, a
                plan.stream_collide(a, b, 0.6); a, b = b, a
            else:
                plan.stream_collide_twice(a, b, 0.6); a, b = b, a
        e1.record(); torch.cuda.synchronize()
        out.setdefault(label, []).append(e0.elapsed_time(e1) / 20)
plan.set_shift_policy(0)
none = Plan("D3Q19", torch.float32, "none", res, [], device=torch.device("cuda:0"))
none.set_two_step(1, 128)
for r in range(5):
    for label, fn in (("none_single", lambda a, b: (none.stream_collide(a, b, 1.0), none.stream_collide(b, a, 1.0))),
                      ("none_twice", lambda a, b: (none.stream_collide_twice(a, b, 1.0), none.stream_collide_twice(b, a, 1.0)))):
        e0, e1 = ev(), ev()
        fn(f, g)
        e0.record()
        for it in range(5):
            fn(f, g)
        e1.record(); torch.cuda.synchronize()
        out.setdefault(label, []).append(e0.elapsed_time(e1) / (20 if label == "none_twice" else 10))
print(json.dumps({"ms_per_step": {k: round(sorted(v)[2], 4) for k, v in out.items()}}), flush=True)
