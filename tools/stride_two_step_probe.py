"""How much does the distance between populations matter for the two-step kernel?  (round 3, before padding)

Slab-layout plans (512 x 512 x nz + 4 ghost planes: the stride between populations is nz + 4 MiB in fp32) and
reference-layout plans (nx x 256 x 256: stride nx / 4 MiB), one workgroup per tile and segment as the product
launches do; ms per intermediate plane, so that grids of different depth compare.  Results are not checked (the
ghost planes hold noise); only the launch time matters here.
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan, LAYOUT_SLAB


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(plan, f, g, reps=10):
    best = 1e9
    for _ in range(3):
        e0, e1 = ev(), ev()
        plan.stream_collide_twice(f, g, 0.6)
        e0.record()
        for _ in range(reps):
            plan.stream_collide_twice(f, g, 0.6)
            plan.stream_collide_twice(g, f, 0.6)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (2 * reps))
    return best


dev = torch.device("cuda:0")
for nz in (60, 61, 62, 63, 64, 65, 66, 68):
    plan = Plan("D3Q19", torch.float32, "bgk", [512, 512, nz], [], layout=LAYOUT_SLAB, ghost_planes=2, device=dev)
    plan.set_two_step(1, nz)
    f = torch.rand(plan.f_shape, device=dev) * 0.01 + 0.05
    g = torch.empty_like(f)
    ms = timed(plan, f, g)
    print(json.dumps({"layout": "slab", "res": [512, 512, nz], "pop_stride_MiB": (nz + 4) * 1.0, "ms_per_launch": round(ms, 4),
                      "us_per_plane": round(ms * 1e3 / (nz + 2), 3),
                      "glups": round(2 * 512 * 512 * nz / ms / 1e6, 2)}), flush=True)
    del plan, f, g
    torch.cuda.empty_cache()

for nx in (256, 258, 260, 264, 272):
    plan = Plan("D3Q19", torch.float32, "bgk", [nx, 256, 256], [], device=dev)
    plan.set_two_step(1, nx // 2)
    f = torch.rand(plan.f_shape, device=dev) * 0.01 + 0.05
    g = torch.empty_like(f)
    ms = timed(plan, f, g)
    print(json.dumps({"layout": "reference", "res": [nx, 256, 256], "pop_stride_MiB": nx / 4, "ms_per_launch": round(ms, 4),
                      "us_per_plane": round(ms * 1e3 / (nx + 4), 3),
                      "glups": round(2 * nx * 65536 / ms / 1e6, 2)}), flush=True)
    # the one-step kernel on the same grid
    plan.set_two_step(0, 0)
    e0, e1 = ev(), ev()
    plan.stream_collide(f, g, 0.6)
    e0.record()
    for _ in range(10):
        plan.stream_collide(f, g, 0.6); plan.stream_collide(g, f, 0.6)
    e1.record(); torch.cuda.synchronize()
    ms1 = e0.elapsed_time(e1) / 20
    print(json.dumps({"layout": "reference", "res": [nx, 256, 256], "kernel": "one-step", "ms_per_launch": round(ms1, 4),
                      "us_per_plane": round(ms1 * 1e3 / nx, 3)}), flush=True)
    del plan, f, g
    torch.cuda.empty_cache()
