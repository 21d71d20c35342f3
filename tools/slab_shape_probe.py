"""Why is the two-step sweep slower per plane on a 512 x 512 x 64 slab than on the 256^3 block?  (round 3)
Slab-layout plans with the same number of nodes and different plane shapes, the whole slab in one two-step launch,
ms per launch and microseconds per workgroup and phase unit (one intermediate or one output plane of a tile)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan, LAYOUT_SLAB

dev = torch.device("cuda:0")


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(plan, f, g, reps=8):
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


import sys as _sys
SHAPES = ((512, 512, 64, 64, 32832), (512, 504, 64, 64, 32832), (512, 520, 64, 64, 32832), (512, 528, 64, 64, 32832),
          (512, 496, 64, 64, 32832), (512, 544, 64, 64, 32832), (576, 512, 64, 64, 32832), (448, 512, 64, 64, 32832),
          (512, 512, 64, 64, 32832)) if len(_sys.argv) > 1 and _sys.argv[1] == "planes" else None
for nx, ny, nz, seg, pad in SHAPES or ((256, 256, 256, 128, 0), (256, 256, 256, 128, 32832), (512, 256, 128, 64, 0), (512, 256, 128, 64, 32832),
                             (512, 512, 64, 64, 0), (512, 512, 64, 64, 32832), (512, 512, 64, 32, 32832),
                             (1024, 512, 32, 32, 32832), (1024, 1024, 16, 16, 32832), (256, 512, 128, 64, 32832),
                             (256, 1024, 64, 64, 32832), (128, 1024, 128, 64, 32832), (64, 2048, 128, 64, 32832)):
    plan = Plan("D3Q19", torch.float32, "bgk", [nx, ny, nz], [], layout=LAYOUT_SLAB, ghost_planes=2, device=dev)
    nodes = nx * ny * (nz + 4)
    if pad:
        plan.set_population_stride(-(-(nodes + pad) // 64) * 64)
    plan.set_two_step(1, seg)
    f = plan.empty_populations()
    f.uniform_(0.05, 0.06)
    g = plan.empty_populations()
    g.zero_()
    ms = timed(plan, f, g)
    tiles = (nx // 64) * (ny // 8)
    segs = -(-nz // seg)
    wgs = tiles * segs
    rounds = -(-wgs // 256)
    units = 2 * seg + 2
    print(json.dumps({"slab": [nx, ny, nz], "plane_KiB": nx * ny * 4 // 1024, "seg": seg, "pad": pad, "workgroups": wgs, "rounds": rounds,
                      "ms_per_launch": round(ms, 4), "glups": round(2 * nx * ny * nz / ms / 1e6, 2),
                      "us_per_wg_unit": round(ms * 1e3 / rounds / units, 3)}), flush=True)
    del plan, f, g
    torch.cuda.empty_cache()
