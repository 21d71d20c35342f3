"""cfg4 (Obstacle3D D3Q27 256^3 KBC fp32) with several builds of the engine library in ONE process, batches alternating:
ms per update of the fused launches.  Every build gets its own simulation (own buffers: +- 1-2 % from placement), so
each library is given twice.   usage: cfg4_same_process_ab.py other.so [other2.so ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt
import lettuce_amd._native as nat


def build(path):
    nat._LIB = None
    if path:
        os.environ["LT_ENGINE_LIBRARY"] = path
    else:
        os.environ.pop("LT_ENGINE_LIBRARY", None)
    ctx = lt.Context("cuda:0", torch.float32, True)
    flow = lt.Obstacle(ctx, [256, 256, 256], 100, 0.1, domain_length_x=4, stencil=lt.D3Q27())
    x, y, z = flow.grid
    flow.mask = ((x - 1) ** 2 + (y - 2) ** 2 + (z - 2) ** 2) < 0.5 ** 2
    flow.initialize()
    sim = lt.Simulation(flow, lt.KBCCollision(), [])
    sim(5)
    return sim


def timed(sim, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sim._native.fused_events = (e0, e1)
    sim(steps); torch.cuda.synchronize()
    info = sim._native.plan.last_run_info(); sim._native.fused_events = None
    return e0.elapsed_time(e1) / max(1, info["single_step_launches"] + 2 * info["two_step_launches"])


libs = [""] + sys.argv[1:]
specs = libs + libs                                   # every library twice, at different times of the process
sims = [build(p) for p in specs]
times = [[] for _ in specs]
for rep in range(4):
    for k, sim in enumerate(sims):
        times[k].append(round(timed(sim, 40), 4))
for p, t in zip(specs, times):
    print(json.dumps({"lib": os.path.basename(p) or "product", "ms_per_update": t[1:]}), flush=True)
