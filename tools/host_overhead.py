"""Host-side cost per step of the slab driver and of the plain driver (tiny grid => GPU time ~ 0)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt
ctx = lt.Context("cuda:0", torch.float32, True)
res = [32, 32, 8]
slab = lt.ZSlab(res, 0, 1)
for overlap in (True, False):
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 100, 0.1, lt.D3Q19(), slab=slab, initialize_fneq=False)
    sim = lt.SlabSimulation(flow, lt.BGKCollision(0.6), slab, overlap=overlap)
    sim(50); torch.cuda.synchronize(); t0 = time.perf_counter(); sim(2000); dt = time.perf_counter() - t0
    print(json.dumps({"slab_overlap": overlap, "us_per_step": round(dt / 2000 * 1e6, 1)}), flush=True)
flow = lt.TaylorGreenVortex(ctx, res, 100, 0.1, lt.D3Q19(), initialize_fneq=False)
sim = lt.Simulation(flow, lt.BGKCollision(0.6), [])
sim(50); t0 = time.perf_counter(); sim(2000); dt = time.perf_counter() - t0
print(json.dumps({"plain_lt_run": True, "us_per_step": round(dt / 2000 * 1e6, 1)}), flush=True)
