"""Enqueue (host) time vs completion (device) time per step of the slab driver with RCCL
self-send (LT_SLAB_FORCE_P2P=1).  Run under torchrun with one rank."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import lettuce_amd as lt
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
ctx = lt.Context("cuda:0", torch.float32, True)
for res in ([512, 512, 64], [64, 64, 8]):
    slab = lt.ZSlab(res)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab, initialize_fneq=False)
    sim = lt.SlabSimulation(flow, lt.BGKCollision(0.53), slab)
    sim(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); sim._advance(200); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(json.dumps({"res": res, "force_p2p": sim._force_p2p, "enqueue_us_per_step": round((t1 - t0) / 200 * 1e6, 1),
                      "total_us_per_step": round((t2 - t0) / 200 * 1e6, 1)}), flush=True)
dist.destroy_process_group()
