"""The slab rehearsal's candidates built one after the other in ONE process, in the order given on the command line
(round 4: the copy transport ran 20 % slower after the RCCL candidates than alone; which predecessor does that, and does
it last?).  usage: slab_order_probe.py two-step/rccl two-step/copy two-step/rccl ...   (one rank, 512 x 512 x 64)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LT_SLAB_FORCE_P2P", "1")
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
import lettuce_amd as lt
ctx = lt.Context(device=dev, dtype=torch.float32, use_native=True)
for name in sys.argv[1:]:
    driver, transport = name.split("/")
    slab = lt.ZSlab([512, 512, 64])
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab)
    coll = lt.BGKCollision(flow.units.relaxation_parameter_lu)
    if driver == "two-step":
        sim = lt.TwoStepSlabSimulation(flow, coll, slab, transport=transport, direct=True)
    else:
        sim = lt.SlabSimulation(flow, coll, slab, transport=transport)
    sim(23)
    out = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim(100)
        torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) / 100 * 1e3, 5))
    print(json.dumps({"candidate": name, "ms_per_step": out, "free_GiB": round(torch.cuda.mem_get_info()[0] / 2**30, 1)}), flush=True)
    del sim, flow, coll, slab
    if os.environ.get("LT_PROBE_KEEP_CACHE") != "1":
        torch.cuda.empty_cache()
dist.destroy_process_group()
