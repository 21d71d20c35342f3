"""Does running several slab simulations one after another in one process slow the later ones down?
(bench.py --gpus N probes several driver/transport candidates before the timed run.)"""
import sys, os, json, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("NCCL_DEBUG", "WARN")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import lettuce_amd as lt

ctx = lt.Context(device="cuda:0", dtype=torch.float32, use_native=True)
res = [512, 512, 64]
seq = sys.argv[1:] or ["two-step/rccl"] * 3 + ["two-step/window", "two-step/rccl", "two-step/window", "two-step/rccl"]
for name in seq:
    driver, transport = name.split("/")
    if transport == "p2p":
        os.environ["LT_SLAB_FORCE_P2P"] = "1"; transport = "rccl"
    else:
        os.environ.pop("LT_SLAB_FORCE_P2P", None)
    slab = lt.ZSlab(res)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab)
    cls = lt.TwoStepSlabSimulation if driver == "two-step" else lt.SlabSimulation
    sim = cls(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab, transport=transport)
    sim(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); sim(60); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"config": name, "ms_per_step": round(dt / 60 * 1e3, 4),
                      "allocated_GB": round(torch.cuda.memory_allocated() / 2**30, 2),
                      "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2)}), flush=True)
    del sim, flow
    gc.collect()
    if os.environ.get('EMPTY_CACHE', '1') == '1':
        torch.cuda.empty_cache()
dist.destroy_process_group()
