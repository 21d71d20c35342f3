"""The slab rehearsal (512 x 512 x 64, direct two-step schedule, halo messages through RCCL to the rank itself) with two
builds of the engine library IN ONE PROCESS, batches alternating: ms per step each.  (Separate processes differ by
+- 4 % through clocks and page placement alone.)   usage: slab_same_process_ab.py other.so | ENV=value[,ENV=value] ..."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
os.environ["LT_SLAB_FORCE_P2P"] = "1"
import torch
import torch.distributed as dist
import lettuce_amd as lt
import lettuce_amd._native as nat

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = lt.Context(device=torch.device("cuda", 0), dtype=torch.float32, use_native=True)
slab = lt.ZSlab([512, 512, 64])


def build(spec):
    """spec: a library path, or "ENV=value,ENV2=value2" (driver switches read at construction), or "" = the product"""
    nat._LIB = None
    path, env = (spec, {}) if "=" not in spec else ("", dict(kv.split("=", 1) for kv in spec.split(",")))
    if path:
        os.environ["LT_ENGINE_LIBRARY"] = path
    else:
        os.environ.pop("LT_ENGINE_LIBRARY", None)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _build()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _build():
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab)
    sim = lt.TwoStepSlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab, transport="rccl", direct=True)
    sim(6)
    return sim


libs = [""] + sys.argv[1:]
sims = [build(p) for p in libs]
times = {os.path.basename(p) or "product": [] for p in libs}
for rep in range(6):
    for p, sim in zip(libs, sims):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim(200)
        torch.cuda.synchronize()
        times[os.path.basename(p) or "product"].append(round((time.perf_counter() - t0) / 200 * 1e3, 4))
print(json.dumps({"ms_per_step": times}))
same = torch.equal(sims[0].local_f(), sims[1].local_f()) if len(sims) > 1 else None
print(json.dumps({"bit_identical_after_the_same_number_of_steps": same}))
dist.destroy_process_group()
