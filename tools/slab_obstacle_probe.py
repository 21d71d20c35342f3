"""Obstacle (inlet, anti-bounce-back outlet along x, sphere bounce-back) on ONE rank's z-slab exchanging its
ghost planes with itself through RCCL: the one-exchange-per-step slab driver against the two-step slab driver
(lbm2m_kernel AX = 0 on a plan with two ghost planes).  Dev tool; one JSON line per driver."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29512")
os.environ["LT_SLAB_FORCE_P2P"] = "1"
import torch
import torch.distributed as dist
import lettuce_amd as lt


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    ctx = lt.Context("cuda:0", torch.float32, True)
    cases = [("D3Q19", [512, 512, 64]), ("D3Q27", [512, 512, 64]), ("D3Q19", [256, 256, 256])]
    if len(sys.argv) > 1:
        cases = cases[:int(sys.argv[1])]
    for lattice, res in cases:
        finals = {}
        for driver in ("SlabSimulation", "TwoStepSlabSimulation"):
            slab = lt.ZSlab(res, 0, 1)
            flow = lt.Obstacle(ctx, slab.extended_resolution, 100, 0.1, domain_length_x=4, stencil=getattr(lt, lattice)(), slab=slab)
            x, y, z = flow.grid
            flow.mask = ((x - 1) ** 2 + (y - 2) ** 2 + (z - 0.25) ** 2) < 0.4 ** 2
            flow.initialize()
            sim = getattr(lt, driver)(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
            sim(21)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sim(200)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            n = res[0] * res[1] * res[2]
            f = sim.gather_f()
            finals[driver] = f
            print(json.dumps({"flow": f"Obstacle {lattice} BGK fp32", "res": res, "driver": driver, "kernel": sim.engine.kernel_name(),
                              "ms_per_step": round(dt / 200 * 1e3, 4), "mlups": round(200 * n / dt / 1e6, 1),
                              "finite": bool(torch.isfinite(f).all())}), flush=True)
            del sim, flow
            torch.cuda.empty_cache()
        print(json.dumps({"bit_identical": bool(torch.equal(finals["SlabSimulation"], finals["TwoStepSlabSimulation"]))}), flush=True)
        del finals
        torch.cuda.empty_cache()
    dist.destroy_process_group()


main()
