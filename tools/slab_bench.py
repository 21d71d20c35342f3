"""Single-GPU timing of the z-slab path (self exchange) against the plain path.  Dev tool."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt

def main():
    ctx = lt.Context("cuda:0", torch.float32, True)
    for res in ([256, 256, 256], [512, 512, 64], [1024, 1024, 16]):
        slab = lt.ZSlab(res, 0, 1)
        flow = None
        for overlap, prio in ((True, -1), (True, 0), (False, 0)):
            sim = lt.SlabSimulation(lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab, initialize_fneq=False),
                                    lt.BGKCollision(0.53), slab, overlap=overlap, comm_priority=prio)
            sim(20)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sim(200)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            n = res[0] * res[1] * res[2]
            print(json.dumps({"res": res, "overlap": overlap, "comm_priority": prio, "ms_per_step": round(dt / 200 * 1e3, 4),
                              "mlups": round(200 * n / dt / 1e6, 1)}), flush=True)
            del sim
        flow2 = lt.TaylorGreenVortex(ctx, res, 1600, 0.1, lt.D3Q19())
        s2 = lt.Simulation(flow2, lt.BGKCollision(0.53), [])
        s2(20); torch.cuda.synchronize(); t0 = time.perf_counter(); s2(200); dt = time.perf_counter() - t0
        print(json.dumps({"res": res, "plain": True, "ms_per_step": round(dt / 200 * 1e3, 4),
                          "mlups": round(200 * res[0] * res[1] * res[2] / dt / 1e6, 1)}), flush=True)
        del flow2, s2, flow
        torch.cuda.empty_cache()

main()
