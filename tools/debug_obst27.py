import os, sys, numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(rank, world, port, name, lattice):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import lettuce_amd as lt
    from conftest import golden
    g = golden(name)
    ctx = lt.Context("cuda:0", torch.float32, use_native=True)
    res = [int(r) for r in g["resolution"]]
    for n in (1, 2, 3):
        out = {}
        for driver in ("SlabSimulation", "TwoStepSlabSimulation"):
            slab = lt.ZSlab(res)
            flow = lt.Obstacle(ctx, slab.extended_resolution, 100, 0.1, float(g["domain_length_x"]), stencil=getattr(lt, lattice)(), slab=slab)
            flow.mask = torch.tensor(g["obstacle_mask"])[:, :, slab.z_indices()]
            flow.initialize()
            sim = getattr(lt, driver)(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
            sim(n)
            f = sim.gather_f()
            if rank == 0:
                out[driver] = f.cpu().numpy()
                if driver == "TwoStepSlabSimulation": print("kernel", sim.engine.kernel_name(), "pop_stride", getattr(sim.engine, "pop_stride", None), flush=True)
        if rank == 0:
            a, b = out["TwoStepSlabSimulation"], out["SlabSimulation"]
            bad = np.argwhere(a != b)
            gd = g[f"f{n}"]
            print("n", n, "world", world, "mismatch", len(bad), "two-step vs golden", float(np.abs(a - gd).max()), "single vs golden", float(np.abs(b - gd).max()), flush=True)
            if len(bad):
                print(" q:", sorted(set(bad[:, 0]))[:30], "\n x:", sorted(set(bad[:, 1])), "\n y:", sorted(set(bad[:, 2])), "\n z:", sorted(set(bad[:, 3])), flush=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    for world, name, lat in ((4, "obstacle3d_d3q27_bgk_64x8x16_f32", "D3Q27"), (2, "obstacle3d_d3q27_bgk_64x8x16_f32", "D3Q27"), (4, "obstacle3d_d3q19_bgk_64x8x16_f32", "D3Q19")):
        print("=== world", world, name, "PAD", os.environ.get("LT_SLAB_PAD"), flush=True)
        mp.spawn(worker, args=(world, 29700 + world + (os.getpid() % 100), name, lat), nprocs=world, join=True)
