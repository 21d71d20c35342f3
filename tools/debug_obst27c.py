import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OLD = os.environ.get("LT_OLD_TREE")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, OLD if OLD else ROOT); sys.path.append(ROOT)
import lettuce_amd as lt
from conftest import golden
os.environ.setdefault("LT_SLAB_PAD", "0")
for name, lattice in (("obstacle3d_d3q27_bgk_64x8x16_f32", "D3Q27"), ("obstacle3d_d3q19_bgk_64x8x16_f32", "D3Q19")):
    g = golden(name)
    ctx = lt.Context("cuda:0", torch.float32, use_native=True)
    res = [int(r) for r in g["resolution"]]
    slab = lt.ZSlab(res, 0, 1)
    flow = lt.Obstacle(ctx, slab.extended_resolution, 100, 0.1, float(g["domain_length_x"]), stencil=getattr(lt, lattice)(), slab=slab)
    flow.mask = torch.tensor(g["obstacle_mask"])[:, :, slab.z_indices()]
    flow.initialize()
    sim = lt.TwoStepSlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
    tau = float(sim._tau(sim.flow))
    eng, lo, hi = sim.engine, sim.lo, sim.hi
    print(name, "entries", [(type(b).__name__) for b in sim.boundaries[1:]], "kernel", eng.kernel_name(), "lo hi", lo, hi, "f shape", list(sim._f.shape), flush=True)
    cur, nxt, _ = sim._start_batch(tau)
    a = cur.clone()
    t1, t2, t3 = torch.zeros_like(a), torch.zeros_like(a), torch.zeros_like(a)
    eng.stream_collide_planes(a, t1, tau, lo - 1, hi + 1)
    eng.stream_collide_planes(t1, t2, tau, lo, hi)
    for seg in (0, 2, 4, 16):
        eng.set_two_step(1, seg)
        t3.zero_()
        eng.stream_collide_twice_planes(a, t3, tau, lo, hi)
        torch.cuda.synchronize()
        d = (t2[:, lo:hi] != t3[:, lo:hi])
        bad = torch.nonzero(d).cpu().numpy()
        print(" seg", seg, "mismatches", len(bad), flush=True)
        if len(bad):
            print("  q", sorted(set(bad[:, 0]))[:30], "z", sorted(set(bad[:, 1])), "y", sorted(set(bad[:, 2])), "x", sorted(set(bad[:, 3])))
            q, z, y, x = bad[0]
            print("  first", bad[0], "one-step", float(t2[q, lo + z, y, x]), "two-step", float(t3[q, lo + z, y, x]))
            ncm = sim.no_collision_mask
            print("  ncm at x=0..3 (y=0, z=lo):", ncm[lo, 0, :4].tolist(), "ncm at x=60..63:", ncm[lo, 0, 60:].tolist())
            print("  t2 q=0..3 at x=0:", t2[:4, lo, 0, 0].tolist(), " t3:", t3[:4, lo, 0, 0].tolist())
            print("  t2 q=0..3 at x=1:", t2[:4, lo, 0, 1].tolist(), " t3:", t3[:4, lo, 0, 1].tolist())
