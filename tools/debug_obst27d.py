import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lettuce_amd as lt
from lettuce_amd._native import Plan, LAYOUT_SLAB
from conftest import golden
os.environ.setdefault("LT_SLAB_PAD", "0")
name, lattice = "obstacle3d_d3q27_bgk_64x8x16_f32", "D3Q27"
g = golden(name)
ctx = lt.Context("cuda:0", torch.float32, use_native=True)
res = [int(r) for r in g["resolution"]]
slab = lt.ZSlab(res, 0, 1)
flow = lt.Obstacle(ctx, slab.extended_resolution, 100, 0.1, float(g["domain_length_x"]), stencil=lt.D3Q27(), slab=slab)
flow.mask = torch.tensor(g["obstacle_mask"])[:, :, slab.z_indices()]
flow.initialize()
sim = lt.TwoStepSlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
entries = [b.native_generator(i).plan_entry(sim.flow) for i, b in enumerate(sim.boundaries[1:], start=1)]
for e in entries:
    print({k: (v if not isinstance(v, (list, torch.Tensor)) else "...") for k, v in e.items()})
ncm, nsm = sim.no_collision_mask, sim.no_streaming_mask
for variant in ("all", "no-obstacle", "inlet-only"):
    ents, m, s = entries, ncm.clone(), nsm.clone()
    if variant != "all":
        bb = [i for i, e in enumerate(entries, 1) if e["kind"] == "bounce_back"][0]
        m[m == bb] = 0
    plan = Plan("D3Q27", torch.float32, "none", slab.local_resolution, ents, layout=LAYOUT_SLAB, ghost_planes=2)
    plan.set_masks(m, s)
    q, n2, n1, n0 = plan.f_shape
    idx = torch.arange(q * n2 * n1 * n0, device="cuda", dtype=torch.float32).reshape(plan.f_shape)   # exact: < 2^24
    f = idx.clone()
    t1, t2, t3 = torch.zeros_like(f), torch.zeros_like(f), torch.zeros_like(f)
    plan.stream_collide_planes(f, t1, 0.6, 1, n2 - 1)
    plan.stream_collide_planes(t1, t2, 0.6, 2, n2 - 2)
    plan.set_two_step(1, 4)
    print(variant, "admitted:", plan.two_step_admitted())
    plan.stream_collide_twice_planes(f, t3, 0.6, 2, n2 - 2)
    torch.cuda.synchronize()
    bad = torch.nonzero(t2[:, 2:n2 - 2] != t3[:, 2:n2 - 2]).cpu().numpy()
    print(variant, "mismatches", len(bad))

    def dec(v):
        v = int(v); x = v % n0; v //= n0; y = v % n1; v //= n1; z = v % n2; v //= n2
        return (v, z, y, x)
    for b in bad[:12]:
        qq, z, y, x = b
        want, got = float(t2[qq, 2 + z, y, x]), float(t3[qq, 2 + z, y, x])
        print("  at q,z,y,x", (int(qq), int(2 + z), int(y), int(x)), "want", want, dec(want) if want == int(want) and want >= 0 else "", "got", got, dec(got) if got == int(got) and got >= 0 else "")
