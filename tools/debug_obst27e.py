import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lettuce_amd as lt
from lettuce_amd._native import Plan, LAYOUT_SLAB
from conftest import golden
os.environ.setdefault("LT_SLAB_PAD", "0")
name = "obstacle3d_d3q27_bgk_64x8x16_f32"
g = golden(name)
ctx = lt.Context("cuda:0", torch.float32, use_native=True)
res = [int(r) for r in g["resolution"]]
slab = lt.ZSlab(res, 0, 1)
flow = lt.Obstacle(ctx, slab.extended_resolution, 100, 0.1, float(g["domain_length_x"]), stencil=lt.D3Q27(), slab=slab)
flow.mask = torch.tensor(g["obstacle_mask"])[:, :, slab.z_indices()]
flow.initialize()
sim = lt.TwoStepSlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
tau = float(sim._tau(sim.flow))
entries = [b.native_generator(i).plan_entry(sim.flow) for i, b in enumerate(sim.boundaries[1:], start=1)]
ncm, nsm = sim.no_collision_mask, sim.no_streaming_mask
eng, lo, hi = sim.engine, sim.lo, sim.hi
cur, nxt, _ = sim._start_batch(tau)
a = cur.clone()
n2 = a.shape[1]
t1, t2, t3 = torch.zeros_like(a), torch.zeros_like(a), torch.zeros_like(a)
eng.stream_collide_planes(a, t1, tau, lo - 1, hi + 1)
eng.stream_collide_planes(t1, t2, tau, lo, hi)
eng.set_two_step(1, 4)
eng.stream_collide_twice_planes(a, t3, tau, lo, hi)
torch.cuda.synchronize()
print("baseline mismatches", int((t2[:, lo:hi] != t3[:, lo:hi]).sum()))
z, y, x = lo, 2, 1
print("want", t2[:6, z, y, x].tolist()); print("got ", t3[:6, z, y, x].tolist())
inlet = [i for i, e in enumerate(entries, 1) if e["kind"] == "equilibrium"][0]
print("inlet slot", inlet, "feq[:6]", [round(v, 8) for v in entries[inlet - 1]["feq"][:6]])


def second_step_from(t1mod, label):
    out = torch.zeros_like(a)
    eng.stream_collide_planes(t1mod, out, tau, lo, hi)
    torch.cuda.synchronize()
    same = bool(torch.equal(out[:, z, y, x], t3[:, z, y, x]))
    print(f"candidate {label}: reproduces the two-step value at (z,y,x)=({z},{y},{x}): {same}; q0 = {float(out[0, z, y, x])}")
    if same:
        print("   whole first plane equal:", bool(torch.equal(out[:, z], t3[:, z])))

# A: the inlet nodes of intermediate plane z-1 are zero (feq of the wrong cache slot)
for rows, lab in ((slice(None), "all rows"), (slice(3, 5), "rows 3-4"), ([3, 7], "rows 3,7"), ([0, 3, 4, 7], "rows 0,3,4,7")):
    m = t1.clone(); m[:, z - 1, rows, 0] = 0.0
    second_step_from(m, f"A zero inlet in plane z-1, {lab}")
# B: the inlet nodes of plane z-1 hold the pulled + collided populations (no feq overwrite)
plan_noinlet = Plan("D3Q27", torch.float32, "bgk", slab.local_resolution, entries, layout=LAYOUT_SLAB, ghost_planes=2)
m2 = ncm.clone(); m2[m2 == inlet] = 0
plan_noinlet.set_masks(m2, nsm)
tB = torch.zeros_like(a)
plan_noinlet.stream_collide_planes(a, tB, tau, lo - 1, hi + 1)
for rows, lab in ((slice(None), "all rows"), ([3, 7], "rows 3,7"), ([0, 3, 4, 7], "rows 0,3,4,7")):
    m = t1.clone(); m[:, z - 1, rows, 0] = tB[:, z - 1, rows, 0]
    second_step_from(m, f"B collided inlet in plane z-1, {lab}")
# C: the inlet nodes of plane z-1 hold the pulled, uncollided populations
plan_s = Plan("D3Q27", torch.float32, "none", slab.local_resolution, entries, layout=LAYOUT_SLAB, ghost_planes=2)
plan_s.set_masks(m2, nsm)
tC = torch.zeros_like(a)
plan_s.stream_collide_planes(a, tC, tau, lo - 1, hi + 1)
for rows, lab in ((slice(None), "all rows"), ([3, 7], "rows 3,7")):
    m = t1.clone(); m[:, z - 1, rows, 0] = tC[:, z - 1, rows, 0]
    second_step_from(m, f"C streamed-only inlet in plane z-1, {lab}")
# D: the same three but in ALL planes (if the first-plane reasoning is off)
m = t1.clone(); m[:, :, :, 0] = 0.0
second_step_from(m, "D zero inlet everywhere")
