"""fp64 one-step kernels of the D3Q19 unit (cfg5's shape with the two-step kernel switched off, and the Obstacle with
its masked kernel): ms per update.  Used with LT_ENGINE_LIBRARY to check what a compiler setting tried on the two-step
kernel does to the other kernels of the unit (round 3: max-ilp costs the masked kernel 8 %, hence its own object)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, lettuce_amd as lt
def timed(sim, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sim._native.fused_events = (e0, e1)
    sim(steps); torch.cuda.synchronize()
    info = sim._native.plan.last_run_info(); sim._native.fused_events = None
    launches = info["two_step_launches"] or info["single_step_launches"]
    return e0.elapsed_time(e1) / launches / (2 if info["two_step_launches"] else 1)
ctx = lt.Context("cuda:0", torch.float64, True)
flow = lt.DoublyPeriodicShear3D(ctx, [384, 384, 96], 10000, 0.1)
sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
sim._native.batch(1); sim._native.plan.set_two_step(0)
sim(10)
print("cfg5 one-step fp64", [round(timed(sim, 40), 4) for _ in range(3)], sim._native.plan.kernel_name())
del sim, flow; torch.cuda.empty_cache()
flow = lt.Obstacle(ctx, [256, 256, 256], 100, 0.1, domain_length_x=4, stencil=lt.D3Q19())
x, y, z = flow.grid
flow.mask = ((x - 1) ** 2 + (y - 2) ** 2 + (z - 2) ** 2) < 0.5 ** 2
flow.initialize()
sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
sim(10)
print("obstacle D3Q19 fp64 (auto)", [round(timed(sim, 40), 4) for _ in range(3)], sim._native.plan.kernel_name())
