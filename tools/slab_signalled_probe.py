"""lt_stream_collide_twice_slab alone (no exchange, nobody waiting for the counter) against
lt_stream_collide_twice_planes over the same planes, by segment length.  Dev tool."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


def main():
    ctx = lt.Context("cuda:0", torch.float32, True)
    res = [int(v) for v in sys.argv[1].split("x")] if len(sys.argv) > 1 else [512, 512, 64]
    slab = lt.ZSlab(res, 0, 1)
    flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab, initialize_fneq=False)
    sim = lt.TwoStepSlabSimulation(flow, lt.BGKCollision(0.53), slab)
    eng, lo, hi = sim.engine, sim.lo, sim.hi
    a, b = sim.f, sim.f_next
    out = {}
    for seg in ([int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 16, 32, res[2] // 2 - 1 if res[2] > 64 else 62)):
        eng.set_two_step(1, seg)
        out[f"planes seg{seg}"] = timed(lambda: eng.stream_collide_twice_planes(a, b, 0.53, lo, hi))
        out[f"slab launch seg{seg}"] = timed(lambda: eng.stream_collide_twice_slab(a, b, 0.53))
    out["res"] = res
    print(json.dumps(out), flush=True)


main()
