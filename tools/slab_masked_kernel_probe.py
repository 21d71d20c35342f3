"""lbm2m_kernel AX = 0 alone (no exchange): the two-step launch over the interior planes of a one-rank slab with
the Obstacle's boundaries, by segment length; beside it the one-step masked slab kernel.  Dev tool."""
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
    return e0.elapsed_time(e1) / n


def main():
    ctx = lt.Context("cuda:0", torch.float32, True)
    variant = sys.argv[1] if len(sys.argv) > 1 else "obstacle"
    for lattice, res in (("D3Q19", [512, 512, 64]),):
        slab = lt.ZSlab(res, 0, 1)
        class Variant(lt.Obstacle):
            @property
            def boundaries(self):
                full = lt.Obstacle.boundaries.fget(self)
                names = {"obstacle": ("EquilibriumBoundaryPU", "AntiBounceBackOutlet", "BounceBackBoundary"),
                         "sphere": ("BounceBackBoundary",), "inlet": ("EquilibriumBoundaryPU", "BounceBackBoundary"),
                         "outlet": ("AntiBounceBackOutlet", "BounceBackBoundary")}[variant]
                return [b for b in full if type(b).__name__ in names]
        flow = Variant(ctx, slab.extended_resolution, 100, 0.1, domain_length_x=4, stencil=getattr(lt, lattice)(), slab=slab)
        x, y, z = flow.grid
        flow.mask = ((x - 1) ** 2 + (y - 2) ** 2 + (z - 0.25) ** 2) < 0.4 ** 2
        flow.initialize()
        try:
            sim = lt.TwoStepSlabSimulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), slab)
        except lt.LettuceException as e:
            print(json.dumps({"boundaries": variant, "refused": str(e)[:200]}), flush=True)
            continue
        eng, lo, hi = sim.engine, sim.lo, sim.hi
        a, b = sim.f, sim.f_next
        tau = 0.6
        out = {"boundaries": variant, "lattice": lattice, "res": res, "kernel": eng.kernel_name(), "planes": hi - lo}
        out["one_step_ms"] = round(timed(lambda: eng.stream_collide_planes(a, b, tau, lo, hi)), 4)
        for seg in (0, 8, 16, 32, 64):
            eng.set_two_step(1, seg)
            out[f"two_step_seg{seg}_ms_per_update"] = round(timed(lambda: eng.stream_collide_twice_planes(a, b, tau, lo, hi)) / 2, 4)
        print(json.dumps(out), flush=True)
        del sim, flow
        torch.cuda.empty_cache()


main()
