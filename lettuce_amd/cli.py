"""`python -m lettuce_amd benchmark|convergence` -- the two integration commands of the reference's
command line ("next" row F4; lettuce/cli.py:57-125 benchmark, :128-180 convergence) on this engine.
Plain argparse instead of click; same defaults where they exist."""
import argparse
import sys

import numpy as np
import torch

from . import (BGKCollision, Context, ErrorReporter, Simulation, TaylorGreenVortex, flow_by_name)

__all__ = ["main"]


def _context(args):
    device = "cuda:0" if args.cuda and torch.cuda.is_available() else "cpu"
    dtype = {"half": torch.float16, "single": torch.float32, "double": torch.float64}[args.precision]
    return Context(device=device, dtype=dtype, use_native=args.use_native and device != "cpu")


def benchmark(args):
    """MLUPS of `steps` steps of a periodic flow (no reporters)."""
    ctx = _context(args)
    flow_class, stencil = flow_by_name[args.flow]
    flow = flow_class(ctx, args.resolution, 1, 0.05, stencil)
    sim = Simulation(flow, BGKCollision(tau=flow.units.relaxation_parameter_lu), [])
    sim(min(10, args.steps))                                  # warm-up
    mlups = sim(args.steps)
    print(f"Finished {args.steps} steps of {args.flow} at {flow.resolution} on {ctx.device} "
          f"({'HIP engine' if sim._native else 'torch ops'}): {mlups:.1f} MLUPS")
    return 0


def convergence(args):
    """Taylor-Green 2-D in diffusive scaling: order 2 in u, order 1 in p."""
    ctx = _context(args)
    print(("{:>15} " * 6).format("resolution", "error (u)", "order (u)", "error (p)", "order (p)", "MLUPS"))
    old_u = old_p = None
    factor_u = factor_p = 0.0
    for i in range(4, 9):
        res = 2 ** i
        flow = TaylorGreenVortex(ctx, [res] * 2, reynolds_number=10000, mach_number=8 / res)
        reporter = ErrorReporter(flow.analytic_solution, interval=1, out=None)
        sim = Simulation(flow, BGKCollision(tau=flow.units.relaxation_parameter_lu), [reporter])
        mlups = sim(10 * res)
        err_u, err_p = np.mean(np.abs(reporter.out), axis=0).tolist()
        factor_u = 0 if old_u is None else old_u / err_u
        factor_p = 0 if old_p is None else old_p / err_p
        old_u, old_p = err_u, err_p
        print(f"{res:15} {err_u:15.2e} {factor_u / 2:15.2f} {err_p:15.2e} {factor_p / 2:15.2f} {mlups:15.2f}")
    tol = 1e-1
    if not (2 - tol) < factor_u / 2 < (2 + tol):
        print(f"FAILED: Velocity convergence order {factor_u / 2} is not in [1.9, 2.1]")
        return 1
    if not (1 - tol) < factor_p / 2 < (1 + tol):
        print(f"FAILED: Pressure convergence order {factor_p / 2} is not in [0.9, 1.1].")
        return 1
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="lettuce_amd")
    ap.add_argument("--cuda", dest="cuda", action="store_true", default=True)
    ap.add_argument("--no-cuda", dest="cuda", action="store_false")
    ap.add_argument("-p", "--precision", choices=["half", "single", "double"], default="double")   # lettuce/cli.py:33-37
    ap.add_argument("--use-native", dest="use_native", action="store_true", default=True)
    ap.add_argument("--use-no-native", dest="use_native", action="store_false")
    sub = ap.add_subparsers(dest="command", required=True)
    b = sub.add_parser("benchmark")
    b.add_argument("-s", "--steps", type=int, default=10)
    b.add_argument("-r", "--resolution", type=int, default=1024)
    b.add_argument("-f", "--flow", default="taylor2D", choices=sorted(flow_by_name))
    sub.add_parser("convergence")
    args = ap.parse_args(argv)
    return benchmark(args) if args.command == "benchmark" else convergence(args)


if __name__ == "__main__":
    sys.exit(main())
