"""Kinetic-energy decay of the 2-D Taylor-Green vortex -- the configuration of the reference's
`examples/00_simplest_TGV.py` (D2Q9, 128^2, Re 100, Ma 0.05, BGK, 1000 steps), written against
`lettuce_amd`.  On an MI355X the default context runs the HIP engine; `--cpu` uses the torch path.

    python examples/tgv2d_energy_decay.py [--cpu] [--steps 1000]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lettuce_amd as lt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--steps", type=int, default=1000)
    args = ap.parse_args()
    context = (lt.Context("cpu", torch.float64, use_native=False) if args.cpu
               else lt.Context(dtype=torch.float64))
    flow = lt.TaylorGreenVortex(context, resolution=128, reynolds_number=100, mach_number=0.05,
                                stencil=lt.D2Q9)
    energy = lt.ObservableReporter(lt.IncompressibleKineticEnergy(flow), interval=100, out=None)
    simulation = lt.Simulation(flow, lt.BGKCollision(tau=flow.units.relaxation_parameter_lu), [energy])
    mlups = simulation(args.steps)
    for step, time_pu, value in energy.out:
        print(f"step {step:5d}  t = {time_pu:8.4f}  E = {value:.12f}")
    print(f"{mlups:.1f} MLUPS on {context.device} "
          f"({'HIP engine' if context.use_native else 'torch ops'})")


if __name__ == "__main__":
    main()
