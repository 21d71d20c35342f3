"""One rank of the CPU rehearsal of bench.py's N > 1 candidate loop (tests/test_bench_failsoft.py): gloo ranks, the
z-slab drivers on the oracle-backed CPU engine, and stub candidates that misbehave the way a transport might on real
links -- raise while being built, raise in a timed batch, disagree with the reference, take longer than their wall
budget, or never return.  The line rank 0 prints must be the held one whatever they do."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench                                   # noqa: E402
import lettuce_amd as lt                       # noqa: E402
from slab_cpu_engine import OracleSlabEngine   # noqa: E402


class Misbehaving:
    """a slab driver that goes wrong in the way its name says, after `good` well-behaved calls"""

    def __init__(self, sim, how, good):
        self.sim, self.how, self.calls, self.good = sim, how, 0, good
        self.engine = sim.engine

    def __call__(self, steps):
        self.calls += 1
        if self.calls > self.good:
            if self.how == "raises-in-batch":
                raise RuntimeError("stub: transfer failed")
            if self.how == "stalls":
                time.sleep(1.5)                  # every call: the budget (2 s) runs out between batches
            if self.how == "never-returns":
                time.sleep(10000)
        return self.sim(steps)

    def local_f(self):
        f = self.sim.local_f()
        if self.how == "disagrees":
            f = f + 1e-9
        return f


def main():
    stubs = sys.argv[1].split(",")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    args = bench.parse(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--batches", "2",
                        "--candidate-budget", "2", "--watchdog-grace", "2", "--first-candidate-budget", "20"])
    if "first-never-returns" in stubs:
        args.first_candidate_budget = 2.0
    bench.MIN_BATCH_S = 0.0
    ctx = lt.Context("cpu", torch.float64, use_native=False)
    res = [8, 4, 8 * world]
    slab = lt.ZSlab(res)

    def real(driver):
        flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 100, 0.1, lt.D3Q19(), slab=slab)
        coll = lt.BGKCollision(flow.units.relaxation_parameter_lu)
        cls = lt.TwoStepSlabSimulation if driver == "two-step" else lt.SlabSimulation
        return cls(flow, coll, slab, engine=OracleSlabEngine("D3Q19", torch.float64, "bgk"))

    def build(driver, transport):
        if transport == "raises-in-build":
            raise RuntimeError("stub: no such transport on this node")
        sim = real(driver)
        if transport in ("raises-in-batch", "stalls", "never-returns"):
            # good calls: set-up (1) + probe (2) + warm-up (1) + sizing (1), then it goes wrong in the timed batches
            return Misbehaving(sim, transport, 5)
        if transport == "disagrees":
            return Misbehaving(sim, transport, 10 ** 9)
        return sim

    if "first-never-returns" in stubs:          # the reference candidate itself hangs in its first timed batch
        wanted = [("single-step", "never-returns"), ("two-step", "rccl")]
    else:
        wanted = [("single-step", "rccl")] + [("two-step", t) for t in stubs]
    ranks = bench.Ranks(dist, world, rank, 0, torch.device("cpu"))
    cpu_row = {"value": 1.0, "unit": "MLUPS", "cores": 1, "kind": "port", "sample": "stub"} if rank == 0 else None
    bench.candidate_loop(args, ranks, wanted, build, "stub workload", res, res[0] * res[1] * slab.nz_local, 304, "f64",
                         probe_steps=2, cpu_baseline_row=cpu_row, traffic_workload="no such workload")
    dist.barrier()
    dist.destroy_process_group()


main()
