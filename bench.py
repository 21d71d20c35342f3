"""Headline benchmark: MLUPS of the LBM stream-and-collide hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json): TaylorGreenVortex3D, D3Q19, BGK, fp32, Re 1600, Ma 0.1, synthetic
(analytic) initial condition incl. the f_neq initialisation.
  N = 1   256^3 on one GPU (configs[1]) through lt.Simulation and the fused HIP kernel.
  N > 1   weak scaling at 256^3 = 16.8 M nodes per GPU, z-slab decomposition with RCCL
          ghost-plane exchange: global 512 x 512 x 64N, i.e. every rank holds the 512 x 512 x 64
          slab of BASELINE's configs[2] (N = 8 is exactly its 512^3).
  --workload cfg5   BASELINE's configs[4] through the same slab path: DoublyPeriodicShear3D D3Q19 fp64,
          global 384 x 384 x 96N (N = 4 is its 384^3); with N = 1 add --slab.
A "step" is one full lattice update (collide, stream) of every node.  The timed region is
exactly K steps between barrier + device synchronise, max over ranks; value = all nodes of all
ranks * K / time.  One JSON line is printed by rank 0.

Launching.  ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts its
own N ranks (a ``torch.distributed.run`` child process, before this process touches the GPU),
relays rank 0's JSON line and the exit code; fewer than N visible devices is an error, never a
silent N = 1 run.  Under a launcher (WORLD_SIZE set) the process is one rank.

Timing.  Five timed batches (barrier + device synchronise on both sides, max over ranks);
``ms_per_step`` / ``value`` are those of the median batch, all five are listed in
``batches_ms_per_step``.  A batch is the K-step call repeated R times back to back, R chosen so that
a batch lasts at least ~50 ms (R = 1 when K steps already do): with the driver's K = 20 a single call
is 5 ms and one 20 % outlier in five moved the median; ``timing`` states R, ``steps`` stays K and
every number in the line is per K steps.

N > 1 fails soft.  The slab candidates (driver x transport) are tried one after the other; the first
one -- the single-step driver over RCCL send/recv -- is timed for its K steps as soon as its warm-up
probe has finished, and that line is HELD.  A later candidate replaces it only if it is bit-identical to
it after the warm-up probe AND after its own timed batches, and faster; if one raises, is rejected, or
exceeds its wall budget (checked on the host between batches; a watchdog thread covers a call that never
returns), rank 0 still prints the held line with the failure recorded in ``config.transport`` and all ranks
exit 0.  Nothing is re-executed and no process that has touched the GPU is restarted.  The first candidate has a
budget too (--first-candidate-budget, 600 s): if IT never returns, every rank ends itself (exit code 3) and rank 0
prints a line with value 0 and the reason rather than hanging until the caller's limit.

roofline: HBM-bound kernel.  ``achieved`` = ALGORITHMIC bytes per launch of the dominant kernel --
SURVEY.md 8(d)'s 2 * 19 * 4 = 152 B per lattice update x the lattice updates one launch performs (nodes x
``lattice_updates_per_node_per_launch``) -- divided by the launch's average duration, measured with HIP
events recorded on the launch stream around the fused launches of the median batch; ``frac`` =
achieved / 8 TB/s.  The two-step kernel performs two lattice updates per node and launch with the
intermediate state in LDS, so HBM carries every population once in and once out per TWO updates and
``frac`` can exceed 1; what HBM physically has to carry per second (populations read once + written
once per launch / launch time, <= the peak) is ``physical_GBps`` / ``physical_frac``.  ``traffic`` = HBM
bytes per launch from the rocprofv3 PMC passes in profiles/traffic.json, used only if that file was
produced from the kernel sources of this build (hash of lettuce_amd/csrc), else null.
verified: after the timed batches the same steps are repeated, untimed, from a copy of the
pre-timed populations with the one-step kernel only (``set_two_step(0)``); ``verified`` says
whether the two final states are bit-identical; final mass and kinetic energy are in the line.
other_configs (N = 1): BASELINE's configs[3] (Obstacle3D D3Q27 256^3 KBC fp32) and the per-GPU shape of
configs[4] (shear layer D3Q19 384 x 384 x 96 fp64) through lt.Simulation: kernel, ms per update, physical and
algorithmic GB/s and a bit-identity flag each (about 15 s; --no-other-configs skips them).
cpu_baseline: the CPU oracle (a torch-CPU restatement of the reference's path, validated
against the reference) timed on this box's host cores on the same 256^3 workload for a few
steps (rank 0, N = 1 only), at all host threads and at 32 threads; the better one is ``value``.
"""
import argparse
import json
import math
import os
import sys
import threading
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")     # RCCL kernels beside the interior kernel
# stdout carries exactly one JSON line: RCCL's NCCL_DEBUG=VERSION banner (exported on the GPU boxes)
# goes to stdout, so keep RCCL to warnings and send those to stderr
os.environ["NCCL_DEBUG"] = "WARN"
os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s copy ceiling
MIN_BATCH_S = 0.05               # a timed batch lasts at least this long (the K-step call is repeated)

WORKLOADS = {
    # name: (flow, stencil, dtype, bytes per node and update, per-GPU block when --size 256)
    "cfg3": ("TaylorGreenVortex3D", "D3Q19", "float32", 2 * 19 * 4, (512, 512, 64)),
    "cfg5": ("DoublyPeriodicShear3D", "D3Q19", "float64", 2 * 19 * 8, (384, 384, 96)),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batches", type=int, default=5,
                    help="timed batches; the median one is reported")
    ap.add_argument("--size", type=int, default=256, help="nodes per side of the per-GPU block")
    ap.add_argument("--workload", choices=["cfg3", "cfg5"], default="cfg3",
                    help="slab path: cfg3 = TGV D3Q19 fp32, 512 x 512 x 64 per GPU (BASELINE configs[2] at N = 8); "
                         "cfg5 = shear layer D3Q19 fp64, 384 x 384 x 96 per GPU (configs[4] at N = 4)")
    ap.add_argument("--cpu-baseline-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1: skip the cfg4 / cfg5-shape rows of the line (profiling passes)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the untimed re-run with the one-step kernel (profiling passes)")
    ap.add_argument("--no-overlap", action="store_true", help="slab path: exchange without overlap")
    ap.add_argument("--transport", choices=["auto", "all", "rccl", "window", "copy"],
                    default=os.environ.get("LT_BENCH_TRANSPORT", "auto"),
                    help="slab path: ghost-plane transport (lettuce_amd/_slab.py).  auto = RCCL send/recv (both slab "
                         "drivers) and, for the two-step driver, the copy-engine transport (no RCCL kernel beside the sweep; "
                         "a lost message is a time-out the driver raises, not a trap); results must be bit-identical; "
                         "all = also the launch that signals the "
                         "exchange from inside and the one-sided peer-window transports, which are faster in "
                         "the one-GPU rehearsal but have never run across real xGMI links and end in a device "
                         "trap if a signal is lost -- opt in with --transport all or LT_BENCH_TRANSPORT=all")
    ap.add_argument("--driver", choices=["auto", "single-step", "two-step"], default="auto",
                    help="slab path: restrict the candidates to one slab driver")
    ap.add_argument("--candidate-budget", type=float, default=float(os.environ.get("LT_BENCH_CANDIDATE_BUDGET", "120")),
                    help="slab path: wall seconds a candidate after the first may take (checked between batches; "
                         "a watchdog prints the held line and ends the rank 60 s later if a call never returns)")
    ap.add_argument("--first-candidate-budget", type=float,
                    default=float(os.environ.get("LT_BENCH_FIRST_BUDGET", "600")),
                    help="N > 1: wall seconds the first (reference) candidate may take; past it (+ the grace) every rank "
                         "ends itself and rank 0 prints a line with value 0 and the reason, instead of hanging until "
                         "somebody else's limit")
    ap.add_argument("--watchdog-grace", type=float, default=60.0,
                    help="slab path: seconds past a candidate's budget after which the watchdog prints the held line "
                         "and ends the rank")
    ap.add_argument("--hang-exit-code", type=int, default=int(os.environ.get("LT_BENCH_HANG_EXIT_CODE", "0")),
                    help="slab path: exit code of a rank whose candidate call never returned, after the held line of an "
                         "earlier candidate has been printed (default 0: the line is a complete measurement; the hang is "
                         "recorded in config.transport.aborted and on stderr)")
    ap.add_argument("--slab", action="store_true",
                    help="use the z-slab driver (and an RCCL process group) even with one GPU: "
                         "rehearsal of the N > 1 code path")
    return ap.parse_args(argv)


# ---- launching N ranks ------------------------------------------------------------------------
def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, worker=None, visible=None, timeout=None):
    """Run ``worker`` (default: this file) as n ranks of one node through ``torch.distributed.run``
    in a CHILD process and return its exit code; the child's stdout (rank 0's JSON line) goes to
    this process's stdout.  Must be called before anything in this process has initialised the
    GPU: ``torch.cuda.device_count()`` does not.  ``visible`` overrides the device count (tests)."""
    import subprocess
    if visible is None:
        visible = torch.cuda.device_count()
    if visible < n:
        sys.stderr.write(f"bench.py: --gpus {n} requested but only {visible} GPU(s) are visible on this "
                         f"node; refusing to run a smaller job under the name of n_gpus = {n}\n")
        return 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           worker or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    proc = subprocess.run(cmd, env=env, timeout=timeout)
    if proc.returncode != 0:
        sys.stderr.write(f"bench.py: the {n}-rank job exited with code {proc.returncode}\n")
    return proc.returncode


def cpu_baseline(size, steps):
    """The oracle on the host cores, same workload, bounded sample: at every host thread torch
    uses by default and at 32 threads (the torch-CPU path stops scaling long before 128)."""
    from oracle import lettuce_oracle as orc
    default = torch.get_num_threads()
    runs = []
    for threads in dict.fromkeys([default, min(default, 32)]):
        torch.set_num_threads(threads)
        sim = orc.taylor_green([size] * 3, 1600, 0.1, "D3Q19", torch.float32)
        sim.step(1)                                   # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        sim.step(steps)
        dt = time.perf_counter() - t0
        runs.append((round(steps * size ** 3 / 1e6 / dt, 3), threads))
        del sim
    torch.set_num_threads(default)
    best = max(runs)
    return {"value": best[0], "unit": "MLUPS", "cores": best[1], "kind": "port",
            "sample": f"oracle/lettuce_oracle.py (torch CPU ops), TGV3D D3Q19 BGK fp32 {size}^3, "
                      f"{steps} steps after 1 warm-up step per thread count; "
                      + ", ".join(f"{v} MLUPS at {t} threads" for v, t in runs)
                      + f"; host has {os.cpu_count()} logical CPUs"}


def copy_ceiling(device, nbytes=1 << 30, iters=20):
    """This device's plain streaming-copy rate (16 B per lane, read + write bytes counted), the
    practical HBM ceiling next to the 8 TB/s spec number."""
    from lettuce_amd._native import probe_copy
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    best = 0.0
    for policy in (0, 3):                      # cached, nontemporal
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        probe_copy(dst, src, policy)
        e0.record()
        for _ in range(iters):
            probe_copy(dst, src, policy)
        e1.record()
        torch.cuda.synchronize(device)
        best = max(best, 2 * nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def source_hash():
    """Hash of the kernel sources the in-tree library is built from (lettuce_amd/csrc)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "lettuce_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hpp", ".hip", ".inc")):
            with open(os.path.join(src, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def traffic_from_profile(kernel_name, workload="tgv3d_d3q19_bgk_f32_256"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes
    (profiles/traffic.json, produced by tools/pmc_traffic.py on the GPU box) -- only if that file
    was measured on a build of the same kernel sources (its ``source_hash``); else null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            table = json.load(fh)
        if table.get("source_hash") != source_hash():
            return None
        # the engine's name and rocprofv3's differ in case, in the "void lt::" prefix and in what
        # follows the argument list
        def core(name):
            name = name.lower().replace(" ", "")
            name = name[name.index("lbm"):] if "lbm" in name else name
            return name.split(">(")[0].rstrip(">")
        want = core(kernel_name)
        for row in table.get("kernels", []):
            have = core(row.get("kernel", ""))
            if (row.get("workload") == workload and row.get("hbm_bytes_per_launch")
                    and have.split("<")[0] == want.split("<")[0]
                    and (want.startswith(have) or have.startswith(want))):
                return row["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        pass
    return None


def ensure_library(local_rank):
    """The in-tree engine library normally travels with the tree; on a fresh tree local rank 0
    builds it (hipcc, ~1 min) while the other ranks wait for the file."""
    from lettuce_amd import _native
    path = _native.library_path()
    if os.path.exists(path):
        return
    if local_rank == 0:
        import __graft_entry__
        __graft_entry__.build()
        return
    deadline = time.time() + 900
    while not os.path.exists(path):
        if time.time() > deadline:
            raise SystemExit(f"{path} was not built")
        time.sleep(2)
    time.sleep(2)          # let the linker finish writing


def median_index(values):
    order = sorted(range(len(values)), key=lambda i: values[i])
    return order[len(order) // 2]


def repeats_for(seconds_per_call):
    """how often the K-step call is repeated inside one timed batch (>= MIN_BATCH_S per batch)"""
    if seconds_per_call <= 0:
        return 1
    return max(1, min(1000, int(math.ceil(MIN_BATCH_S / seconds_per_call))))


# ---- N = 1: the other BASELINE configurations -------------------------------------------------
def other_configs(lt, device):
    """cfg4 (Obstacle3D D3Q27 256^3 KBC fp32: inlet + anti-bounce-back outlet + sphere) and the per-GPU shape of
    cfg5 (shear layer D3Q19 384 x 384 x 96 fp64) through lt.Simulation: the kernel lt_run's fused steps use, its
    time per lattice update (HIP events around the fused launches), physical and algorithmic GB/s, and one
    bit-identity flag each."""
    rows = []

    def timed(sim, warm, steps):
        sim(warm)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sim._native.fused_events = (e0, e1)
        t0 = time.perf_counter()
        sim(steps)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        sim._native.fused_events = None
        info = sim._native.plan.last_run_info()
        per_launch = 2 if info["two_step_launches"] else 1
        launches = info["two_step_launches"] or info["single_step_launches"]
        return dt, e0.elapsed_time(e1) / max(1, launches), per_launch

    def row(name, flow, sim, steps, dt, launch_ms, per_launch, bytes_per_node, workload_key):
        n = int(torch.tensor(flow.resolution).prod())
        kernel = sim._native.plan.kernel_name()
        physical = bytes_per_node * n / (launch_ms * 1e-3) / 1e9
        traffic = traffic_from_profile(kernel, workload_key)
        out = {"config": name, "resolution": list(flow.resolution), "kernel": kernel,
               "lattice_updates_per_launch": per_launch, "ms_per_update": round(launch_ms / per_launch, 5),
               "avg_launch_ms": round(launch_ms, 5), "MLUPS_wall": round(steps * n / dt / 1e6, 1),
               "bytes_per_node_and_update": bytes_per_node,
               "physical_GBps": round(physical, 1), "physical_frac_of_8TBs": round(physical / HBM_PEAK_GBS, 4),
               "algorithmic_GBps": round(physical * per_launch, 1),
               "algorithmic_frac_of_8TBs": round(physical * per_launch / HBM_PEAK_GBS, 4),
               "traffic": traffic}
        if traffic:
            out["hbm_traffic_GBps"] = round(traffic / (launch_ms * 1e-3) / 1e9, 1)
        return out

    # cfg4
    ctx = lt.Context(device=device, dtype=torch.float32, use_native=True)
    flow = lt.Obstacle(ctx, [256, 256, 256], 100, 0.1, domain_length_x=4, stencil=lt.D3Q27())
    x, y, z = flow.grid
    flow.mask = ((x - 1) ** 2 + (y - 2) ** 2 + (z - 2) ** 2) < 0.5 ** 2
    flow.initialize()
    sim = lt.Simulation(flow, lt.KBCCollision(), [])
    dt, launch_ms, per_launch = timed(sim, 10, 60)
    r = row("cfg4: Obstacle3D D3Q27 256^3 KBC fp32, inlet + anti-bounce-back outlet + sphere bounce-back", flow, sim,
            60, dt, launch_ms, per_launch, 2 * 27 * 4 + 1, "obstacle3d_d3q27_kbc_f32_256")
    # bit identity: 3 whole steps through lt_run (collide, fused, fused, stream) against collide -> stream three
    # times through the operator entry points, from the same populations
    plan = sim._native.plan
    start = flow.f.clone()
    tau = float(sim._native.collision.tau(flow))
    a, b = start.clone(), torch.empty_like(start)
    for _ in range(3):
        plan.collide(a, b, tau)
        plan.stream(b, a)
    flow.f = start
    sim(3)
    same = bool(torch.equal(flow.f, a))
    diff, scale = float((flow.f - a).abs().max()), float(a.abs().max())
    r["check"] = {"what": "3 steps of lt.Simulation (fused stream-collide launches) against collide, stream through "
                          "the operator entry points; KBC is compared at rounding level (hipcc contracts its "
                          "multiply-adds differently in the fused and the collide-only kernel; BGK flows are bit-identical)",
                  "bit_identical": same, "max_abs_diff": diff,
                  "agrees_at_fp32_rounding_level": diff <= 4e-6 * scale,
                  "finite": bool(torch.isfinite(flow.f).all())}
    rows.append(r)
    del sim, flow, plan, start, a, b, x, y, z
    torch.cuda.empty_cache()

    # cfg5's per-GPU shape
    ctx = lt.Context(device=device, dtype=torch.float64, use_native=True)
    flow = lt.DoublyPeriodicShear3D(ctx, [384, 384, 96], 10000, 0.1)
    sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
    dt, launch_ms, per_launch = timed(sim, 10, 60)
    r = row("cfg5 per-GPU shape: DoublyPeriodicShear3D D3Q19 384 x 384 x 96 BGK fp64", flow, sim, 60, dt, launch_ms,
            per_launch, 2 * 19 * 8, "shear3d_d3q19_bgk_f64_384x384x96")
    start = flow.f.clone()
    sim(5)
    two = flow.f.clone()
    info = sim._native.plan.last_run_info()
    flow.f = start
    sim._native.plan.set_two_step(0)
    sim(5)
    info1 = sim._native.plan.last_run_info()
    r["check"] = {"what": "5 steps with two updates per launch against the one-step kernel, torch.equal",
                  "bit_identical": bool(torch.equal(flow.f, two)), "two_step_launches": info["two_step_launches"],
                  "one_step_run_two_step_launches": info1["two_step_launches"]}
    rows.append(r)
    del sim, flow, start, two
    torch.cuda.empty_cache()

    from lettuce_amd._native import experiments_built
    if not experiments_built():
        return rows
    # cfg2 again with the BGK collision in fast arithmetic (opt-in: collision.arithmetic = "fast"; the headline line
    # above is the exact arithmetic, bit-identical to the reference's CPU path)
    ctx = lt.Context(device=device, dtype=torch.float32, use_native=True)
    flow = lt.TaylorGreenVortex(ctx, [256, 256, 256], 1600, 0.1, lt.D3Q19())
    start = flow.f.clone()
    coll = lt.BGKCollision(flow.units.relaxation_parameter_lu)
    coll.arithmetic = "fast"
    sim = lt.Simulation(flow, coll, [])
    dt, launch_ms, per_launch = timed(sim, 20, 100)
    r = row("cfg2 with the BGK collision in fast arithmetic (opt-in, rounding-level parity): TGV3D D3Q19 256^3 fp32", flow,
            sim, 100, dt, launch_ms, per_launch, 2 * 19 * 4, "tgv3d_d3q19_bgk_f32_256")
    flow.f = start.clone()
    sim(10)
    fast10 = flow.f.clone()
    exact = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
    flow.f = start
    exact(10)
    diff, scale = float((flow.f - fast10).abs().max()), float(flow.f.abs().max())
    r["check"] = {"what": "10 steps against the exact arithmetic from the same populations (SURVEY 8(d): max |df| <= 1e-5 max |f|)",
                  "bit_identical": bool(torch.equal(flow.f, fast10)), "max_abs_diff": diff, "max_abs_f": scale,
                  "within_1e-5_of_max_f": diff <= 1e-5 * scale}
    rows.append(r)
    del sim, exact, flow, start, fast10
    torch.cuda.empty_cache()
    return rows


# ---- the held line of the N > 1 path ----------------------------------------------------------
HANG_EXIT_CODE = 0


class HeldLine:
    """The line rank 0 will print, replaced only by better candidates; a watchdog prints it and ends the rank when
    a candidate call does not return."""

    def __init__(self, rank, nothing_held=None):
        self.rank, self.line, self.lock, self.printed = rank, None, threading.Lock(), False
        self.deadline, self.what = None, ""
        # nothing_held(why) -> the line to print when no candidate has produced one: value 0 and the reason
        self.nothing_held = nothing_held
        self._thread = threading.Thread(target=self._watch, daemon=True)
        self._thread.start()

    def arm(self, seconds, what):
        self.deadline, self.what, self.phase = time.time() + seconds, what, "set-up"

    def at(self, phase):
        """which call of the candidate is in flight (for the watchdog's diagnostic)"""
        self.phase = phase

    def disarm(self):
        self.deadline = None

    def emit(self):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if self.rank == 0 and self.line is not None:
                print(json.dumps(self.line), flush=True)

    def emit_nothing(self, why):
        """no candidate produced a line: rank 0 says so in the line's own format (value 0)"""
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if self.rank == 0 and self.nothing_held is not None:
                print(json.dumps(self.nothing_held(why)), flush=True)

    def _watch(self):
        while True:
            time.sleep(1.0)
            d = self.deadline
            if d is None or time.time() <= d:
                continue
            where = f"{self.what} ({getattr(self, 'phase', '?')})"
            sys.stderr.write(f"bench.py: rank {self.rank}: a call of candidate {where} did not return within its budget: "
                             f"a GPU or RCCL call is stuck; the rank ends itself\n")
            sys.stderr.flush()
            if self.line is not None:
                self.line["config"]["transport"]["aborted"] = f"{where}: no return within its budget; held line printed by the watchdog"
                self.emit()
                # the held line is a complete, verified measurement of an earlier candidate: by default the job still
                # counts as run (exit code 0, the hang is in config.transport.aborted and on stderr);
                # --hang-exit-code N makes the hang visible to a harness that only looks at exit codes
                os._exit(HANG_EXIT_CODE)
            # the reference candidate itself never came back: end the rank (every rank has the same deadline) rather
            # than hang until the caller's limit
            self.emit_nothing(f"{where}: no return within its budget and no line held; ended by the watchdog")
            os._exit(3)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one (nothing in this process has touched the GPU yet)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(or drop the launcher and let bench.py start the ranks itself)")
    ensure_library(local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"local rank {local_rank} has no GPU: {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1 or args.slab
    if args.slab and world == 1:
        # the rehearsal is only worth something if the halo messages really travel through RCCL (to the rank itself)
        os.environ.setdefault("LT_SLAB_FORCE_P2P", "1")
    if args.workload != "cfg3" and not distributed:
        raise SystemExit("--workload cfg5 is a slab workload: use --gpus N (N = 4 is BASELINE's configs[4]) or --slab")
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)

    import lettuce_amd as lt
    if distributed:
        return slab_bench(args, lt, dist, world, rank, local_rank, device)
    single_gpu_bench(args, lt, device)


# ---- N = 1 -----------------------------------------------------------------------------------
def single_gpu_bench(args, lt, device):
    bytes_per_node = WORKLOADS["cfg3"][3]
    ctx = lt.Context(device=device, dtype=torch.float32, use_native=True)
    n = args.size
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    flow = lt.TaylorGreenVortex(ctx, [n, n, n], 1600, 0.1, lt.D3Q19())
    sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
    nodes = n ** 3
    kernel = sim._native.plan.kernel_name()
    resident, resident_stride = sim._native.plan.resident_enabled()
    sim(args.warmup)
    # the plain copy rate of this device (roofline.copy_ceiling_GBps) is measured before the timed batches
    ceiling = copy_ceiling(device)
    # one more untimed K-step call sizes the batches
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    sim(args.steps)
    torch.cuda.synchronize(device)
    repeat = repeats_for(time.perf_counter() - t0)
    pre_timed = None if args.no_verify else flow.f.clone()   # lettuce's convention: post-streaming
    sim._native.fused_events = (start, end)

    batch_s, batch_fused_ms, batch_info = [], [], []
    for _ in range(max(1, args.batches)):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(repeat):
            sim(args.steps)
        torch.cuda.synchronize(device)
        batch_s.append((time.perf_counter() - t0) / repeat)      # seconds per K steps
        if args.steps > 1:
            batch_info.append(sim._native.plan.last_run_info())
            batch_fused_ms.append(start.elapsed_time(end))       # the fused launches of the batch's last call
    # lettuce's convention for flow.f is post-streaming; the engine keeps the populations post-collision between
    # batches and streams when flow.f is read (Flow.f): that one pass, timed here, is what a caller pays per look
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    _ = flow.f
    torch.cuda.synchronize(device)
    present_ms = (time.perf_counter() - t0) * 1e3
    mid = median_index(batch_s)
    elapsed = batch_s[mid]
    mlups = args.steps * nodes / 1e6 / elapsed

    roofline, check = None, None
    if args.steps > 1:
        # what the events bracketed: the two-step launches (two lattice updates per node each) when
        # lt_run paired its fused steps, else the single-step launches
        info = batch_info[mid]
        paired = info["two_step_launches"] > 0
        launches = info["two_step_launches"] if paired else info["single_step_launches"]
        updates_per_launch = 2 if paired else 1
        fused_ms = batch_fused_ms[mid] / launches
        hbm_bytes = bytes_per_node * nodes                     # one read + one write of every population
        physical = hbm_bytes / (fused_ms * 1e-3) / 1e9
        achieved = physical * updates_per_launch               # 152 B per node and lattice UPDATE
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic_from_profile(kernel),
                    "kernel": kernel, "avg_launch_ms": round(fused_ms, 5),
                    "algorithmic_bytes_per_launch": hbm_bytes * updates_per_launch,
                    "lattice_updates_per_node_per_launch": updates_per_launch,
                    "hbm_bytes_per_launch_required": hbm_bytes,
                    "physical_GBps": round(physical, 1),
                    "physical_frac": round(physical / HBM_PEAK_GBS, 4),
                    "launches_timed": launches, "source_hash": source_hash(),
                    "populations": (f"engine-owned ping-pong buffers, {resident_stride - nodes} elements of padding between "
                                    f"populations (lt_resident_*); flow.f / flow.f_next stay the reference's dense "
                                    f"[q, *res] tensors" if resident else "the caller's dense [q, *res] tensors")}
        if roofline["traffic"]:
            # what HBM really carried per second during the launch (PMC bytes / live duration)
            real = roofline["traffic"] / (fused_ms * 1e-3) / 1e9
            roofline["hbm_traffic_GBps"] = round(real, 1)
            roofline["hbm_traffic_frac_of_peak"] = round(real / HBM_PEAK_GBS, 4)
        roofline["note"] = ("achieved / frac = SURVEY 8(d)'s 152 B per node and lattice update x the updates of one launch / "
                            "launch time" + ("; the launch performs two lattice updates per node with the intermediate "
                                             "state in LDS, so frac may exceed 1: physical_GBps / physical_frac = "
                                             "populations read once + written once per launch / launch time is the rate "
                                             "HBM has to sustain (<= peak), hbm_traffic_* what the PMC counters saw"
                                             if paired else ""))
        roofline["copy_ceiling_GBps"] = round(ceiling, 1)
        roofline["physical_frac_of_copy_ceiling"] = round(physical / ceiling, 4)
        # ---- self-check of the timed result (untimed) ------------------------------------------
        mass = float(sim._native.plan.mass(flow.f))             # device reductions (lt_mass, lt_kinetic_energy)
        energy = float(lt.IncompressibleKineticEnergy(flow)())
        checked = len(batch_s) * repeat * args.steps
        check = {"verified": None, "final_mass_lu": mass, "final_E_pu": energy, "steps_checked": checked}
        if pre_timed is not None:
            final = flow.f.clone()
            flow2 = lt.TaylorGreenVortex(ctx, [n, n, n], 1600, 0.1, lt.D3Q19())
            flow2.f = pre_timed
            sim2 = lt.Simulation(flow2, lt.BGKCollision(flow2.units.relaxation_parameter_lu), [])
            sim2._native.plan.set_two_step(0)                  # one lattice update per launch, the caller's dense buffers
            sim2(checked)
            info2 = sim2._native.plan.last_run_info()
            check["verified"] = bool(torch.equal(flow2.f, final)) and info2["two_step_launches"] == 0
            check["how"] = ("the same steps repeated from a copy of the pre-timed populations with the one-step "
                            f"kernel ({sim2._native.plan.kernel_name()}) on dense buffers, torch.equal on all populations")
            del flow2, sim2, final, pre_timed

    passes = ("K fused stream-collide steps per call, continuing from the post-collision populations of the call before; the "
              "streaming pass that presents flow.f in lettuce's post-streaming convention runs when flow.f is read (not "
              "inside the timed batches)")
    if roofline and roofline.get("lattice_updates_per_node_per_launch", 0) == 2:
        passes = ("K fused stream-collide steps (K streamings, K collisions) as K/2 two-step launches (+1 single when K "
                  "is odd), continuing from the post-collision populations of the previous call; the streaming pass "
                  "that presents flow.f in lettuce's post-streaming convention runs when flow.f is read -- here once, "
                  "after the timed batches (presentation_pass_ms); ms_per_step_if_read_after_every_batch adds it to "
                  "every K-step call")
    line = {
        "metric": "MLUPS (million lattice updates/s) D3Q19 256³ TGV; achieved HBM GB/s vs peak",
        "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"TaylorGreenVortex3D D3Q19 BGK fp32 {n}x{n}x{n} ({nodes} nodes per GPU), Re=1600 Ma=0.1",
                   "global_resolution": [n, n, n], "parallelism": "single GPU", "passes_per_batch": passes},
        "batches_ms_per_step": [round(t / args.steps * 1e3, 5) for t in batch_s],
        "timing": (f"median of {len(batch_s)} timed batches; a batch is the {args.steps}-step call repeated {repeat}x back "
                   f"to back between device synchronisations (>= {int(MIN_BATCH_S * 1e3)} ms per batch), every number is "
                   f"per {args.steps} steps; the device's plain copy rate (roofline.copy_ceiling_GBps) and one untimed "
                   f"{args.steps}-step call that sizes the batches come between the warm-up steps and the timed batches"),
        "repeats_per_batch": repeat,
        "roofline": roofline,
        "presentation_pass_ms": round(present_ms, 4),
        "ms_per_step_if_read_after_every_batch": round((elapsed * 1e3 + present_ms) / args.steps, 5),
    }
    if check is not None:
        line.update(check)
    del sim, flow
    torch.cuda.empty_cache()
    if not args.no_other_configs:
        try:
            line["other_configs"] = other_configs(lt, device)
        except Exception as exc:                       # the headline line must not depend on the extra rows
            line["other_configs"] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
        torch.cuda.empty_cache()
    line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(n, args.cpu_baseline_steps)
    print(json.dumps(line), flush=True)


# ---- N > 1 (and the one-GPU rehearsal of that path) -------------------------------------------
class Ranks:
    """the collectives the candidate loop needs, on the ranks' devices (RCCL) or on the host (gloo: the CPU test of
    the loop, tests/test_bench_failsoft.py)"""

    def __init__(self, dist, world, rank, local_rank, device):
        self.dist, self.world, self.rank, self.local_rank, self.device = dist, world, rank, local_rank, device
        self.cuda = device.type == "cuda"

    def barrier(self):
        if self.cuda:
            self.dist.barrier(device_ids=[self.local_rank])
            torch.cuda.synchronize(self.device)
        else:
            self.dist.barrier()

    def all_ranks(self, flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def max_over_ranks(self, seconds: float) -> float:
        t = torch.tensor([seconds], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, value: float):
        mine = torch.tensor([value], dtype=torch.float64, device=self.device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [round(float(x.item()), 6) for x in out]

    def devices(self):
        """what every rank computes on, all-gathered: the device's UUID (RCCL ranks) or host + pid (the CPU test) -- the
        record that the job really ran on `world` DISTINCT devices"""
        import hashlib
        import socket
        if self.cuda:
            ident = str(getattr(torch.cuda.get_device_properties(self.device), "uuid", "")) or f"cuda:{self.device.index}"
            ident = f"{socket.gethostname()}/{ident}"
        else:
            ident = f"{socket.gethostname()}/cpu-process-{os.getpid()}"
        mine = torch.tensor(list(hashlib.sha256(ident.encode()).digest()[:8]), dtype=torch.uint8, device=self.device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        ids = [bytes(t.cpu().tolist()).hex() for t in out]
        return {"ranks": self.world, "distinct_devices": len(set(ids)), "device_ids_sha256_8": ids,
                "rank0_device": ident}

    def release(self):
        if self.cuda:
            torch.cuda.empty_cache()


def slab_bench(args, lt, dist, world, rank, local_rank, device):
    flow_name, stencil_name, dtype_name, bytes_per_node, block = WORKLOADS[args.workload]
    dtype = getattr(torch, dtype_name)
    ctx = lt.Context(device=device, dtype=dtype, use_native=True)
    n = args.size
    if n == 256:
        global_res = [block[0], block[1], block[2] * world]    # cfg3: N = 8 is BASELINE configs[2] (512^3)
    else:
        global_res = [n, n, n * world]
    slab = lt.ZSlab(global_res)
    nodes_per_rank = global_res[0] * global_res[1] * slab.nz_local
    ranks = Ranks(dist, world, rank, local_rank, device)

    def build(driver, transport):
        if args.workload == "cfg5":
            flow = lt.DoublyPeriodicShear3D(ctx, slab.extended_resolution, 10000, 0.1, slab=slab)
        else:
            flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab)
        coll = lt.BGKCollision(flow.units.relaxation_parameter_lu)
        if driver == "two-step":
            # "rccl": the direct schedule (edge launch fed from the receive buffers, then the planes in between)
            # "rccl-edges": edge launches + pack / unpack beside the interior launch (rounds 1-2)
            # "rccl-signalled": one launch per double step whose edge workgroups run first and release the exchange
            # "window-fused": edge launches store into the neighbour's window themselves, one stream
            # "copy-2streams": the copy transport with a stream of its own per direction (two copy engines side by side)
            return lt.TwoStepSlabSimulation(flow, coll, slab, overlap=not args.no_overlap,
                                            transport=transport.split("-")[0],
                                            fused_remote_pack=transport.endswith("-fused"),
                                            signalled=transport.endswith("-signalled"),
                                            direct=transport in ("rccl", "copy", "copy-2streams"),
                                            copy_streams=2 if transport == "copy-2streams" else None)
        return lt.SlabSimulation(flow, coll, slab, overlap=not args.no_overlap, transport=transport)

    # Candidates: slab driver (two lattice updates per launch and one halo exchange per two updates / one update per
    # launch) x transport.  The single-step driver over RCCL comes first and is the reference: see the module
    # docstring ("N > 1 fails soft").
    transports = {"auto": ["rccl"], "all": ["rccl", "window"]}.get(args.transport, [args.transport])
    drivers = ["two-step", "single-step"] if args.driver == "auto" else [args.driver]
    wanted = [(d, t) for d in drivers for t in transports]
    if args.transport in ("auto", "all") and "two-step" in drivers and not args.no_overlap:
        # the direct schedule with the halo messages moved by copy engines into windows the neighbours mapped
        # (lettuce_amd/_slab.py, _CopyWindow): no RCCL kernel beside the sweep.  A candidate like the others: it must be
        # bit-identical to single-step/rccl twice and faster than the held line; a message that does not arrive ends in
        # a time-out the driver raises after the batch (nothing traps), i.e. in config.transport.failures
        wanted.insert(wanted.index(("two-step", "rccl")) + 1, ("two-step", "copy"))
        if world > 1:
            # across real links the two messages of an exchange travel one after the other when they share a stream
            # (an SDMA engine moves ~50 GB/s: 2 x 0.4 ms against a 0.5 ms sweep); with a stream per direction two
            # engines work side by side.  On ONE device the extra streams cost the sweep (profiles/
            # r04g_slab_stream_count.jsonl), so the rehearsal does not offer it.  Never measured across real links: a
            # candidate like the others, kept only if it is bit-identical twice and faster.
            wanted.insert(wanted.index(("two-step", "copy")) + 1, ("two-step", "copy-2streams"))
    if args.transport == "all" and "two-step" in drivers and not args.no_overlap:
        at = wanted.index(("two-step", "rccl")) + 1
        wanted[at:at] = [("two-step", "rccl-edges"), ("two-step", "rccl-signalled")]
    if "two-step" in drivers and "window" in transports:
        wanted.insert(wanted.index(("two-step", "window")) + 1, ("two-step", "window-fused"))
    if ("single-step", "rccl") in wanted:
        wanted = [("single-step", "rccl")] + [w for w in wanted if w != ("single-step", "rccl")]

    what = (f"{flow_name} {stencil_name} BGK {'fp32' if dtype == torch.float32 else 'fp64'} "
            f"{global_res[0]}x{global_res[1]}x{global_res[2]} ({nodes_per_rank} nodes per GPU)")
    # the CPU baseline of the N = 1 line, measured on rank 0's host cores before the candidates while the other ranks
    # wait in the first collective of the loop (same per-GPU node count; ~20 s)
    cpu_row = None
    if rank == 0 and not args.no_cpu_baseline and args.workload == "cfg3":
        cpu_row = cpu_baseline(256 if n == 256 else n, args.cpu_baseline_steps)
        cpu_row["sample"] += "; measured on rank 0 before the process group's first collective"
    traffic_key = {"cfg3": "slab_tgv3d_d3q19_bgk_f32_512x512x64", "cfg5": "slab_shear3d_d3q19_bgk_f64_384x384x96"}.get(args.workload)
    candidate_loop(args, ranks, wanted, build, what, global_res, nodes_per_rank, bytes_per_node,
                   "f32" if dtype == torch.float32 else "f64", cpu_baseline_row=cpu_row,
                   traffic_workload=traffic_key if n == 256 else None)
    dist.barrier(device_ids=[local_rank])
    dist.destroy_process_group()


def candidate_loop(args, ranks, wanted, build, what, global_res, nodes_per_rank, bytes_per_node, dtype_tag,
                   probe_steps=None, cpu_baseline_row=None, traffic_workload=None):
    """Try the candidates in order, time each one that passes its checks, hold the best line; rank 0 prints the held
    line at the end -- or the watchdog does -- whatever the later candidates did.  ``build(driver, transport)``
    returns a slab driver (callable with a step count, ``local_f()``, ``engine.kernel_name()``)."""
    world, rank, device = ranks.world, ranks.rank, ranks.device
    barrier, all_ranks, max_over_ranks = ranks.barrier, ranks.all_ranks, ranks.max_over_ranks
    probe, checks, failures, seen_box = {}, {}, {}, {}

    def nothing_held(why):
        return {"metric": "MLUPS (million lattice updates/s) D3Q19 256³ TGV; achieved HBM GB/s vs peak",
                "value": 0.0, "unit": "MLUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": dtype_tag, "data": "synthetic",
                "config": {"workload": what, "global_resolution": global_res, "parallelism": f"z-slab x{world}",
                           "transport": {"chosen": None, "warmup_ms_per_step": probe, "checks": checks,
                                         "failures": failures, "ranks_seen": seen_box.get("seen", world),
                                         "aborted": why}},
                "roofline": None, "cpu_baseline": cpu_baseline_row, "verified": None}

    global HANG_EXIT_CODE
    HANG_EXIT_CODE = int(getattr(args, "hang_exit_code", 0))
    held = HeldLine(rank, nothing_held)
    seen = seen_box["seen"] = ranks.devices()             # collective: before any candidate
    reference = {}                     # the first candidate's populations after the probe and after its timed batches
    probe_steps = max(args.warmup, 60) if probe_steps is None else probe_steps
    window_ok = None

    def describe(driver, transport):
        how_copy = ("halo messages moved by device-to-device copies without compute units (copy engines) into receive "
                    "windows the neighbours mapped through HIP IPC, arrival counters written in stream order")
        how = {"rccl": "RCCL send/recv of the halo messages",
               "rccl-edges": "RCCL send/recv ghost planes; edge launches, pack and unpack beside the interior launch",
               "rccl-signalled": "RCCL send/recv ghost planes, released by the edge workgroups of the one launch per double step",
               "copy": how_copy, "copy-2streams": how_copy,
               }.get(transport, "one-sided ghost-plane stores into peer windows (xGMI peer access)"
                     + (", issued by the edge launches" if transport.endswith("-fused") else ""))
        if transport == "copy-2streams":
            how = how_copy + ", one stream per direction"
        if driver == "two-step" and transport in ("rccl", "copy", "copy-2streams"):
            how += (" straight out of / into the buffers the edge launch writes / reads (no pack, no unpack); edge launch, "
                    "then the planes in between, on one stream")
        how += ("; two lattice updates per launch, one exchange per two updates" if driver == "two-step"
                else "; one exchange per update")
        return f"z-slab x{world}, {how}" + ("" if not args.no_overlap else " (no overlap)")

    def make_line(name, driver, transport, batch_s, repeat, kernel, verified):
        mid = median_index(batch_s)
        elapsed = batch_s[mid]
        mlups = args.steps * nodes_per_rank * world / 1e6 / elapsed
        # per-rank fused-kernel rate is not separable from the exchange here: the whole-step rate of one rank.
        # achieved = SURVEY 8(d)'s bytes per node and lattice update x updates / time, like the N = 1 line's; the
        # two-step driver does two updates per launch, so HBM physically carries half of that
        eff = bytes_per_node * nodes_per_rank * args.steps / elapsed / 1e9
        per_launch = 2 if driver == "two-step" else 1
        # HBM bytes per launch of the dominant kernel (the sweep between the cuts / the interior launch) from the
        # committed PMC passes of the one-rank rehearsal of this workload (profiles/traffic.json; null when the file
        # belongs to other kernel sources)
        traffic = traffic_from_profile(kernel, traffic_workload) if traffic_workload else None
        roofline = {"bound": "hbm", "achieved": round(eff, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(eff / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kernel,
                    "lattice_updates_per_node_per_launch": per_launch, "physical_GBps": round(eff / per_launch, 1),
                    "physical_frac": round(eff / per_launch / HBM_PEAK_GBS, 4),
                    "note": f"whole-step rate per GPU (includes the halo exchange): {bytes_per_node} B per node and lattice "
                            "update / time; physical_* = populations read once + written once per launch / time"}
        return {
            "metric": "MLUPS (million lattice updates/s) D3Q19 256³ TGV; achieved HBM GB/s vs peak",
            "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_tag, "data": "synthetic",
            "config": {"workload": what, "global_resolution": global_res, "parallelism": describe(driver, transport),
                       "passes_per_batch": "K fused stream-collide steps per call, continuing from the post-collision "
                                           "populations (ghost planes exchanged) of the call before; the streaming pass "
                                           "that presents the populations in lettuce's post-streaming convention runs "
                                           "when they are read (not inside the timed batches)",
                       "transport": {"chosen": name, "warmup_ms_per_step": probe, "checks": checks,
                                     "failures": failures, "ranks_seen": seen}},
            "batches_ms_per_step": [round(t / args.steps * 1e3, 5) for t in batch_s],
            "timing": (f"median of {len(batch_s)} timed batches (barrier + device synchronise on both sides, max over "
                       f"ranks); a batch is the {args.steps}-step call repeated {repeat}x back to back "
                       f"(>= {int(MIN_BATCH_S * 1e3)} ms per batch), every number is per {args.steps} steps"),
            "repeats_per_batch": repeat,
            "roofline": roofline,
            "verified": verified,
            "cpu_baseline": cpu_baseline_row,
        }

    for index, (driver, transport) in enumerate(wanted):
        name = f"{driver}/{transport}"
        first = index == 0
        started = time.time()
        budget = args.first_candidate_budget if first else args.candidate_budget
        held.arm(budget + args.watchdog_grace, name)

        def over_budget():
            """between batches: has ANY rank used up this candidate's wall budget?  (collective)"""
            return not all_ranks(time.time() - started <= budget)

        try:
            if transport.startswith("window"):
                if window_ok is None:
                    # preflight on every rank before the collective rendezvous inside build(): a rank
                    # that cannot allocate peer-mappable memory must not leave the others waiting
                    try:
                        import torch.distributed._symmetric_memory as symm
                        symm.empty(1024, dtype=torch.float32, device=device)
                        ok = True
                    except Exception as exc:
                        ok = False
                        failures[name] = f"unavailable: {type(exc).__name__}: {str(exc)[:120]}"
                    window_ok = all_ranks(ok)
                if not window_ok:
                    failures.setdefault(name, "unavailable: no peer-mappable memory on some rank")
                    continue
            # ---- build + connection set-up ----
            cand, err = None, None
            try:
                held.at("build")
                cand = build(driver, transport)
                held.at("first three steps (connection set-up)")
                cand(3)                         # connection set-up, first launches: not timed
            except Exception as exc:            # unsupported grid for the two-step kernel, no symmetric memory ...
                err = f"unavailable: {type(exc).__name__}: {str(exc)[:120]}"
            if not all_ranks(err is None):
                failures[name] = err or "unavailable on another rank"
                cand = None
                ranks.release()
                continue
            # ---- warm-up probe: two batches of >= 60 steps, the faster one counts ----
            best, err = None, None
            for probe_round in range(2):
                held.at(f"warm-up probe {probe_round + 1} of 2")
                barrier()
                t0 = time.perf_counter()
                try:
                    cand(probe_steps)
                except Exception as exc:        # e.g. the signalled driver's time-out: raised after the batch's
                    err = f"failed: {type(exc).__name__}: {str(exc)[:120]}"   # exchanges, so the ranks stay in step
                barrier()
                t = max_over_ranks(time.perf_counter() - t0)
                best = t if best is None else min(best, t)
                if not all_ranks(err is None) or over_budget():
                    err = err or "failed on another rank / over its wall budget"
                    break
            if err is not None:
                failures[name] = err
                cand = None
                ranks.release()
                continue
            probe[name] = round(best / probe_steps * 1e3, 5)
            state = cand.local_f()
            if first:
                reference["probe"] = state.clone()
            elif "probe" in reference:
                same = all_ranks(torch.equal(state, reference["probe"]))
                checks[name] = f"bit-identical to {wanted[0][0]}/{wanted[0][1]} after the warm-up probe" if same else "MISMATCH after the warm-up probe: rejected"
                if not same:
                    cand = None
                    ranks.release()
                    continue
            del state
            # ---- the timed region: W more warm-up steps, then `batches` times (R x) exactly K steps ----
            cand(args.warmup)
            barrier()
            t0 = time.perf_counter()
            cand(args.steps)                    # sizes the batches (untimed)
            barrier()
            # later candidates repeat as often as the first one did: the same number of steps, so that the populations
            # after the timed batches can be compared too
            repeat = reference.get("repeat") or repeats_for(max_over_ranks(time.perf_counter() - t0))
            batch_s, err = [], None
            for batch in range(max(1, args.batches)):
                held.at(f"timed batch {batch + 1} of {max(1, args.batches)}")
                barrier()
                t0 = time.perf_counter()
                try:
                    for _ in range(repeat):
                        cand(args.steps)
                except Exception as exc:
                    err = f"failed in the timed batches: {type(exc).__name__}: {str(exc)[:120]}"
                barrier()
                batch_s.append(max_over_ranks(time.perf_counter() - t0) / repeat)
                if not all_ranks(err is None) or over_budget():
                    err = err or "failed on another rank / over its wall budget"
                    break
            if err is not None:
                failures[name] = err
                cand = None
                ranks.release()
                continue
            # ---- after the timed batches: the same number of steps as the reference candidate has done ----
            state = cand.local_f()
            verified = None
            if first:
                reference["final"] = state.clone()
                reference["repeat"] = repeat
            elif "final" in reference and reference["repeat"] == repeat:
                verified = all_ranks(torch.equal(state, reference["final"]))
                checks[name] += ("; bit-identical after the timed batches too" if verified
                                 else "; MISMATCH after the timed batches: rejected")
                if not verified:
                    cand = None
                    ranks.release()
                    continue
            # what every rank ended with, for the record: RCCL really saw `world` ranks, each with its own slab
            sums = ranks.gather(float(state.double().sum()))
            del state
            kernel = cand.engine.kernel_name() if hasattr(cand.engine, "kernel_name") else type(cand.engine).__name__
            line = make_line(name, driver, transport, batch_s, repeat, kernel, verified)
            line["config"]["transport"]["rank_checksums"] = sums
            if getattr(cand, "_cw", None) is not None:
                line["config"]["transport"]["copy_engine"] = sorted(cand._cw.engines_used)
                line["config"]["transport"]["copy_streams"] = cand._cw.n_streams
            if held.line is None or line["value"] > held.line["value"]:
                held.line = line
            else:
                checks[name] = checks.get(name, "") + "; not faster than the held line"
            cand = None
            ranks.release()
        except Exception as exc:                # anything unforeseen in a candidate must not cost the held line
            failures[name] = f"failed: {type(exc).__name__}: {str(exc)[:160]}"
            if first:
                raise
        finally:
            held.disarm()

    if held.line is None:
        held.emit_nothing("no usable slab configuration")
        raise SystemExit(f"no usable slab configuration: {failures}")
    held.emit()
    reference.clear()


if __name__ == "__main__":
    main()
