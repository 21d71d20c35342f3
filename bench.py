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
A "step" is one full lattice update (collide, stream) of every node.  The timed region is
exactly K steps between barrier + device synchronise, max over ranks; value = all nodes of all
ranks * K / time.  One JSON line is printed by rank 0.

Launching.  ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts its
own N ranks (a ``torch.distributed.run`` child process, before this process touches the GPU),
relays rank 0's JSON line and the exit code; fewer than N visible devices is an error, never a
silent N = 1 run.  Under a launcher (WORLD_SIZE set) the process is one rank.

Timing.  Five timed batches of exactly K steps each (barrier + device synchronise on both sides,
max over ranks); ``ms_per_step`` / ``value`` are those of the median batch, all
five are listed in ``batches_ms_per_step``.

roofline: HBM-bound kernel.  ``achieved`` = the bytes a launch of the dominant kernel has to move
through HBM -- every population read once and written once, 2 * 19 * 4 = 152 B per node and
launch, however many lattice updates the launch performs on them -- divided by
its average duration, measured with HIP events recorded on the launch stream around the fused
launches of the median batch.  ``frac`` = achieved / 8 TB/s, a physical HBM fraction (<= 1).  The
two-step kernel does two lattice updates per launch with the intermediate state in LDS; the rate
in algorithmic bytes of lattice updates (152 B per node and UPDATE) is reported separately as
``algorithmic_update_GBps`` and may exceed the HBM peak.  ``traffic`` = HBM bytes per launch from
the rocprofv3 PMC passes in profiles/traffic.json, used only if that file was produced from the
kernel sources of this build (hash of lettuce_amd/csrc), else null.
verified: after the timed batches the same 5 K steps are repeated, untimed, from a copy of the
pre-timed populations with the one-step kernel only (``set_two_step(0)``); ``verified`` says
whether the two final states are bit-identical; final mass and kinetic energy are in the line.
cpu_baseline: the CPU oracle (a torch-CPU restatement of the reference's path, validated
against the reference) timed on this box's host cores on the same 256^3 workload for a few
steps (rank 0, N = 1 only), at all host threads and at 32 threads; the better one is ``value``.
"""
import argparse
import json
import os
import sys
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
BYTES_PER_NODE = 2 * 19 * 4      # D3Q19 fp32: every population read once, written once


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batches", type=int, default=5,
                    help="timed batches of --steps steps each; the median one is reported")
    ap.add_argument("--size", type=int, default=256, help="nodes per side of the per-GPU block")
    ap.add_argument("--cpu-baseline-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the untimed re-run with the one-step kernel (profiling passes)")
    ap.add_argument("--no-overlap", action="store_true", help="slab path: exchange without overlap")
    ap.add_argument("--transport", choices=["auto", "all", "rccl", "window"],
                    default=os.environ.get("LT_BENCH_TRANSPORT", "auto"),
                    help="slab path: ghost-plane transport (lettuce_amd/_slab.py).  auto = RCCL send/recv only "
                         "(both slab drivers are tried in the warm-up, results must be bit-identical, the faster "
                         "one is timed); all = also the one-sided peer-window transports, which are faster in "
                         "the one-GPU rehearsal but have never run across real xGMI links and end in a device "
                         "trap if a signal is lost -- opt in with --transport all or LT_BENCH_TRANSPORT=all")
    ap.add_argument("--driver", choices=["auto", "single-step", "two-step"], default="auto",
                    help="slab path: restrict the candidates to one slab driver")
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


def traffic_from_profile(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
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
            if (row.get("workload") == "tgv3d_d3q19_bgk_f32_256" and row.get("hbm_bytes_per_launch")
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
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)

    import lettuce_amd as lt
    ctx = lt.Context(device=device, dtype=torch.float32, use_native=True)
    n = args.size
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def barrier():
        if distributed:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize(device)

    if not distributed:
        flow = lt.TaylorGreenVortex(ctx, [n, n, n], 1600, 0.1, lt.D3Q19())
        sim = lt.Simulation(flow, lt.BGKCollision(flow.units.relaxation_parameter_lu), [])
        global_res = [n, n, n]
        nodes_per_rank = n ** 3
        kernel = sim._native.plan.kernel_name()
        sim(args.warmup)
        # the plain copy rate of this device (roofline.copy_ceiling_GBps) is measured before the timed batches
        ceiling = copy_ceiling(device)
        pre_timed = None if args.no_verify else flow.f.clone()   # lettuce's convention: post-streaming
        sim._native.fused_events = (start, end)
        step = sim
        parallelism = "single GPU"
    else:
        if n == 256:
            global_res = [512, 512, 64 * world]    # N = 8: BASELINE configs[2] (512^3)
        else:
            global_res = [n, n, n * world]
        slab = lt.ZSlab(global_res)

        def build(driver, transport):
            flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab)
            coll = lt.BGKCollision(flow.units.relaxation_parameter_lu)
            if driver == "two-step":
                # "window-fused": edge launches store into the neighbour's window themselves, one stream
                # "rccl-signalled": one launch per double step whose edge workgroups run first and release the exchange
                return lt.TwoStepSlabSimulation(flow, coll, slab, overlap=not args.no_overlap,
                                                transport=transport.split("-")[0],
                                                fused_remote_pack=transport.endswith("-fused"),
                                                signalled=transport.endswith("-signalled"))
            return lt.SlabSimulation(flow, coll, slab, overlap=not args.no_overlap, transport=transport)

        def all_ranks(flag: bool) -> bool:
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        # Candidates: slab driver (two lattice updates per launch and one halo exchange per two
        # updates / one update per launch) x transport (RCCL send/recv / one-sided peer windows).
        # The single-step driver over RCCL is the reference: every candidate does 3 + W warm-up
        # steps from the same initial state (two timed batches of max(W, 60) steps, max over ranks), must end with populations
        # bit-identical to the reference's on every rank, and the fastest eligible one runs the
        # timed K steps.  All warm-up rates go into the JSON line.
        transports = {"auto": ["rccl"], "all": ["rccl", "window"]}.get(args.transport, [args.transport])
        drivers = ["two-step", "single-step"] if args.driver == "auto" else [args.driver]
        wanted = [(d, t) for d in drivers for t in transports]
        if "two-step" in drivers and "rccl" in transports and not args.no_overlap:
            wanted.insert(wanted.index(("two-step", "rccl")) + 1, ("two-step", "rccl-signalled"))
        if "two-step" in drivers and "window" in transports:
            wanted.insert(wanted.index(("two-step", "window")) + 1, ("two-step", "window-fused"))
        if args.driver == "auto" and args.transport in ("auto", "all"):
            wanted = [("single-step", "rccl")] + [w for w in wanted if w != ("single-step", "rccl")]
        finals, probe = {}, {}
        window_ok = None
        probe_steps = max(args.warmup, 60)
        for driver, transport in wanted:
            name = f"{driver}/{transport}"
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
                        probe[name] = f"unavailable: {type(exc).__name__}: {str(exc)[:120]}"
                    window_ok = all_ranks(ok)
                if not window_ok:
                    probe.setdefault(name, "unavailable: no peer-mappable memory on some rank")
                    continue
            try:
                cand = build(driver, transport)
                cand(3)                         # connection set-up, first launches: not timed
                ok = True
            except Exception as exc:            # unsupported grid for the two-step kernel, no symmetric memory ...
                cand, ok = None, False
                probe[name] = f"unavailable: {type(exc).__name__}: {str(exc)[:120]}"
            if not all_ranks(ok):
                probe.setdefault(name, "unavailable on another rank")
                cand = None
                continue
            best, failed = None, None
            for _ in range(2):                  # two batches of >= 60 steps, the faster one counts
                barrier()
                t0 = time.perf_counter()
                try:
                    cand(probe_steps)
                except Exception as exc:        # e.g. the signalled driver's time-out: raised after the batch's
                    failed = f"failed: {type(exc).__name__}: {str(exc)[:120]}"   # exchanges, so the ranks stay in step
                barrier()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                best = float(t.item()) if best is None else min(best, float(t.item()))
            if not all_ranks(failed is None):
                probe[name] = failed or "failed on another rank"
                cand = None
                torch.cuda.empty_cache()
                continue
            probe[name] = round(best / probe_steps * 1e3, 5)
            # keep only the final populations; candidates must not share the device while timed
            finals[name] = cand.local_f().clone()
            cand = None
            torch.cuda.empty_cache()
        if not finals:
            raise SystemExit(f"no usable slab configuration: {probe}")
        reference = finals.get("single-step/rccl")
        checks, eligible = {}, []
        for name, state in finals.items():
            if reference is not None and state is not reference:
                same = all_ranks(torch.equal(state, reference))
                checks[name] = "bit-identical to single-step/rccl" if same else "MISMATCH: rejected"
                if not same:
                    continue
            eligible.append(name)
        chosen = min(eligible, key=lambda k: probe[k])
        # what every rank ended the warm-up with, for the record: RCCL really saw `world` ranks, each
        # with its own slab (sum of the populations of the chosen candidate, per rank)
        mine = torch.tensor([float(finals[chosen].double().sum())], dtype=torch.float64, device=device)
        sums = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(sums, mine)
        rank_checksums = [round(float(x.item()), 6) for x in sums]
        finals.clear()
        reference = state = None
        torch.cuda.empty_cache()
        driver, transport = chosen.split("/")
        sim = build(driver, transport)              # fresh instance of the chosen configuration
        sim(args.warmup)
        nodes_per_rank = global_res[0] * global_res[1] * slab.nz_local
        kernel = sim.engine.kernel_name()
        step = sim
        how = ("RCCL send/recv ghost planes" if transport == "rccl"
               else "RCCL send/recv ghost planes, released by the edge workgroups of the one launch per double step"
               if transport == "rccl-signalled"
               else "one-sided ghost-plane stores into peer windows (xGMI peer access)"
               + (", issued by the edge launches" if transport.endswith("-fused") else ""))
        how += ("; two lattice updates per launch, one exchange per two updates" if driver == "two-step"
                else "; one exchange per update")
        parallelism = f"z-slab x{world}, {how}" + ("" if not args.no_overlap else " (no overlap)")
        transport_info = {"chosen": chosen, "warmup_ms_per_step": probe, "checks": checks,
                          "ranks_seen": world, "rank_checksums_after_warmup": rank_checksums}

    # ---- the timed region: `batches` times exactly K steps, each between barrier + synchronise ----
    batch_s, batch_fused_ms, batch_info = [], [], []
    for _ in range(max(1, args.batches)):
        barrier()
        t0 = time.perf_counter()
        step(args.steps)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        batch_s.append(float(t.item()))
        if not distributed and args.steps > 1:
            info = sim._native.plan.last_run_info()
            batch_info.append(info)
            batch_fused_ms.append(start.elapsed_time(end))
    # lettuce's convention for flow.f is post-streaming; the engine keeps the populations post-collision between
    # batches and streams when flow.f is read (Flow.f): that one pass, timed here, is what a caller pays per look
    present_ms = None
    if not distributed:
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        _ = flow.f
        torch.cuda.synchronize(device)
        present_ms = (time.perf_counter() - t0) * 1e3
    mid = median_index(batch_s)
    elapsed = batch_s[mid]
    total_nodes = nodes_per_rank * world
    mlups = args.steps * total_nodes / 1e6 / elapsed

    roofline = None
    check = None
    if not distributed and args.steps > 1:
        # what the events bracketed: the two-step launches (two lattice updates per node each) when
        # lt_run paired its fused steps, else the single-step launches
        info = batch_info[mid]
        paired = info["two_step_launches"] > 0
        launches = info["two_step_launches"] if paired else info["single_step_launches"]
        updates_per_launch = 2 if paired else 1
        fused_ms = batch_fused_ms[mid] / launches
        hbm_bytes = BYTES_PER_NODE * nodes_per_rank            # one read + one write of every population
        achieved = hbm_bytes / (fused_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic_from_profile(kernel),
                    "kernel": kernel, "avg_launch_ms": round(fused_ms, 5),
                    "hbm_bytes_per_launch_required": hbm_bytes,
                    "lattice_updates_per_node_per_launch": updates_per_launch,
                    "algorithmic_update_GBps": round(achieved * updates_per_launch, 1),
                    "launches_timed": launches, "source_hash": source_hash()}
        if roofline["traffic"]:
            # what HBM really carried per second during the launch (PMC bytes / live duration)
            real = roofline["traffic"] / (fused_ms * 1e-3) / 1e9
            roofline["hbm_traffic_GBps"] = round(real, 1)
            roofline["hbm_traffic_frac_of_peak"] = round(real / HBM_PEAK_GBS, 4)
        roofline["note"] = ("achieved = populations read once + written once per launch / launch time (a physical "
                            "HBM rate)" + ("; the launch performs two lattice updates per node with the intermediate "
                                           "state in LDS, algorithmic_update_GBps counts 152 B per node and update"
                                           if paired else ""))
        roofline["copy_ceiling_GBps"] = round(ceiling, 1)
        roofline["frac_of_copy_ceiling"] = round(achieved / ceiling, 4)
        # ---- self-check of the timed result (untimed) ------------------------------------------
        mass = float(sim._native.plan.mass(flow.f))             # device reductions (lt_mass, lt_kinetic_energy)
        energy = float(lt.IncompressibleKineticEnergy(flow)())
        check = {"verified": None, "final_mass_lu": mass, "final_E_pu": energy,
                 "steps_checked": len(batch_s) * args.steps}
        if pre_timed is not None:
            final = flow.f.clone()
            flow2 = lt.TaylorGreenVortex(ctx, [n, n, n], 1600, 0.1, lt.D3Q19())
            flow2.f = pre_timed
            sim2 = lt.Simulation(flow2, lt.BGKCollision(flow2.units.relaxation_parameter_lu), [])
            sim2._native.plan.set_two_step(0)                  # one lattice update per launch
            sim2(len(batch_s) * args.steps)
            info2 = sim2._native.plan.last_run_info()
            check["verified"] = bool(torch.equal(flow2.f, final)) and info2["two_step_launches"] == 0
            check["how"] = ("the same steps repeated from a copy of the pre-timed populations with the one-step "
                            f"kernel ({sim2._native.plan.kernel_name()}), torch.equal on all populations")
            del flow2, sim2, final, pre_timed
    elif distributed:
        # per-rank fused-kernel rate is not separable from the exchange here: the whole-step rate of one rank.
        # achieved = bytes HBM has to carry (populations read once + written once per LAUNCH; the two-step
        # driver does two lattice updates per launch) / time, a physical rate like the N = 1 line's
        eff = BYTES_PER_NODE * nodes_per_rank * args.steps / elapsed / 1e9
        per_launch = 2 if driver == "two-step" else 1
        roofline = {"bound": "hbm", "achieved": round(eff / per_launch, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(eff / per_launch / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kernel,
                    "lattice_updates_per_node_per_launch": per_launch, "algorithmic_update_GBps": round(eff, 1),
                    "note": "whole-step rate per GPU (includes halo exchange and the collide/stream passes of the "
                            "batch): populations read once + written once per launch / time; "
                            "algorithmic_update_GBps counts 152 B per node and update"}

    passes = ("K fused stream-collide steps per batch, continuing from the post-collision populations (ghost planes "
              "exchanged) of the batch before; the streaming pass that presents the populations in lettuce's "
              "post-streaming convention runs when they are read (not inside the timed batches)")
    if roofline and roofline.get("lattice_updates_per_node_per_launch", 0) == 2:
        # the timed call continues from the post-collision state the warm-up call left (lt_continue)
        passes = ("K fused stream-collide steps (K streamings, K collisions) as K/2 two-step launches (+1 single when K "
                  "is odd), continuing from the post-collision populations of the previous batch; the streaming pass "
                  "that presents flow.f in lettuce's post-streaming convention runs when flow.f is read -- here once, "
                  "after the timed batches (presentation_pass_ms); ms_per_step_if_read_after_every_batch adds it to "
                  "every batch")
    if rank == 0:
        line = {
            "metric": "MLUPS (million lattice updates/s) D3Q19 256³ TGV; achieved HBM GB/s vs peak",
            "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"TaylorGreenVortex3D D3Q19 BGK fp32 "
                                   f"{global_res[0]}x{global_res[1]}x{global_res[2]} "
                                   f"({nodes_per_rank} nodes per GPU), Re=1600 Ma=0.1",
                       "global_resolution": global_res, "parallelism": parallelism,
                       "passes_per_batch": passes},
            "batches_ms_per_step": [round(t / args.steps * 1e3, 5) for t in batch_s],
            "timing": f"median of {len(batch_s)} timed batches of {args.steps} steps"
                      + ("" if distributed else "; the device's plain copy rate (roofline.copy_ceiling_GBps) is measured "
                         "between the warm-up steps and the timed batches"),
            "roofline": roofline,
        }
        if present_ms is not None:
            line["presentation_pass_ms"] = round(present_ms, 4)
            line["ms_per_step_if_read_after_every_batch"] = round((elapsed * 1e3 + present_ms) / args.steps, 5)
        if check is not None:
            line.update(check)
        if distributed:
            line["config"]["transport"] = transport_info
        if not distributed and not args.no_cpu_baseline:
            del sim, flow
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline(n, args.cpu_baseline_steps)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
