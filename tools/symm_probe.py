"""Probe: does torch symmetric memory (peer-mapped buffers + signal pads) work on this box?

Two ranks on ONE GPU (gloo group for the rendezvous).  Each rank allocates a symmetric buffer,
maps the peer's, stores a pattern into the peer's buffer from a kernel, signals, waits, checks.
Prints one JSON line per rank.  Run:  python tools/symm_probe.py   (spawns its own 2 ranks)
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, n_mb):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = {"rank": rank}
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        import torch.distributed._symmetric_memory as symm

        n = n_mb * (1 << 20) // 4
        buf = symm.empty(n, dtype=torch.float32, device="cuda:0")
        hdl = symm.rendezvous(buf, dist.group.WORLD)
        out["rendezvous"] = "ok"
        out["signal_pad_size"] = hdl.signal_pad_size
        peer = (rank + 1) % world
        src = (rank - 1) % world
        remote = hdl.get_buffer(peer, (n,), torch.float32, 0)
        buf.zero_()
        torch.cuda.synchronize()
        hdl.barrier(0)
        payload = torch.full((n,), float(rank + 1), device="cuda:0")
        torch.cuda.synchronize()
        # direct store into the peer's buffer, then signal / wait
        t0 = time.perf_counter()
        reps = 20
        ok = True
        for it in range(reps):
            payload.fill_(float(rank + 1 + 10 * it))
            remote.copy_(payload)
            hdl.put_signal(peer, 0)
            hdl.wait_signal(src, 0)
            got = buf.clone()
            hdl.barrier(1)  # peer may overwrite only after everybody has read
            torch.cuda.synchronize()
            expect = float(src + 1 + 10 * it)
            if not bool((got == expect).all()):
                ok = False
                out["first_bad"] = {"it": it, "got": got[:4].tolist(), "expect": expect}
                break
        out["exchange_ok"] = ok
        out["ms_per_exchange_incl_sync"] = (time.perf_counter() - t0) / reps * 1e3
        # device-side cost of copy+signal+wait without host syncs
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hdl.barrier(0)
        s.record()
        for it in range(reps):
            remote.copy_(payload)
            hdl.put_signal(peer, 0)
            hdl.wait_signal(src, 0)
        e.record()
        torch.cuda.synchronize()
        out["device_ms_per_exchange"] = s.elapsed_time(e) / reps
        out["mb"] = n_mb
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # report, do not hang the other rank longer than its timeouts
        out["error"] = f"{type(exc).__name__}: {exc}"[:600]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(world, 29531, 10), nprocs=world, join=True)
