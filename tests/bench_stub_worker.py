"""Stand-in for bench.py's per-rank body in the launcher test: joins a gloo group of WORLD_SIZE
ranks, all-reduces, and rank 0 prints one JSON line (no GPU, no engine)."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([rank + 1.0])
dist.all_reduce(t)
if "--fail" in sys.argv and rank == world - 1:
    sys.exit(7)
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": float(t.item()), "argv": sys.argv[1:]}), flush=True)
dist.destroy_process_group()
