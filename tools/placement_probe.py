"""Does it matter WHERE the population buffers lie?  One plan, one process, several pairs of padded buffers kept alive
at the same time (cfg2: 256^3 D3Q19 fp32, two-step kernel), launches alternating between the pairs: ms per launch and
the addresses.  usage: placement_probe.py [pairs]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
plan = Plan("D3Q19", torch.float32, "bgk", [256, 256, 256], [], device=dev)
plan.set_two_step(1, 0)
plan.set_population_stride(-(-(256 ** 3 + 32832) // 64) * 64)
pairs, spacers = [], []
for k in range(n):
    f = plan.empty_populations(); f.uniform_(0.04, 0.06)
    g = plan.empty_populations(); g.zero_()
    pairs.append((f, g))
    spacers.append(torch.empty((k + 1) * 37 * 1024 * 1024 // 4, device=dev))      # odd-sized blocks in between
times = [[] for _ in pairs]
for rep in range(6):
    for k, (f, g) in enumerate(pairs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        plan.stream_collide_twice(f, g, 0.6)
        e0.record()
        for _ in range(10):
            plan.stream_collide_twice(f, g, 0.6)
            plan.stream_collide_twice(g, f, 0.6)
        e1.record(); torch.cuda.synchronize()
        times[k].append(round(e0.elapsed_time(e1) / 20, 4))
for k, (f, g) in enumerate(pairs):
    print(json.dumps({"pair": k, "f": hex(f.data_ptr()), "g": hex(g.data_ptr()), "f_mod_2MiB": f.data_ptr() % (2 << 20),
                      "g_minus_f_MiB": round((g.data_ptr() - f.data_ptr()) / 2 ** 20, 3), "ms_per_launch": times[k][1:]}), flush=True)
