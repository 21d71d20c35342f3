"""The two-step kernel at 256^3 D3Q19 fp32 with the populations spread apart (64 MB ... 1 GB between consecutive populations,
always off a power of two): what the distance between the 19 + 19 streams of a workgroup costs by itself, at unchanged grid,
planes and bytes moved (round 4: 15.9 -> 17-18.5 ps per node and update; profiles/r04zk_population_distance_probe.jsonl)."""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lettuce_amd._native import Plan
dev = torch.device("cuda:0")
n = 256; nodes = n ** 3
for pad_mb in (0.125, 64, 192, 448, 960):
    pad = int(pad_mb * 2 ** 20 // 4) // 64 * 64 + 64 + (32832 if pad_mb > 1 else 0)
    plan = Plan("D3Q19", torch.float32, "bgk", [n] * 3, [], device=dev)
    plan.set_population_stride(nodes + pad)
    a, b = plan.empty_populations(), plan.empty_populations()
    for q in range(19):
        a[q].fill_(0.05); a[q] += 0.001 * torch.rand([n] * 3, device=dev)
    plan.set_two_step(1, 128)
    plan.run(a, b, 0.6, 5); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(a, b, 0.6, 41); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 41)
    print(json.dumps({"grid": [n] * 3, "MB_between_populations": round((nodes + pad) * 4 / 2 ** 20, 1), "GB_per_buffer": round(19 * (nodes + pad) * 4 / 2 ** 30, 2),
                      "ms_per_update": round(best, 5), "ps_per_node_and_update": round(best * 1e9 / nodes, 2)}), flush=True)
    del a, b, plan
    torch.cuda.empty_cache()
