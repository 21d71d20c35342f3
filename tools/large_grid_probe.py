"""The two-step kernel on periodic D3Q19 fp32 grids larger than BASELINE's 256^3: ms per launch and ps per node by
planes per workgroup and distance between populations (dense / the resident buffers' pad / other pads).
usage: large_grid_probe.py [edge ...]   (default 256 384 512)   -> one JSON line per (grid, setting)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

edges = [int(a) for a in sys.argv[1:]] or [256, 384, 512]
dev = torch.device("cuda:0")
for n in edges:
    res = [n, n, n]
    nodes = n ** 3
    for pad in (0, 32832 + 64, 4 * 32832 + 64, 1048576 + 32832 + 64):
        plan = Plan("D3Q19", torch.float32, "bgk", res, [], device=dev)
        if pad:
            plan.set_population_stride(nodes + pad)
        a = plan.empty_populations()
        a.fill_(0.05)
        a += 0.001 * torch.rand(a.shape, device=dev)
        b = plan.empty_populations()
        for seg in (0, 32, 64, 128, 256, n):
            if seg and n % seg:
                continue
            try:
                plan.set_two_step(1, seg)
                plan.run(a, b, 0.6, 3)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                steps = 41 if n <= 384 else 21
                e0.record(); plan.run(a, b, 0.6, steps); e1.record(); torch.cuda.synchronize()
                info = plan.last_run_info()
                ms = e0.elapsed_time(e1) / steps
                print(json.dumps({"grid": res, "pad_elements": pad, "planes_per_workgroup": seg or "automatic",
                                  "ms_per_update": round(ms, 5), "ps_per_node_and_update": round(ms * 1e9 / nodes, 2),
                                  "glups": round(nodes / ms / 1e6, 1), "two_step_launches": info["two_step_launches"],
                                  "kernel": plan.kernel_name()}), flush=True)
            except Exception as exc:                       # a setting the plan refuses
                print(json.dumps({"grid": res, "pad_elements": pad, "planes_per_workgroup": seg, "refused": str(exc)[:120]}), flush=True)
        del a, b, plan
        torch.cuda.empty_cache()
