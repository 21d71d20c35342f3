"""One slab shape, a few two-step launches over the whole slab: for rocprofv3 --pmc TCP_UTCL1_* passes (does the sweep's
address-translation traffic grow with the plane size?).  usage: slab_tlb_probe.py nx ny nz seg"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan, LAYOUT_SLAB
nx, ny, nz, seg = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
plan = Plan("D3Q19", torch.float32, "bgk", [nx, ny, nz], [], layout=LAYOUT_SLAB, ghost_planes=2, device=dev)
nodes = nx * ny * (nz + 4)
plan.set_population_stride(-(-(nodes + 32832) // 64) * 64)
plan.set_two_step(1, seg)
f = plan.empty_populations(); f.uniform_(0.05, 0.06)
g = plan.empty_populations(); g.zero_()
for _ in range(6):
    plan.stream_collide_twice(f, g, 0.6)
    plan.stream_collide_twice(g, f, 0.6)
torch.cuda.synchronize()
print(json.dumps({"slab": [nx, ny, nz], "seg": seg}))
