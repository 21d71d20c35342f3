"""a few launches of the three-step and of the two-step kernel at 256^3 for rocprofv3 --pmc passes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan
dev = torch.device("cuda:0")
plan = Plan("D3Q19", torch.float32, "bgk", [256, 256, 256], [], device=dev)
plan.set_two_step(1, 0)
nodes = 256 ** 3
plan.set_population_stride(-(-(nodes + 32832) // 64) * 64)
f = plan.empty_populations(); f.uniform_(0.04, 0.06)
g = plan.empty_populations(); g.zero_()
for _ in range(4):
    plan.stream_collide_thrice(f, g, 0.6); plan.stream_collide_thrice(g, f, 0.6)
    plan.stream_collide_twice(f, g, 0.6); plan.stream_collide_twice(g, f, 0.6)
torch.cuda.synchronize()
