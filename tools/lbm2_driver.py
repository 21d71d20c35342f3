"""A few two-step launches at 256^3 (D3Q19 BGK fp32) for rocprofv3 passes: python3 tools/lbm2_driver.py [policy] [launches] [seg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

policy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 12
seg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
plan = Plan("D3Q19", torch.float32, "bgk", [256] * 3, [], device=torch.device("cuda:0"))
plan.set_two_step(1, seg)
plan.set_shift_policy(policy)
f = torch.rand(plan.f_shape, device="cuda") * 0.01 + 0.05
g = torch.empty_like(f)
for _ in range(launches // 2):
    plan.stream_collide_twice(f, g, 0.6)
    plan.stream_collide_twice(g, f, 0.6)
torch.cuda.synchronize()
print("done", plan.kernel_name())
