"""Slab path (self exchange) for a kernel-trace timeline.  Dev tool: python tools/slab_trace.py nx ny nz overlap"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lettuce_amd as lt
res = [int(v) for v in sys.argv[1:4]]; overlap = bool(int(sys.argv[4]))
ctx = lt.Context("cuda:0", torch.float32, True)
slab = lt.ZSlab(res, 0, 1)
flow = lt.TaylorGreenVortex(ctx, slab.extended_resolution, 1600, 0.1, lt.D3Q19(), slab=slab, initialize_fneq=False)
sim = lt.SlabSimulation(flow, lt.BGKCollision(0.53), slab, overlap=overlap)
sim(5)
torch.cuda.synchronize()
sim(30)
