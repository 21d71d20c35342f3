"""Launch time of the two-step kernel (cfg2, padded populations) against the ALIGNMENT of the two population buffers:
both carved out of one large allocation at chosen offsets from a 1 GiB boundary.  One plan, one process, launches
alternating between the placements."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan
dev = torch.device("cuda:0")
MiB = 1 << 20
plan = Plan("D3Q19", torch.float32, "bgk", [256, 256, 256], [], device=dev)
plan.set_two_step(1, 0)
stride = -(-(256 ** 3 + 32832) // 64) * 64
plan.set_population_stride(stride)
need = plan.q * stride * 4
arena = torch.empty(8 * 1024 * MiB, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
first = (-base) % (1024 * MiB)                      # offset of the first 1 GiB boundary inside the arena
inner = torch.empty(plan.f_shape[1:], device="meta").stride()


def carve(offset_bytes):
    flat = arena[offset_bytes:offset_bytes + need].view(torch.float32)
    return flat.as_strided(plan.f_shape, (stride,) + tuple(inner))


# (offset of f from the GiB boundary, offset of g from the next region's GiB-aligned start), MiB
cases = [(0, 0), (2, 2), (12, 58), (0, 58), (58, 0), (64, 64), (128, 128), (256, 256), (32, 32), (16, 16), (8, 8), (4, 4), (1, 1), (0.25, 0.25)]
bufs = []
for fo, go in cases:
    f = carve(first + int(fo * MiB)); g = carve(first + 2048 * MiB + int(go * MiB))
    bufs.append((f, g))
f0, g0 = bufs[0]
f0.uniform_(0.04, 0.06)
times = [[] for _ in cases]
for rep in range(5):
    for k, (f, g) in enumerate(bufs):
        f.copy_(f0) if k else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        plan.stream_collide_twice(f, g, 0.6)
        e0.record()
        for _ in range(10):
            plan.stream_collide_twice(f, g, 0.6)
            plan.stream_collide_twice(g, f, 0.6)
        e1.record(); torch.cuda.synchronize()
        times[k].append(round(e0.elapsed_time(e1) / 20, 4))
print(json.dumps({"arena_base": hex(base), "first_GiB_boundary_at_MiB": first / MiB}))
for (fo, go), t in zip(cases, times):
    print(json.dumps({"f_offset_MiB": fo, "g_offset_MiB": go, "ms_per_launch": t[1:], "median": sorted(t[1:])[len(t[1:]) // 2]}), flush=True)
