"""How does the address offset between the input and the output buffer affect the fused kernels?
Both live in one allocation; the output starts `pad` bytes after the end of the input."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

res = [256] * 3
plan = Plan("D3Q19", torch.float32, "bgk", res, [], device=torch.device("cuda:0"))
numel = 19 * 256 ** 3
slack = (256 << 20) // 4
pool = torch.empty(2 * numel + slack, device="cuda")
pool[:numel] = torch.rand(numel, device="cuda") * 0.01 + 0.05
a = pool[:numel].view(plan.f_shape)
pads_kb = [0, 4, 64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 1024 + 256, 2048 + 64, 3 * 1024, 5 * 1024, 33 * 1024]
out = {}
for rnd in range(3):
    for kb in pads_kb:
        off = numel + kb * 256
        b = pool[off:off + numel].view(plan.f_shape)
        for label, fn in (("single", lambda: (plan.stream_collide(a, b, 0.6), plan.stream_collide(b, a, 0.6))),
                          ("twice", lambda: (plan.stream_collide_twice(a, b, 0.6), plan.stream_collide_twice(b, a, 0.6)))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            for _ in range(4):
                fn()
            e1.record(); torch.cuda.synchronize()
            out.setdefault((label, kb), []).append(e0.elapsed_time(e1) / (8 if label == "single" else 16))
print(json.dumps({"base_address_mod_2MiB": a.data_ptr() % (2 << 20),
                  "ms_per_step": {f"{l}_pad{kb}KiB": round(sorted(v)[1], 4) for (l, kb), v in out.items()}}))
