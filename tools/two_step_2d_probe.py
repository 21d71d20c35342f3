"""2-D two-step kernel (twostep2d.hpp): ms per lattice update at 4096^2 / 8192^2 against the one-step kernel."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan


def ev():
    return torch.cuda.Event(enable_timing=True)


for dtype, esize in ((torch.float32, 4), (torch.float64, 8)):
    for n in (4096, 8192, 2048):
        plan = Plan("D2Q9", dtype, "bgk", [n, n], [], device=torch.device("cuda:0"))
        plan.set_many_step(0)
        f = torch.rand(plan.f_shape, device="cuda", dtype=dtype) * 0.01 + 0.05
        g = torch.empty_like(f)
        out = {}
        segs = [0, 16, 32, 64, 128, 256]
        for r in range(5):
            for seg in [-1] + segs:
                e0, e1 = ev(), ev()
                a, b = f, g
                if seg >= 0:
                    plan.set_two_step(1, seg)
                for it in range(12):
                    if it == 2:
                        e0.record()
                    if seg < 0:
                        plan.stream_collide(a, b, 0.6); a, b = b, a
                        plan.stream_collide(a, b, 0.6); a, b = b, a
                    else:
                        plan.stream_collide_twice(a, b, 0.6); a, b = b, a
                e1.record(); torch.cuda.synchronize()
                out.setdefault("one-step" if seg < 0 else f"two-step seg{seg}", []).append(e0.elapsed_time(e1) / 20)
        med = {k: sorted(v)[2] for k, v in out.items()}
        print(json.dumps({"grid": [n, n], "dtype": str(dtype), "kernel": plan.kernel_name(),
                          "ms_per_update": {k: round(v, 4) for k, v in med.items()},
                          "GLUPS": {k: round(n * n / v / 1e6, 1) for k, v in med.items()},
                          "hbm_frac_one_step": round(2 * 9 * esize * n * n / med["one-step"] / 1e-3 / 8e12, 3)}), flush=True)
        del plan, f, g
        torch.cuda.empty_cache()
