"""Distance between populations (engine-owned padded buffers, lt_resident_*) against launch time (round 3).

For the cfg2 grid (256^3 D3Q19 fp32) and a few others: ms per launch of the two-step and of the one-step kernel
on resident buffers with `pad` elements between consecutive populations; pad 0 = the dense layout of the
reference's tensor.  Every setting is first checked bit for bit against the dense path (5 steps).
usage: pad_sweep_probe.py [quick]
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan

dev = torch.device("cuda:0")


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, reps):
    best = 1e9
    for _ in range(3):
        e0, e1 = ev(), ev()
        fn(2)
        e0.record()
        fn(reps)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def sweep(stencil, dtype, coll, res, pads, two_step):
    plan = Plan(stencil, dtype, coll, res, [], device=dev)
    torch.manual_seed(1)
    q = plan.q
    w = torch.rand(q, *([1] * len(res)), device=dev, dtype=dtype) * 0.05 + 0.02
    f = (w * (1 + 0.1 * torch.rand(plan.f_shape, device=dev, dtype=dtype))).contiguous()
    del w
    # the dense reference: 1 collide + 4 fused + stream through lt_run
    plan.set_resident(0)
    plan.set_two_step(1 if two_step else 0, 0)
    a, b = f.clone(), torch.empty_like(f)
    ref, _ = plan.run(a, b, 0.6, 5)
    ref = ref.clone()
    del a, b
    out = torch.empty_like(f)
    for pad in pads:
        plan.set_resident(1, pad)
        plan.resident_load(f, 0.6)
        plan.resident_advance(0.6, 4)
        plan.resident_store(out)
        same = bool(torch.equal(out, ref))
        ms = timed(lambda n: plan.resident_advance(0.6, n), 20 if two_step else 10)
        info = plan.last_run_info()
        launches = info["two_step_launches"] + info["single_step_launches"]
        print(json.dumps({"stencil": stencil, "dtype": str(dtype).split(".")[1], "collision": coll, "res": res,
                          "kernel": "two-step" if info["two_step_launches"] else "one-step",
                          "pad_elements": pad, "stride_elements": plan.resident_enabled()[1],
                          "bit_identical_to_dense": same,
                          "ms_per_step": round(ms, 4),
                          "ms_per_launch": round(ms * (20 if two_step else 10) / max(1, launches), 4)}), flush=True)
    plan.resident_free()
    del plan, f, out, ref
    torch.cuda.empty_cache()


if len(sys.argv) > 1 and sys.argv[1] == "robust":
    # the candidates of the first sweep on several grids, three times over (boxes and runs differ by a per cent)
    cand = [0, 128, 320, 2112, 2368, 32832]
    for rep in range(3):
        sweep("D3Q19", torch.float32, "bgk", [256, 256, 256], cand, True)
        sweep("D3Q19", torch.float32, "bgk", [512, 512, 64], cand, True)
        sweep("D3Q19", torch.float32, "bgk", [256, 256, 512], cand, True)
        sweep("D3Q19", torch.float32, "bgk", [384, 384, 384], cand, True)
        sweep("D3Q19", torch.float64, "bgk", [256, 256, 256], cand, True)
        sweep("D3Q15", torch.float32, "bgk", [256, 256, 256], cand, True)
        sweep("D2Q9", torch.float32, "bgk", [4096, 4096], cand, True)
    sys.exit(0)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
fine = [0, 64, 128, 192, 256, 320, 448, 512, 576, 1024, 1088, 2048, 2112, 2368, 4096, 4160, 8256, 16448, 32832, 65600,
        131136, 262208, 262144, 524288, 524352, 1048576 + 64, 2 * 1048576 + 64]
if quick:
    fine = [0, 64, 576, 2368, 4160, 65600, 262208]
sweep("D3Q19", torch.float32, "bgk", [256, 256, 256], fine, True)
sweep("D3Q19", torch.float32, "bgk", [256, 256, 256], fine if not quick else [0, 2368], False)
if not quick:
    short = [0, 64, 576, 2368, 4160, 65600, 262208]
    sweep("D3Q19", torch.float64, "bgk", [256, 256, 256], short, True)
    sweep("D3Q27", torch.float32, "bgk", [256, 256, 256], short, True)
    sweep("D3Q27", torch.float32, "kbc", [256, 256, 256], short, False)
    sweep("D3Q19", torch.float32, "bgk", [512, 512, 64], short, True)
