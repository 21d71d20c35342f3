"""Time the experimental kernel variants (interleaved rounds, HIP events).  Dev tool."""
import ctypes, json, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libexperiments.so"))
lib.lt_experiment.restype = ctypes.c_int
lib.lt_experiment.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                              ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

def run(vid, a, b, n, cap, iters, tpb=256):
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        rc = lib.lt_experiment(vid, a.data_ptr(), b.data_ptr(), n[2], n[1], n[0], 0.6, cap, tpb, st); a, b = b, a
        if rc != 0: return None
    e0.record()
    for _ in range(iters):
        lib.lt_experiment(vid, a.data_ptr(), b.data_ptr(), n[2], n[1], n[0], 0.6, cap, tpb, st); a, b = b, a
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def main():
    groups = {"d3q19_f32": (19, torch.float32, [41, 46]), "d3q19_f64": (19, torch.float64, [20, 25, 26, 50, 51]),
              "d3q27_f32": (27, torch.float32, [30, 35, 36, 61, 62, 63]),
              "kbc27_f32": (27, torch.float32, [80, 81, 82, 83]), "kbc27_f64": (27, torch.float64, [84, 85, 86])}
    sel = sys.argv[1:] or list(groups)
    rounds = int(os.environ.get("ROUNDS", 3)); caps = [int(c) for c in os.environ.get("CAPS", "0").split(",")]
    tpbs = [int(c) for c in os.environ.get("TPBS", "256").split(",")]
    n = [256, 256, 256]
    for g in sel:
        q, dt, ids = groups[g]
        a = torch.full([q] + n, 1.0 / q, dtype=dt, device="cuda") * (1 + 0.01 * torch.rand([q] + n, dtype=dt, device="cuda"))
        b = torch.empty_like(a)
        # correctness cross-check against variant ids[0]
        ref = torch.empty_like(a); st = torch.cuda.current_stream().cuda_stream
        lib.lt_experiment(ids[0], a.data_ptr(), ref.data_ptr(), n[2], n[1], n[0], 0.6, 0, 256, st)
        res = {}
        for r in range(rounds):
            for vid in ids:
                for cap in caps:
                    for tpb in tpbs:
                        ms = run(vid, a, b, n, cap, 20, tpb)
                        if ms is not None: res.setdefault((vid, cap, tpb), []).append(ms)
        es = 4 if dt == torch.float32 else 8
        lib.lt_experiment(ids[0], a.data_ptr(), ref.data_ptr(), n[2], n[1], n[0], 0.6, 0, 256, st)
        for (vid, cap, tpb), v in res.items():
            v = sorted(v); ms = v[len(v) // 2]
            out = torch.empty_like(a)
            lib.lt_experiment(vid, a.data_ptr(), out.data_ptr(), n[2], n[1], n[0], 0.6, cap, tpb, st)
            err = float((out - ref).abs().max())
            print(json.dumps({"group": g, "id": vid, "cap": cap, "tpb": tpb, "ms": round(ms, 4), "min": round(v[0], 4),
                              "GBps": round(2 * q * es * 256 ** 3 / ms / 1e6, 1), "maxdiff_vs_first": err}), flush=True)
        del a, b, ref

main()
