"""How evenly do the workgroups of a two-step launch finish?  (twostep_exp.hip, DBG 256: s_memrealtime stamps)"""
import ctypes, json, os
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libtwostep_exp.so"))
lib.lt_twostep_stamps.restype = ctypes.c_int
lib.lt_twostep_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
n = 256
a = torch.rand([19, n, n, n], device="cuda") * 0.01 + 0.05
b = torch.empty_like(a)
st = torch.cuda.current_stream().cuda_stream
for seg in (128, 64, 32):
    grid = (n // 64) * (n // 8) * (n // seg)
    stamps = torch.zeros(3 * grid, dtype=torch.int64, device="cuda")
    for it in range(4):
        rc = lib.lt_twostep_stamps(a.data_ptr(), b.data_ptr(), n, n, n, 0.6, seg, stamps.data_ptr(), st)
        assert rc == 0
        a, b = b, a
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(grid, 3)
    t0 = s[:, 0].min()
    start, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0          # microseconds
    dur = end - start
    per_xcc = {int(x): round(float(dur[s[:, 2] == x].mean()), 1) for x in sorted(set(s[:, 2].tolist()))}
    print(json.dumps({"seg": seg, "workgroups": grid, "launch_us": round(float(end.max()), 1),
                      "start_spread_us": round(float(start.max()), 1),
                      "duration_us": {"min": round(float(dur.min()), 1), "p10": round(float(np.percentile(dur, 10)), 1),
                                      "median": round(float(np.median(dur)), 1), "p90": round(float(np.percentile(dur, 90)), 1),
                                      "max": round(float(dur.max()), 1)},
                      "end_us": {"p10": round(float(np.percentile(end, 10)), 1), "median": round(float(np.median(end)), 1),
                                 "max": round(float(end.max()), 1)},
                      "busy_fraction": round(float(dur.sum() / (min(grid, 256) * end.max())), 4),
                      "mean_duration_by_xcc": per_xcc}), flush=True)
