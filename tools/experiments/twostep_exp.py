"""Timing ablations of the two-step skeleton (twostep_exp.hip).  Dev tool: python tools/experiments/twostep_exp.py"""
import ctypes, json, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libtwostep_exp.so"))
lib.lt_twostep_experiment.restype = ctypes.c_int
lib.lt_twostep_experiment.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
n = 256
a = torch.rand([19, n, n, n], device="cuda") * 0.01 + 0.05
b = torch.empty_like(a)
NAMES = {0: "full", 1: "no halo-column loads", 2: "no LDS", 3: "no halo, no LDS", 4: "no barrier", 6: "no LDS, no barrier",
         7: "no halo/LDS/barrier", 32: "vmcnt(0) before loads", 64: "vmcnt(0) after barrier", 128: "vmcnt(19) after barrier", 160: "vmcnt(19) after barrier + vmcnt(0) before loads", 8: "no stores", 16: "no loads", 24: "no loads, no stores", 10: "no LDS, no stores", 18: "no LDS, no loads"}
cases = [(0, d) for d in (0, 32, 64)] + [(1, d) for d in (0, 32, 64, 128, 160)]
res = {}
st = torch.cuda.current_stream().cuda_stream
for r in range(3):
    for coll, dbg in cases:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x, y = a, b
        for it in range(8):
            if it == 2:
                e0.record()
            rc = lib.lt_twostep_experiment(coll, dbg, x.data_ptr(), y.data_ptr(), n, n, n, 0.6, 128, st)
            assert rc == 0, (coll, dbg, rc)
            x, y = y, x
        e1.record(); torch.cuda.synchronize()
        res.setdefault(f"{'bgk' if coll else 'stream'}: {NAMES[dbg]}", []).append(e0.elapsed_time(e1) / 6)
print(json.dumps({"ms_per_launch": {k: round(sorted(v)[1], 4) for k, v in res.items()}}, indent=1))
