"""pop_copy.hip: which property of the sweep costs the bandwidth the one-step kernel has?  TB/s (read + written)."""
import ctypes, json, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libpop_copy.so"))
lib.lt_pop_copy.restype = ctypes.c_int
lib.lt_pop_copy.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n = 256
N = n ** 3
a = torch.rand([19 * N], device="cuda"); b = torch.empty_like(a)
st = torch.cuda.current_stream().cuda_stream
res = {}
cases = []
for rows in (1, 2, 4):
    for ntl in (0, 1):
        for seg, pre in ((1, 0), (4, 0), (4, 1), (32, 1), (128, 1)):
            for lds in (0, 50 * 1024, 76 * 1024, 150 * 1024):
                v = rows * 100 + ntl * 10 + pre
                if v in (100, 110, 101, 111, 200, 210, 201, 211, 400, 410, 411):
                    cases.append((v, rows, ntl, 0, pre, seg, lds))
for seg in (32, 128):
    for lds in (76 * 1024, 150 * 1024):
        cases.append((213, 2, 1, 1, 1, seg, lds)); cases.append((413, 4, 1, 1, 1, seg, lds)); cases.append((403, 4, 0, 1, 1, seg, lds))
for r in range(3):
    for v, rows, ntl, bar, pre, seg, lds in cases:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(6):
            if it == 2:
                e0.record()
            rc = lib.lt_pop_copy(v, a.data_ptr(), b.data_ptr(), n, n, seg, lds, st)
            assert rc == 0, (v, rc)
        e1.record(); torch.cuda.synchronize()
        key = f"{256 * rows} threads, nt loads {ntl}, barrier {bar}, prefetch {pre}, {seg} planes/wg, lds {lds // 1024}K"
        res.setdefault(key, []).append(2 * 19 * N * 4 / 1e9 / (e0.elapsed_time(e1) / 4))
out = {k: round(sorted(v)[1], 3) for k, v in res.items()}
print(json.dumps({"TBps": dict(sorted(out.items(), key=lambda kv: -kv[1]))}, indent=1))
