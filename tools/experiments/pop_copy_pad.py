"""pop_copy.hip with padded planes: is a long-lived workgroup slow because its rows sit at a fixed offset within every plane?"""
import ctypes, json, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libpop_copy.so"))
lib.lt_pop_copy.restype = ctypes.c_int
lib.lt_pop_copy.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                            ctypes.c_int]
n = 256
PADS = [0, 64, 256, 1024, 1024 + 64, 4096 + 64, 16384 + 1024 + 64]
size = 19 * n * (n * n + max(PADS))
a = torch.rand([size], device="cuda"); b = torch.empty_like(a)
st = torch.cuda.current_stream().cuda_stream
res = {}
for r in range(3):
    for v, label in ((213, "512 threads, nt loads, barrier, prefetch"), (211, "512 threads, nt loads, prefetch"), (201, "512 threads, prefetch")):
        for seg in (128, 1):
            for pad in PADS:
                vv = v if seg > 1 else 210
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for it in range(6):
                    if it == 2:
                        e0.record()
                    rc = lib.lt_pop_copy(vv, a.data_ptr(), b.data_ptr(), n, n, seg, 150 * 1024 if seg > 1 else 76 * 1024, st, pad)
                    assert rc == 0, (vv, rc)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(f"{label if seg > 1 else '512 threads, nt loads, one plane per workgroup'}, {seg} planes/wg, plane pad {pad}", []).append(
                    2 * 19 * n ** 3 * 4 / 1e9 / (e0.elapsed_time(e1) / 4))
print(json.dumps({"TBps": {k: round(sorted(v)[1], 3) for k, v in res.items()}}, indent=1))
