"""pop_copy_depth.hip: sweep copy with 1, 2 or 4 planes of loads in flight per wave.  TB/s (read + written)."""
import ctypes, json, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libpop_copy_depth.so"))
lib.lt_pop_copy_depth.restype = ctypes.c_int
lib.lt_pop_copy_depth.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_void_p]
n = 256
N = n ** 3
a = torch.rand([19 * N], device="cuda"); b = torch.empty_like(a)
st = torch.cuda.current_stream().cuda_stream
res = {}
for r in range(3):
    for depth in (1, 2, 4):
        for bar in (0, 1):
            for seg in (128, 32, 4):
                for lds in (150 * 1024, 76 * 1024):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for it in range(6):
                        if it == 2:
                            e0.record()
                        rc = lib.lt_pop_copy_depth(depth, bar, a.data_ptr(), b.data_ptr(), n, n, seg, lds, st)
                        assert rc == 0
                    e1.record(); torch.cuda.synchronize()
                    res.setdefault(f"{depth} planes in flight, barrier {bar}, {seg} planes/wg, lds {lds // 1024}K", []).append(
                        2 * 19 * N * 4 / 1e9 / (e0.elapsed_time(e1) / 4))
out = {k: round(sorted(v)[1], 3) for k, v in res.items()}
print(json.dumps({"TBps": dict(sorted(out.items(), key=lambda kv: -kv[1]))}, indent=1))
