"""Microbenchmark of the two-step kernel's access pattern (tile_copy.hip): TB/s of read + written bytes."""
import ctypes, json, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libtile_copy.so"))
lib.lt_tile_copy.restype = ctypes.c_int
lib.lt_tile_copy.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                             ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong]
n = 256
N = n ** 3
PADS = [0, 64, 4096 + 64, 65536 + 4096 + 64, 1 << 20]
amax = torch.rand([19 * (N + max(PADS))], device="cuda")
bmax = torch.empty_like(amax)
st = torch.cuda.current_stream().cuda_stream
res = {}
cases = []
for variant, name in ((32, "64x8 copy rows+barrier"), (33, "64x8 copy tiled+barrier"), (12, "64x8 read rows+barrier"), (22, "64x8 write rows+barrier")):
    for skew in (0, 1, 4, 16, -1, -2):
        cases.append((variant, name, skew, 0))
    for pad in PADS[1:]:
        cases.append((variant, name, 0, pad))
        cases.append((variant, name, 16, pad))
for r in range(3):
    for variant, name, skew, pad in cases:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(6):
            if it == 2:
                e0.record()
            rc = lib.lt_tile_copy(variant, amax.data_ptr(), bmax.data_ptr(), n, n, n, 128, 150 * 1024, st, skew, N + pad)
            assert rc == 0, (variant, rc)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        gb = 19 * N * 4 * (2 if "copy" in name else 1) / 1e9
        res.setdefault(f"{name} skew{skew} pad{pad}", []).append(gb / ms)
print(json.dumps({"TBps": {k: round(sorted(v)[1], 3) for k, v in res.items()}}, indent=1))
