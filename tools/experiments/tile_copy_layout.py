"""Layouts for the two-step sweep (tile_copy.hip): rows where they lie / tile-major per plane / tile-column-major."""
import ctypes, json, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libtile_copy.so"))
lib.lt_tile_copy.restype = ctypes.c_int
lib.lt_tile_copy.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                             ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong]
n = 256
N = n ** 3
a = torch.rand([19 * N], device="cuda"); b = torch.empty_like(a)
st = torch.cuda.current_stream().cuda_stream
res = {}
cases = [(shape, mode, fl) for shape in (0, 1, 2) for mode in (3, 1, 2) for fl in (2, 3, 4)]
names = {0: "64x8", 1: "128x4", 2: "256x2"}
for r in range(3):
    for shape, mode, fl in cases:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(6):
            if it == 2:
                e0.record()
            rc = lib.lt_tile_copy(shape * 100 + mode * 10 + fl, a.data_ptr(), b.data_ptr(), n, n, n, 128, 150 * 1024, st, 0, N)
            assert rc == 0, (shape, mode, fl, rc)
        e1.record(); torch.cuda.synchronize()
        gb = 19 * N * 4 * (2 if mode == 3 else 1) / 1e9
        key = " ".join([names[shape], {3: "copy", 1: "read", 2: "write"}[mode],
                        {2: "rows", 3: "tile-major per plane", 4: "tile-column-major"}[fl]])
        res.setdefault(key, []).append(gb / (e0.elapsed_time(e1) / 4))
print(json.dumps({"TBps": {k: round(sorted(v)[1], 3) for k, v in res.items()}}, indent=1))
