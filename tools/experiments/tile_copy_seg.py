"""Does the sweep itself cost bandwidth?  The tile copy (tile_copy.hip) with short-lived workgroups: seg planes per
workgroup from 1 (one plane, then exit: the dispatch pattern of the one-step kernel) to 128 (the two-step kernel's)."""
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
for r in range(3):
    for variant, name in ((30, "64x8 copy rows"), (230, "256x2 copy rows"), (10, None), (20, None)):
        if name is None:
            continue
        for seg in (1, 2, 4, 16, 128):
            for lds in (0, 50 * 1024, 150 * 1024):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for it in range(6):
                    if it == 2:
                        e0.record()
                    rc = lib.lt_tile_copy(variant, a.data_ptr(), b.data_ptr(), n, n, n, seg, lds, st, 0, N)
                    assert rc == 0, (variant, rc)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(f"{name} seg{seg} lds{lds // 1024}K", []).append(2 * 19 * N * 4 / 1e9 / (e0.elapsed_time(e1) / 4))
print(json.dumps({"TBps": {k: round(sorted(v)[1], 3) for k, v in res.items()}}, indent=1))
