"""The two-step sweep pattern with 16-byte accesses per lane (tile_copy4.hip): TB/s."""
import ctypes, json, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libtile_copy4.so"))
lib.lt_tile_copy4.restype = ctypes.c_int
lib.lt_tile_copy4.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n = 256
a = torch.rand([19, n, n, n], device="cuda"); b = torch.empty_like(a)
st = torch.cuda.current_stream().cuda_stream
res = {}
for r in range(3):
    for mode, name in ((3, "copy"), (1, "read"), (2, "write")):
        for barrier in (0, 1):
            for lds, seg in ((150 * 1024, 128), (76 * 1024, 64), (50 * 1024, 32)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for it in range(6):
                    if it == 2:
                        e0.record()
                    rc = lib.lt_tile_copy4(mode, barrier, a.data_ptr(), b.data_ptr(), n, n, n, seg, lds, st)
                    assert rc == 0
                e1.record(); torch.cuda.synchronize()
                gb = a.numel() * 4 * (2 if mode == 3 else 1) / 1e9
                res.setdefault(f"64x8 float4 {name} barrier{barrier} lds{lds//1024}K seg{seg}", []).append(gb / (e0.elapsed_time(e1) / 4))
print(json.dumps({"TBps": {k: round(sorted(v)[1], 3) for k, v in res.items()}}, indent=1))
