"""fp64 HIP path vs the reference's fp64 vectors: max|df| (0.0 = bit-identical)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import golden
from lettuce_amd._native import Plan
for name, lat, coll in (("tgv3d_d3q19_bgk_16_f64", "D3Q19", "bgk"), ("tgv2d_d2q9_bgk_32_f64", "D2Q9", "bgk"), ("tgv2d_d2q9_bgk_128_f64", "D2Q9", "bgk"),
                        ("tgv3d_d3q27_bgk_16_f64", "D3Q27", "bgk"), ("tgv3d_d3q19_bgk_ragged_f64", "D3Q19", "bgk"), ("shear3d_d3q19_bgk_f64", "D3Q19", "bgk"),
                        ("tgv3d_d3q27_kbc_16_f64", "D3Q27", "kbc")):
    g = golden(name)
    plan = Plan(lat, torch.float64, coll, list(g["f0"].shape[1:]))
    out = {}
    for key in sorted(k for k in g if k.startswith("f") and k[1:].isdigit() and k != "f0"):
        n = int(key[1:])
        a = torch.tensor(g["f0"], device="cuda"); b = torch.empty_like(a)
        r, _ = plan.run(a, b, float(g["tau"]), n)
        out[f"max_df_{n}"] = float(np.abs(r.cpu().numpy() - g[key]).max())
    print(json.dumps({"case": name, **out}), flush=True)
