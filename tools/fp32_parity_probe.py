"""How close is the fp32 HIP path to the reference's fp32 path?  max|df| and KE deviation vs golden."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import golden
from lettuce_amd._native import Plan
from oracle import lettuce_oracle as orc
for name, lat, coll, res in (("tgv3d_d3q19_bgk_16_f32", "D3Q19", "bgk", [16] * 3), ("tgv2d_d2q9_bgk_32_f32", "D2Q9", "bgk", [32] * 2),
                             ("tgv3d_d3q27_kbc_16_f32", "D3Q27", "kbc", [16] * 3), ("tgv3d_d3q19_bgk_32_f32", "D3Q19", "bgk", [32] * 3),
                             ("shear3d_d3q19_bgk_f32", "D3Q19", "bgk", [16] * 3)):
    g = golden(name)
    plan = Plan(lat, torch.float32, coll, res)
    units = orc.tgv_units(res, float(g.get("reynolds", 100)), float(g.get("mach", 0.1)))
    scale = units.incompressible_energy_to_pu(1.0) * units.length_to_pu(1.0) ** len(res)
    out = {}
    for n in (5, 10, 20, 100):
        if f"f{n}" not in g: continue
        a = torch.tensor(g["f0"], device="cuda"); b = torch.empty_like(a)
        r, _ = plan.run(a, b, float(g["tau"]), n)
        out[f"max_df_{n}"] = float(np.abs(r.cpu().numpy() - g[f"f{n}"]).max())
        ke = float(plan.kinetic_energy_lu(r).cpu()) * scale
        ref = dict(zip(g["energy_steps"].tolist(), g["energy_pu"].tolist())).get(n)
        if ref and "reynolds" in g: out[f"ke_rel_{n}"] = (ke - ref) / ref
    print(json.dumps({"case": name, **out}), flush=True)
