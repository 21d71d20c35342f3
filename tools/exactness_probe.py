"""max|df| of the HIP path against every golden case (0.0 = bit-identical to the reference CPU path)."""
import sys, os, json, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import golden, unpack_nsm, TORCH_DT
import test_gpu_engine as T
for name, lat, coll, dt, snaps in T.TGV:
    g = golden(name); plan = T.plan_for(lat, TORCH_DT[dt], coll, g["f0"].shape[1:])
    print(json.dumps({"case": name, **{f"max_df_{n}": float(np.abs(T.run_engine(plan, g["f0"], float(g["tau"]), n) - g[f"f{n}"]).max()) for n in snaps}}), flush=True)
for name, lat, coll, dt, snaps in T.OBST:
    g = golden(name); plan = T.obstacle_plan(g, lat, coll, dt)
    print(json.dumps({"case": name, **{f"max_df_{n}": float(np.abs(T.run_engine(plan, g["f0"], float(g["tau"]), n) - g[f"f{n}"]).max()) for n in snaps}}), flush=True)
