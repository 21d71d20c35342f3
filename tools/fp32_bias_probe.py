"""One BGK collision in fp32: GPU kernel vs the reference's own output (golden): exact-match rate and
signed mean difference (is there a systematic bias?)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import golden
from lettuce_amd._native import Plan
g = golden("operators_d3q19_f32")
f = torch.tensor(g["f"], device="cuda")
plan = Plan("D3Q19", torch.float32, "bgk", list(g["f"].shape[1:]))
out = plan.collide(f, torch.empty_like(f), float(g["tau"])).cpu().numpy()
ref = g["bgk"]
d = out.astype(np.float64) - ref.astype(np.float64)
print(json.dumps({"exact_match": float((out == ref).mean()), "mean_signed_diff": float(d.mean()), "mean_abs": float(np.abs(d).mean()),
                  "ulp_of_f": float(np.spacing(np.float32(ref.mean())))}))
rho, u = plan.macroscopic(f)
print(json.dumps({"rho_exact": float((rho.cpu().numpy()[None] == g["rho"]).mean()), "u_exact": float((u.cpu().numpy() == g["u"]).mean()),
                  "u_mean_signed": float((u.cpu().numpy().astype(np.float64) - g["u"]).mean())}))
feq = plan.equilibrium(torch.tensor(g["rho"], device="cuda"), torch.tensor(g["u"], device="cuda")).cpu().numpy()
de = feq.astype(np.float64) - g["feq"]
print(json.dumps({"feq_exact_given_same_rho_u": float((feq == g["feq"]).mean()), "feq_mean_signed": float(de.mean()), "feq_mean_abs": float(np.abs(de).mean())}))
# momentum deficit of the equilibrium: sum_q e_q feq_q vs rho u (fp64 evaluation of fp32 data)
from oracle import lettuce_oracle as orc
lat = orc.LATTICES["D3Q19"]
e = np.array(lat.e, dtype=np.float64)
rho64, u64 = g["rho"].astype(np.float64), g["u"].astype(np.float64)
for tag, fe in (("reference", g["feq"]), ("gpu", feq)):
    J = np.einsum("qd,q...->d...", e, fe.astype(np.float64))
    num = (J * u64).sum(axis=0); den = (rho64[0] * (u64 * u64).sum(axis=0))
    print(json.dumps({"feq_from": tag, "mean_momentum_ratio_minus_1": float((num / den - 1).mean()),
                      "mass_ratio_minus_1": float((fe.astype(np.float64).sum(axis=0) / rho64[0] - 1).mean())}))
# anatomy of the feq mismatches
mm = feq != g["feq"]
dq = (feq.astype(np.float64) - g["feq"])
exu = np.einsum("qd,d...->q...", e, u64)
rows = []
for q in range(19):
    m = mm[q]
    if m.sum():
        rows.append((q, int(m.sum()), float(np.sign(dq[q][m]).mean()), float(np.sign(exu[q][m]).mean()), float(np.abs(exu[q][m]).mean()), float(np.abs(exu[q]).mean())))
print(json.dumps({"mismatch_by_q[q,count,mean sign(diff),mean sign(e.u),mean|e.u| at mismatch, overall]": rows}))
