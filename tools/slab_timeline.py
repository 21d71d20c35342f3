"""Per-candidate kernel timeline of a traced slab rehearsal: median duration of the edge launch, the sweep, the gaps
between them and what ran beside the sweep, for every stretch of double steps with the same set of kernels.
usage: python tools/slab_timeline.py <dir with *kernel_trace.csv> [out.json]"""
import csv, glob, json, statistics, sys

def main(d, out=None):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    edge = lambda n: "lbm2_kernel" in n and ", 1, 1>" in n
    sweep = lambda n: "lbm2_kernel" in n and ", 0, 1>" in n
    periods = []
    idx = [i for i, k in enumerate(ks) if edge(k[2])]
    for a, b in zip(idx, idx[1:]):
        seg = ks[a:b]
        sw = [k for k in seg if sweep(k[2])]
        if len(sw) != 1:
            continue
        others = [k for k in seg[1:] if not sweep(k[2])]
        beside = sorted({k[2].split("(")[0][:48] for k in others})
        periods.append({"edge": seg[0][1] - seg[0][0], "gap1": sw[0][0] - seg[0][1], "sweep": sw[0][1] - sw[0][0],
                        "gap2": ks[b][0] - sw[0][1], "period": ks[b][0] - seg[0][0], "beside": tuple(beside),
                        "beside_busy": sum(k[1] - k[0] for k in others)})
    groups = {}
    for p in periods:
        groups.setdefault(p["beside"], []).append(p)
    res = []
    for beside, ps in groups.items():
        if len(ps) < 20:
            continue
        ps = ps[len(ps) // 4:]                       # steady state
        med = lambda key: round(statistics.median(p[key] for p in ps) / 1e3, 1)
        res.append({"beside_the_sweep": list(beside), "double_steps": len(ps), "edge_us": med("edge"), "gap_edge_to_sweep_us": med("gap1"),
                    "sweep_us": med("sweep"), "gap_sweep_to_next_edge_us": med("gap2"), "period_us": med("period"),
                    "other_kernels_busy_us": med("beside_busy")})
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)

if __name__ == "__main__":
    main(*sys.argv[1:3])
