"""Summarise the rocprofv3 runs of tools/gpu_round.sh into profiles/:
  <tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (LBM kernels + top rows)
  traffic.json                HBM bytes per launch of the LBM kernels from the two --pmc passes

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, WRITE_SIZE is exact for
streaming stores.  The factor is calibrated in-run on the collide-only kernel
(lbm_kernel<..., false, true, ...>), which reads every population exactly once with the same
access width as the fused kernel: factor = algorithmic read bytes / (FETCH_SIZE * 1024).
The two-step kernel (lbm2_kernel) loads with the same 4 bytes per lane; its FETCH_SIZE (L2 misses,
Infinity-Cache hits included) is scaled by the same factor.  Its algorithmic bytes are those of the
two lattice updates per node it performs per launch.
"""
import csv, glob, json, os, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def is_lbm(name):
    return "lt::lbm_kernel" in name or "lt::lbm2_kernel" in name


def pmc(path):
    out = {}
    for row in csv.DictReader(open(path)):
        if is_lbm(row["Kernel_Name"]):
            out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return {k: statistics.median(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def main(tag, nodes=256 ** 3, q=19, esize=4, workload="tgv3d_d3q19_bgk_f32_256"):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.reader(open(stats)))
    keep = [rows[0]] + [r for r in rows[1:] if is_lbm(r[0])] + \
           [r for r in rows[1:8] if not is_lbm(r[0])]
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        csv.writer(fh, quoting=csv.QUOTE_ALL).writerows(keep)
    fetch, n_f = pmc(glob.glob(os.path.join(src, "pmc_FETCH_SIZE", "*", "*_counter_collection.csv"))[0])
    write, n_w = pmc(glob.glob(os.path.join(src, "pmc_WRITE_SIZE", "*", "*_counter_collection.csv"))[0])
    alg = q * esize * nodes
    calib = [k for k in fetch if ", false, true, " in k]
    factor = alg / (fetch[calib[0]] * 1024) if calib else 2.0
    kernels = []
    for k in sorted(fetch):
        rd = fetch[k] * 1024 * factor
        wr = write.get(k, float("nan")) * 1024
        updates = 2 if "lbm2_kernel" in k else 1     # the two-step kernel does two lattice updates per node
        kernels.append({"kernel": k, "workload": workload, "launches_sampled": n_f[k],
                        "FETCH_SIZE_KiB_median": fetch[k], "WRITE_SIZE_KiB_median": write.get(k),
                        "fetch_correction_factor": round(factor, 4),
                        "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr),
                        "hbm_bytes_per_launch": round(rd + wr),
                        "lattice_updates_per_node_per_launch": updates,
                        "algorithmic_bytes_per_launch": updates * 2 * alg,
                        "traffic_over_algorithmic": round((rd + wr) / (updates * 2 * alg), 4)})
    sys.path.insert(0, ROOT)
    from bench import source_hash           # bench.py uses the table only for a build of these sources
    json.dump({"tag": tag, "source_hash": source_hash(),
               "note": __doc__.split("gfx950 corrections")[1].strip(), "kernels": kernels},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    for k in kernels:
        print(k["kernel"][:70], k["hbm_bytes_per_launch"], k["traffic_over_algorithmic"])
    # optional extra passes (SQ issue mix, TCC hit/miss): medians per LBM kernel
    extra = {}
    for sub in ("pmc_SQ", "pmc_TCC"):
        files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
        if not files:
            continue
        acc = {}
        for row in csv.DictReader(open(files[0])):
            if is_lbm(row["Kernel_Name"]):
                acc.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(
                    float(row["Counter_Value"]))
        for kname, counters in acc.items():
            extra.setdefault(kname, {}).update({c: statistics.median(v) for c, v in counters.items()})
    if extra:
        for kname, c in extra.items():
            if "SQ_WAVES" in c and c["SQ_WAVES"]:
                c["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
                c["vmem_rd_per_wave"] = round(c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"], 2)
                c["vmem_wr_per_wave"] = round(c["SQ_INSTS_VMEM_WR"] / c["SQ_WAVES"], 2)
            if "TCC_HIT_sum" in c:
                c["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        json.dump({"tag": tag, "note": "medians over the sampled launches; rocprofv3 --pmc, separate "
                   "passes for SQ and TCC (tools/gpu_round.sh)", "kernels": extra},
                  open(os.path.join(dst, f"{tag}_pmc_sq_tcc.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
