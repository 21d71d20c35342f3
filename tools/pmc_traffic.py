"""Summarise the rocprofv3 runs of tools/gpu_round.sh (rounds 3+) into profiles/:
  <tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the judged command (python bench.py): LBM kernels +
                                   the top other rows
  <tag>_other_kernel_stats.csv     the same for the other BASELINE workloads (tools/profile_workload.py), LBM kernels only
  <tag>_slab_kernel_stats.csv      ... and for the slab rehearsal (bench.py --slab)
  traffic.json                     HBM bytes per launch of the LBM kernels of every workload from the --pmc passes
  <tag>_pmc_sq_tcc.json            SQ issue / wait counters, TCC hit / miss

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, WRITE_SIZE is exact for
streaming stores.  The factor is calibrated in-run, per workload, on the collide-only kernel
(lbm_kernel<..., false, true, ...>: the first launch of a batch), which reads every population exactly once with the
same access width as the fused kernels: factor = algorithmic read bytes / (FETCH_SIZE * 1024); a workload without such
a launch in its samples uses 2.0.  The two-step kernels (lbm2_kernel, lbm2m_kernel) load with the same 4 / 8 bytes per
lane; their FETCH_SIZE (L2 misses, Infinity-Cache hits included) is scaled by the same factor.  Their algorithmic bytes
are those of the two lattice updates per node they perform per launch.
"""
import csv, glob, json, os, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ["cfg2", "cfg4", "cfg4bgk", "obst19", "cfg5", "slab", "slab5"]


def is_lbm(name):
    return "lt::lbm" in name


def pmc(path):
    out = {}
    for row in csv.DictReader(open(path)):
        if is_lbm(row["Kernel_Name"]):
            out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return {k: statistics.median(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def first(pattern):
    """the newest match: gpurun merges every session of a tag into the same directory"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def stats_rows(path, others=0):
    rows = list(csv.reader(open(path)))
    return [rows[0]] + [r for r in rows[1:] if is_lbm(r[0])] + [r for r in rows[1:1 + others] if not is_lbm(r[0])]


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = first(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
            csv.writer(fh, quoting=csv.QUOTE_ALL).writerows(stats_rows(stats, 7))
    slab = first(os.path.join(src, "trace_slab", "*", "*_kernel_stats.csv"))
    if slab:
        with open(os.path.join(dst, f"{tag}_slab_kernel_stats.csv"), "w", newline="") as fh:
            csv.writer(fh, quoting=csv.QUOTE_ALL).writerows(stats_rows(slab, 5))
    kernels, other_rows = [], []
    for w in WORKLOADS:
        meta_file = os.path.join(src, f"w_{w}.json")
        if not os.path.exists(meta_file):
            continue
        meta = json.loads([ln for ln in open(meta_file) if ln.startswith("{")][-1])
        st = first(os.path.join(src, f"w_{w}", "trace", "*", "*_kernel_stats.csv"))
        if st:
            rows = stats_rows(st)
            if not other_rows:
                other_rows.append(["workload"] + rows[0])
            other_rows += [[meta["workload"]] + r for r in rows[1:]]
        f_file = first(os.path.join(src, f"w_{w}", "pmc_FETCH_SIZE", "*", "*_counter_collection.csv"))
        w_file = first(os.path.join(src, f"w_{w}", "pmc_WRITE_SIZE", "*", "*_counter_collection.csv"))
        if not f_file or not w_file:
            continue
        fetch, n_f = pmc(f_file)
        write, _ = pmc(w_file)
        alg = meta["q"] * meta["esize"] * meta["nodes"]          # one pass over the populations in one direction
        calib = [k for k in fetch if ", false, true, " in k]
        factor = alg / (fetch[calib[0]] * 1024) if calib else 2.0
        for k in sorted(fetch):
            rd = fetch[k] * 1024 * factor
            wr = write.get(k, float("nan")) * 1024
            updates = 2 if ("lbm2_kernel" in k or "lbm2m_kernel" in k) else 1
            kernels.append({"kernel": k, "workload": meta["workload"], "launches_sampled": n_f[k],
                            "FETCH_SIZE_KiB_median": fetch[k], "WRITE_SIZE_KiB_median": write.get(k),
                            "fetch_correction_factor": round(factor, 4), "factor_calibrated_in_run": bool(calib),
                            "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr),
                            "hbm_bytes_per_launch": round(rd + wr),
                            "lattice_updates_per_node_per_launch": updates,
                            "one_pass_bytes": 2 * alg,
                            "read_over_one_pass_read": round(rd / alg, 4),
                            "algorithmic_bytes_per_launch": updates * 2 * alg,
                            "traffic_over_algorithmic": round((rd + wr) / (updates * 2 * alg), 4)})
    if other_rows:
        with open(os.path.join(dst, f"{tag}_other_kernel_stats.csv"), "w", newline="") as fh:
            csv.writer(fh, quoting=csv.QUOTE_ALL).writerows(other_rows)
    sys.path.insert(0, ROOT)
    from bench import source_hash           # bench.py uses the table only for a build of these sources
    json.dump({"tag": tag, "source_hash": source_hash(),
               "note": __doc__.split("gfx950 corrections")[1].strip(), "kernels": kernels},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    for k in kernels:
        print(k["workload"][:28].ljust(28), k["kernel"][10:80].ljust(70), k["hbm_bytes_per_launch"], "read x", k["read_over_one_pass_read"],
              "traffic/alg", k["traffic_over_algorithmic"])
    # SQ issue mix, TCC hit/miss: medians per LBM kernel and workload
    extra = {}
    for w in WORKLOADS:
        for sub in ("pmc_SQ", "pmc_TCC"):
            f = first(os.path.join(src, f"w_{w}", sub, "*", "*_counter_collection.csv"))
            if not f:
                continue
            acc = {}
            for row in csv.DictReader(open(f)):
                if is_lbm(row["Kernel_Name"]):
                    acc.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for kname, counters in acc.items():
                extra.setdefault(f"{w}: {kname}", {}).update({c: statistics.median(v) for c, v in counters.items()})
    for kname, c in extra.items():
        if c.get("SQ_WAVES"):
            c["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
            c["vmem_rd_per_wave"] = round(c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"], 2)
            c["vmem_wr_per_wave"] = round(c["SQ_INSTS_VMEM_WR"] / c["SQ_WAVES"], 2)
            if c.get("SQ_WAVE_CYCLES"):
                c["wait_any_share_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
        if "TCC_HIT_sum" in c:
            c["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if extra:
        json.dump({"tag": tag, "note": "medians over the sampled launches; rocprofv3 --pmc, separate passes for SQ and TCC "
                   "(tools/gpu_round3.sh)", "kernels": extra},
                  open(os.path.join(dst, f"{tag}_pmc_sq_tcc.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
