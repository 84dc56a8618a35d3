"""Summarise scripts/profile_mhe.sh output: per-launch counters of k_mhe_step (full-window launches only)."""
import csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def find(pat):
    g = sorted(glob.glob(os.path.join(src, pat), recursive=True))
    return g[0] if g else None


vals, kname, nlaunch = {}, None, 0
for name in ("sq1", "sq2", "fetch", "write"):
    f = find(f"pmc_{name}/**/*counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, os.path.join(dst, f"rocprofv3_pmc_{name}_counter_collection.csv"))
    per = {}
    for row in csv.DictReader(open(f)):
        if "k_mhe_step" not in row["Kernel_Name"]:
            continue
        kname = row["Kernel_Name"]
        per.setdefault(int(row["Dispatch_Id"]), {})
        per[int(row["Dispatch_Id"])][row["Counter_Name"]] = per[int(row["Dispatch_Id"])].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    ids = sorted(per)[-2:]             # the last launches: full, moving window
    nlaunch = len(ids)
    for i in ids:
        for k, v in per[i].items():
            vals[k] = vals.get(k, 0.0) + v / len(ids)
for what in ("kernel_stats", "kernel_trace"):
    f = find(f"stats/**/*{what}.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"rocprofv3_{what}_bench_C5.csv"))
if os.path.exists(os.path.join(src, "bench_line.json")):
    shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, "bench_line_C5.json"))
B = 65536
rd, wr = vals.get("FETCH_SIZE", 0.0) * 1024.0, vals.get("WRITE_SIZE", 0.0) * 1024.0
wc = max(1.0, vals.get("SQ_WAVE_CYCLES", 1.0))
d = {"kernel": kname, "launches_averaged": nlaunch, "hbm_read_bytes": rd, "hbm_write_bytes": wr,
     "hbm_bytes_per_solve": (rd + wr) / B,
     "valu_instructions_per_solve": vals.get("SQ_INSTS_VALU", 0) / B,
     "fma_f64_per_solve": vals.get("SQ_INSTS_VALU_FMA_F64", 0) / B,
     "vmem_rd_per_solve": vals.get("SQ_INSTS_VMEM_RD", 0) / B, "vmem_wr_per_solve": vals.get("SQ_INSTS_VMEM_WR", 0) / B,
     "lds_instructions_per_solve": vals.get("SQ_INSTS_LDS", 0) / B,
     "wave_issue_fraction": vals.get("SQ_ACTIVE_INST_ANY", 0) / wc, "wave_wait_fraction": vals.get("SQ_WAIT_ANY", 0) / wc,
     "valu_active_fraction_of_wave_cycles": vals.get("SQ_ACTIVE_INST_VALU", 0) / wc}
json.dump({"per_launch": vals, "derived": d}, open(os.path.join(dst, "pmc_summary_mhe.json"), "w"), indent=1)
print(json.dumps(d, indent=1))
