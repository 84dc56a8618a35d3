"""Summarise the rocprofv3 output of scripts/profile_round.sh into profiles/<tag>/ (per-launch
counter values of the k_step kernel, derived ratios, kernel stats) and refresh
profiles/traffic_k_step.json.   Usage: python scripts/pmc_summary.py gpurun_out/r1b profiles/r1b"""
import csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def find(pat):
    g = sorted(glob.glob(os.path.join(src, pat), recursive=True))
    return g[0] if g else None


vals, batch, kname = {}, None, None
for name in ("sq1", "sq2", "sq3", "fetch", "write"):
    f = find(f"pmc_{name}/**/*counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, os.path.join(dst, f"rocprofv3_pmc_{name}_counter_collection.csv"))
    acc, launches = {}, set()
    for row in csv.DictReader(open(f)):
        if "k_step" not in row["Kernel_Name"]:
            continue
        kname = row["Kernel_Name"]
        batch = int(row["Grid_Size"]) // int(row["Workgroup_Size"])
        launches.add(row["Dispatch_Id"])
        acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for k, v in acc.items():
        vals[k] = v / max(1, len(launches))
for what in ("kernel_stats", "kernel_trace"):
    f = find(f"stats/**/*{what}.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"rocprofv3_{what}_bench_steps5.csv"))
for f in ("bench_line.json", "pytest_gpu_tail.log"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))

d = {}
if vals:
    rd = vals.get("FETCH_SIZE", 0.0) * 1024.0
    wr = vals.get("WRITE_SIZE", 0.0) * 1024.0
    d = {
        "hbm_read_bytes_raw": rd, "hbm_write_bytes": wr,
        "hbm_traffic_bytes_per_launch": rd + wr,
        "hbm_traffic_bytes_per_solve": (rd + wr) / batch,
        "algorithmic_bytes_per_solve": 4192,
        "mfma_cycles_per_instruction": vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, vals.get("SQ_INSTS_MFMA", 0)),
        "mfma_busy_fraction_of_wave_cycles": vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4.0 / max(1.0, vals.get("SQ_WAVE_CYCLES", 1)),
        "valu_instructions_per_solve": vals.get("SQ_INSTS_VALU", 0) / batch,
        "lds_instructions_per_solve": vals.get("SQ_INSTS_LDS", 0) / batch,
        "salu_instructions_per_solve": vals.get("SQ_INSTS_SALU", 0) / batch,
        "fma_f64_instructions_per_solve": vals.get("SQ_INSTS_VALU_FMA_F64", 0) / batch,
        "mfma_instructions_per_solve": vals.get("SQ_INSTS_MFMA", 0) / batch,
        "wave_issue_fraction": vals.get("SQ_ACTIVE_INST_ANY", 0) / max(1.0, vals.get("SQ_WAVE_CYCLES", 1)),
        "wave_wait_fraction": vals.get("SQ_WAIT_ANY", 0) / max(1.0, vals.get("SQ_WAVE_CYCLES", 1)),
        "valu_active_fraction_of_wave_cycles": vals.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, vals.get("SQ_WAVE_CYCLES", 1)),
        "lds_bank_conflict_fraction": vals.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, vals.get("SQ_LDS_IDX_ACTIVE", 1)),
        "lds_unaligned_stall": vals.get("SQ_LDS_UNALIGNED_STALL", 0),
    }
    json.dump({"kernel": kname, "batch": batch, "per_launch": vals, "derived": d},
              open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    if rd + wr > 0:
        json.dump({"config": "C3", "batch": batch, "kernel": "k_step", "hbm_bytes_per_launch": rd + wr,
                   "source": f"{dst}/rocprofv3_pmc_fetch/write_counter_collection.csv (FETCH_SIZE, WRITE_SIZE in KiB, "
                             "separate --pmc passes; raw, 8-byte-per-lane reads: the guide's x2 correction for "
                             "16-B/lane streams is not applied)"},
                  open(os.path.join(os.path.dirname(dst.rstrip('/')), "traffic_k_step.json"), "w"), indent=1)
print(json.dumps(d, indent=1))
