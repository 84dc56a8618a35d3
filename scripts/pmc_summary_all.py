"""Summarise the rocprofv3 output of scripts/profile_all.sh: for EVERY kernel the bench line quotes, the counters of its
last launch in each --pmc pass (the steady-state launch: the MHE kernels fill their window first), per-unit figures and
derived ratios -> <dst>/pmc_summary_<kernel>.json, the kernel-stats CSV, and <dst>/traffic.json (HBM bytes per launch per
kernel, what bench.py / bench_mhe.py put into roofline.traffic).
Usage: python scripts/pmc_summary_all.py gpurun_out/r5a profiles/r5a"""
import csv, glob, json, os, re, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)

def step_algorithmic_bytes(nx, nu, ny, Hp, Hc, neps=1):
    """SURVEY 8(d), on-device condensation: what one LinMPC step must read and write once -- model, state, last input, set
    point, weights, bounds, warm start in; Z~, u and two ints out (bench.py: algorithmic_bytes; C3: 4192 B)."""
    nxh, n = nx + ny, nu * Hc + neps
    ins = nxh * nxh + nxh * nu + ny * nxh + nxh + nu + ny + (ny + 2 * nu + 1) + (2 * nu + 2 * ny) + n
    return 8 * (ins + n + nu + 1)


# (kernel-name pattern, short name, what one launch processes, units per launch, algorithmic bytes per unit or None)
# (team kernels of round 6: k_step_team<StaticDims<..>, T>)
KERNELS = [
    (r"k_step_s<.*StaticDims<4, 4, 16, 30, 10", "k_step_s_C3", "QP solves (C3, B = 65536)", 65536, step_algorithmic_bytes(12, 4, 4, 30, 10)),
    (r"k_step_(s|team)<.*StaticDims<3, 3, 15, 40, 35", "k_step_s_nZ106", "QP solves (nu = ny = 3, Hp = 40, Hc = 35, B = 8192)", 8192, step_algorithmic_bytes(12, 3, 3, 40, 35)),
    (r"k_step_(s|team)<.*StaticDims<3, 3, 15, 50, 50", "k_step_s_nZ151", "QP solves (nu = ny = 3, Hp = Hc = 50, B = 4096)", 4096, step_algorithmic_bytes(12, 3, 3, 50, 50)),
    (r"k_step_small_w1<12", "k_step_small_w1_12", "QP solves (C2, B = 1024)", 1024, step_algorithmic_bytes(4, 2, 2, 20, 5)),
    (r"k_step_small<12", "k_step_small_12", "QP solves (C2, B = 65536)", 65536, step_algorithmic_bytes(4, 2, 2, 20, 5)),
    (r"k_step_small_y<12", "k_step_small_y_12", "QP solves (C2 dims, soft ymax + hard u, B = 65536)", 65536, step_algorithmic_bytes(4, 2, 2, 20, 5)),
    (r"k_ms_step_g", "k_ms_step_g", "QP solves (MultipleShooting, Hp = Hc = 50, B = 8192)", 8192, step_algorithmic_bytes(6, 2, 2, 50, 50)),
    (r"k_mhe_step<12, 1u>", "k_mhe_step_12_hard", "estimator periods (C5, B = 65536)", 65536, None),
    (r"k_mhe_step<12, 15u>", "k_mhe_step_12_soft", "estimator periods (C5 soft, B = 65536)", 65536, None),
]


def find(pat):
    g = sorted(glob.glob(os.path.join(src, pat), recursive=True))
    return g[0] if g else None


def key_of(name):
    for pat, short, *_ in KERNELS:
        if re.search(pat, name):
            return short
    return None


# C5: algorithmic bytes of one MHE period = what one estimator must read and write once per period: the window data
# (He + 1 measurement / input blocks), the model constants, the arrival covariance, the previous window's estimates (warm
# start) in, the new ones out -- stated in DESIGN 4b; computed here from the dimensions of bench_mhe's C5 (nx̂ = 12,
# nym = 4, nu = 4, He = 20).
def mhe_algorithmic_bytes(nx=12, nym=4, nu=4, He=20):
    model = nx * nx + nx * nu + nym * nx + nx * nx + nym * nym          # Â, B̂u, Ĉm, Q̂⁻¹, R̂⁻¹
    window = (He + 1) * (nym + nu) + 2 * (He + 1) * nx                  # Y, U windows; Ŵ / X̂ warm start in, out
    cov = 2 * nx * nx                                                   # P̄ in, out
    return 8 * (model + window + cov)


vals = {}          # short -> counter -> value of the last launch
meta = {}
for name in ("sq1", "sq2", "sq3", "fetch", "write"):
    f = find(f"pmc_{name}/**/*counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, os.path.join(dst, f"rocprofv3_pmc_{name}_counter_collection.csv"))
    per = {}       # short -> dispatch id -> counter -> value
    for row in csv.DictReader(open(f)):
        k = key_of(row["Kernel_Name"])
        if not k:
            continue
        did = int(row["Dispatch_Id"])
        per.setdefault(k, {}).setdefault(did, {})
        per[k][did][row["Counter_Name"]] = per[k][did].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        meta.setdefault(k, {"kernel_name": row["Kernel_Name"], "grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]),
                            "launches_in_pass": 0})
    for k, d in per.items():
        last = max(d)
        meta[k]["launches_in_pass"] = len(d)
        vals.setdefault(k, {}).update(d[last])
for what in ("kernel_stats", "kernel_trace"):
    f = find(f"stats/**/*{what}.csv")
    if f and (what == "kernel_stats" or os.path.getsize(f) < 4 << 20):
        shutil.copy(f, os.path.join(dst, f"rocprofv3_{what}_bench_steps5.csv"))
for f in ("bench_line.json", "pytest_gpu_tail.log"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))

traffic = {}
for pat, short, what, units, algo in KERNELS:
    v = vals.get(short)
    if not v:
        print("no counters for", short)
        continue
    if short.startswith("k_mhe_step"):
        algo = mhe_algorithmic_bytes()
    rd, wr = v.get("FETCH_SIZE", 0.0) * 1024.0, v.get("WRITE_SIZE", 0.0) * 1024.0
    wc = max(1.0, v.get("SQ_WAVE_CYCLES", 1))
    d = {
        "kernel": meta[short]["kernel_name"], "one_launch": what, "units_per_launch": units, "grid_threads": meta[short]["grid"],
        "workgroup": meta[short]["workgroup"], "launches_seen": meta[short]["launches_in_pass"], "which_launch": "the last one of each pass",
        "hbm_read_bytes_raw": rd, "hbm_write_bytes": wr, "hbm_traffic_bytes_per_launch": rd + wr,
        "hbm_traffic_bytes_per_unit": (rd + wr) / units, "algorithmic_bytes_per_unit": algo,
        "traffic_over_algorithmic": ((rd + wr) / units / algo) if algo else None,
        "valu_instructions_per_unit": v.get("SQ_INSTS_VALU", 0) / units,
        "salu_instructions_per_unit": v.get("SQ_INSTS_SALU", 0) / units,
        "lds_instructions_per_unit": v.get("SQ_INSTS_LDS", 0) / units,
        "mfma_instructions_per_unit": v.get("SQ_INSTS_MFMA", 0) / units,
        "fma_f64_per_unit": v.get("SQ_INSTS_VALU_FMA_F64", 0) / units, "mul_f64_per_unit": v.get("SQ_INSTS_VALU_MUL_F64", 0) / units,
        "add_f64_per_unit": v.get("SQ_INSTS_VALU_ADD_F64", 0) / units, "trans_f64_per_unit": v.get("SQ_INSTS_VALU_TRANS_F64", 0) / units,
        "int32_per_unit": v.get("SQ_INSTS_VALU_INT32", 0) / units, "int64_per_unit": v.get("SQ_INSTS_VALU_INT64", 0) / units,
        "vmem_rd_per_unit": v.get("SQ_INSTS_VMEM_RD", 0) / units, "vmem_wr_per_unit": v.get("SQ_INSTS_VMEM_WR", 0) / units,
        "mfma_cycles_per_instruction": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, v.get("SQ_INSTS_MFMA", 0)),
        "mfma_busy_fraction_of_wave_cycles": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4.0 / wc,
        "wave_issue_fraction": v.get("SQ_ACTIVE_INST_ANY", 0) / wc, "wave_wait_fraction": v.get("SQ_WAIT_ANY", 0) / wc,
        "wave_issue_stall_fraction": v.get("SQ_WAIT_INST_ANY", 0) / wc,
        "valu_active_fraction_of_wave_cycles": v.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        "lds_bank_conflict_fraction": v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, v.get("SQ_LDS_IDX_ACTIVE", 1)),
        "lds_unaligned_stall": v.get("SQ_LDS_UNALIGNED_STALL", 0), "raw": v,
        "note": "FETCH_SIZE / WRITE_SIZE in KiB, separate --pmc passes, raw (MI355X_MICROARCH.md: FETCH_SIZE reports half of a 16-B/lane "
                "streaming read on gfx950; these kernels read 8 B per lane, uncalibrated, so no correction is applied); SQ_WAVE_CYCLES, "
                "SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles",
    }
    json.dump(d, open(os.path.join(dst, f"pmc_summary_{short}.json"), "w"), indent=1)
    traffic[short] = {"hbm_bytes_per_launch": rd + wr, "units_per_launch": units, "algorithmic_bytes_per_unit": algo}
    print(f"{short:22s} VALU/unit {d['valu_instructions_per_unit']:9.0f}  MFMA/unit {d['mfma_instructions_per_unit']:7.0f}  LDS/unit {d['lds_instructions_per_unit']:8.0f}  "
          f"HBM B/unit {d['hbm_traffic_bytes_per_unit']:10.0f}  issue {d['wave_issue_fraction']:.2f} wait {d['wave_wait_fraction']:.2f} bank-conflict {d['lds_bank_conflict_fraction']:.2f}")
traffic["source"] = f"{os.path.basename(dst.rstrip('/'))}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) of the last launch of each kernel, separate passes, raw"
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
if os.path.basename(os.path.dirname(os.path.abspath(dst))) == "profiles":        # the copy bench.py / bench_mhe.py read
    json.dump(traffic, open(os.path.join(os.path.dirname(os.path.abspath(dst)), "traffic.json"), "w"), indent=1)
