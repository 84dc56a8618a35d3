"""Static instruction counts per marker span of a -DMPCQP_ISA_MARKERS build (see scripts/isa_phase_table.py for the build line and the
classes).  No loop weighting: every instruction of the kernel is attributed to the innermost open marker at its position in the
listing (the text between "tic X" and "toc X"), so K loops count once.  Robust against the layout changes that break the loop
detection of isa_phase_table.py; with `-v SPAN` the opcode histogram of one span is printed.
  python scripts/isa_span_counts.py /tmp/c3_markers.s [-v span]"""
import collections, re, sys
sys.path.insert(0, "scripts")
path = sys.argv[1]
verbose = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "-v" else None
src = open(path).read().split("\n")
start = [i for i, l in enumerate(src) if re.match(r"^_ZN5mpcqp\d+k_step_s", l)][0]
end = [i for i, l in enumerate(src) if i > start and ".amdhsa_kernel" in l][0]


def kind(op, s):
    if "mfma" in op: return "mfma"
    if op.endswith("_dpp") or " row_" in s or "quad_perm" in s or "row_newbcast" in s:
        return "dpp64" if "_f64" in op else "dpp32"
    if op.startswith(("v_fma_f64", "v_fmac_f64")): return "fma64"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return "trans"
    if op.startswith(("v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_ldexp_f64", "v_frexp", "v_cvt_f64")): return "f64"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if "readlane" in op or "writelane" in op or "readfirstlane" in op: return "lane"
    if op.startswith(("v_cndmask", "v_cmp")): return "sel"
    if op.startswith(("v_mov", "v_accvgpr")): return "mov"
    if op.startswith("v_"): return "int"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_"): return "salu"
    return "other"


PH = {"0": "G v rows", "1": "G'w", "2": "H~z", "3": "GtDG rowpass", "4": "E'DE mfma", "5": "GtDG U/diag", "6": "cholesky", "7": "tri solves",
      "8": "G'w rows+red", "9": "G'w box/U/eps", "10": "G'w E'w", "11": "Gv ucum", "12": "Gv Ev", "13": "step+update", "14": "polish",
      "rowstep": "row step", "looptop": "looptop"}
NAMED = {"tic_gt_": "1"}
stack, insts = [], []          # stack of [name, first instruction index]; insts = [op, kind, span]
for l in src[start:end]:
    s = l.strip()
    m = re.match(r";\s*MPCQP_MARK\s+(tic|toc)\s*(\S*)", s)
    if m:
        if m.group(1) == "tic":
            stack.append([m.group(2), len(insts)])
        else:
            ph = m.group(2)
            want = [i for i, e in enumerate(stack) if e[0] == f"tic{ph}_" or NAMED.get(e[0]) == ph]
            if not want: want = [i for i, e in enumerate(stack) if e[0] == ""]
            if want:
                e = stack.pop(want[-1])
                for t in insts[e[1]:]:
                    if t[2] is None: t[2] = ph
        continue
    if not s or s.startswith((".", ";", "/")) or re.match(r"^[\w.$]+:", s):
        continue
    op = s.split()[0]
    insts.append([op, kind(op, s), None])
counts, ops = collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
for op, k, span in insts:
    span = PH.get(span, span) if span is not None else "(outside)"
    counts[span][k] += 1
    ops[span][op] += 1
cols = ["fma64", "f64", "trans", "dpp64", "int", "sel", "mov", "lane", "dpp32", "mfma", "lds", "vmem", "salu", "br", "wait", "nop"]
print(f"{'span':>12} " + " ".join(f"{c:>6}" for c in cols) + "   VALU")
tot = collections.Counter()
for span, c in sorted(counts.items(), key=lambda kv: -sum(kv[1].values())):
    valu = sum(c[k] for k in ("fma64", "f64", "trans", "dpp64", "int", "sel", "mov", "lane", "dpp32"))
    print(f"{span:>12} " + " ".join(f"{c[k]:6d}" for k in cols) + f" {valu:6d}")
    tot.update(c)
print(f"{'total':>12} " + " ".join(f"{tot[k]:6d}" for k in cols))
if verbose:
    for op, n in ops[verbose].most_common(40): print(f"   {op:32s} {n}")
