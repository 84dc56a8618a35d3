"""Per-phase x instruction-class table of the interior-point loop of a specialised step kernel.
Static analysis of the gfx950 assembly of a build with -DMPCQP_ISA_MARKERS (phase boundaries as assembly comments with
scheduling barriers, csrc/mpcqp_bodies.h), weighted by the known trip counts: the two-pass Newton loop x 2, the K loops
of E'DE by their trip counts; everything else in the loop is straight-line code for compile-time dimensions.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -w -DMPCQP_ISA_MARKERS -Icsrc \
        -DMPCQP_SPEC_DIMS=4,4,16,30,10,1,141u,1 csrc/mpcqp_spec.hip -o /tmp/c3_markers.s
  python scripts/isa_phase_table.py /tmp/c3_markers.s [etde trips, e.g. 2,13,11]

Classes: fma64 (v_fma/v_fmac_f64), f64 (mul/add/min/max), trans (v_rcp/v_rsq_f64), dpp64 (v_*_f64_dpp), int (v_*_u32/i32/
lshl/and/or/mad/mul_u32...), sel (v_cndmask/v_cmp*), mov (v_mov/v_accvgpr), lane (v_readlane/v_writelane/v_readfirstlane),
dpp32 (other *_dpp), mfma, lds, vmem, salu, br, wait (s_waitcnt / s_nop; nops listed with their wait states)."""
import collections, re, sys

path = sys.argv[1]
trips = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 13, 11]
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_ZN5mpcqp\d+k_step_s", l)][0]
end = [i for i, l in enumerate(lines) if i > start and ".amdhsa_kernel" in l][0]

PH = {"0": "G v: rows (apply_G)", "1": "G'w (apply_Gt)", "2": "H~ load + H~ z", "3": "G'DG: row pass", "4": "E'DE (matrix cores)",
      "5": "G'DG: U rows, diagonal, eps row", "6": "Cholesky", "7": "triangular solves", "8": "G'w: row pass + reduce",
      "9": "G'w: box / U / eps part", "10": "G'w: E'w", "11": "G v: held sums (ucum)", "12": "G v: E v", "13": "step rule + update pass",
      "14": "polish (opaque)", "rowstep": "row step pass (ds, dl, ratio test)", "looptop": "convergence test / bookkeeping"}


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


insts, labels, marks = [], {}, {}
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r"^(\.L[\w$]+):", s)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    m = re.match(r";\s*MPCQP_MARK\s+(tic|toc)\s*(\S*)", s)
    if m:
        marks.setdefault(len(insts), []).append((m.group(1), m.group(2)))
        continue
    if not s or s.startswith((".", ";", "/")):
        continue
    op = s.split()[0]
    insts.append((op, s))
loops = []
for i, (op, s) in enumerate(insts):
    if op.startswith("s_cbranch") or op == "s_branch":
        t = s.split()[-1]
        if t in labels and labels[t] <= i:
            loops.append((labels[t], i))
# region of every instruction: stack of open tics in layout order
region = [None] * len(insts)
stack, pend = [], []
# first pass: find matching toc for each tic to name regions
names = {}
open_ = []
for i in range(len(insts) + 1):
    for kind_, nm in marks.get(i, []):
        if nm in ("tic14_", "14"):           # the polish: its body is laid out away from its call site -- not a span
            continue
        if kind_ == "tic":
            open_.append(i)
        elif open_:
            names[open_.pop()] = (nm, i)
spans = sorted((a, b, nm) for a, (nm, b) in names.items())
for a, b, nm in spans:                       # outer spans first (sorted by start), inner overwrite -- except inside polish
    for i in range(a, b):
        region[i] = nm
# main loop: the smallest loop that holds the loop-top span (convergence test, exact residual evaluation and the
# inlined polish all sit inside it) and a step-rule span; the phases of a TYPICAL iteration (no exact re-evaluation, no
# polish) are the spans laid out between the end of the loop-top span and the back edge
lt = [(a, b) for a, b, nm in spans if nm == "looptop"]
upd = [a for a, b, nm in spans if nm == "13"]
cands = [(b - a, a, b) for a, b in loops if any(a <= x and y <= b + 8 for x, y in lt) and any(a <= u <= b for u in upd)]
_, L0, L1 = min(cands)
LT0, LT1 = [(a, b) for a, b in lt if L0 <= a <= L1][0]
L0 = LT1
inner = {}
for a, b in loops:                      # several back edges to one header are one loop
    if L0 <= a and b <= L1:
        inner[a] = max(inner.get(a, 0), b)
inner = sorted(inner.items())
weight = [1.0] * len(insts)
et = 0
for a, b in sorted(inner):
    regs = collections.Counter(region[i] for i in range(a, b + 1))
    top = regs.most_common(1)[0][0]
    has_solve = any(region[i] == "7" for i in range(a, b + 1))
    has_chol = any(region[i] == "6" for i in range(a, b + 1))
    if top == "4" and et < len(trips):
        mult = trips[et]; et += 1
    elif has_solve and not has_chol:
        mult = 2
    else:
        mult = 1
    if mult != 1:
        for i in range(a, b + 1):
            weight[i] *= mult
    print(f"# inner loop {a}-{b} ({b - a + 1} instr, mostly region {top}): x{mult}", file=sys.stderr)
tab = collections.defaultdict(collections.Counter)
nopstates = collections.Counter()
for i in range(L0, L1 + 1):
    op, s = insts[i]
    r = region[i] or "other"
    k = kind(op, s)
    tab[r][k] += weight[i]
    if k == "nop":
        nopstates[r] += weight[i] * (int(s.split()[1]) + 1)
cols = ["fma64", "f64", "trans", "dpp64", "int", "sel", "mov", "lane", "dpp32", "mfma", "lds", "vmem", "salu", "br", "wait", "nop"]
print(f"interior-point loop, typical iteration (no exact residual re-evaluation, no polish): instructions {L0}..{L1} of the kernel ({L1 - L0 + 1} static); dynamic count per iteration")
print("%-38s" % "phase" + "".join("%7s" % c for c in cols) + "%8s%8s" % ("VALU", "nopst"))
valu_c = ["fma64", "f64", "trans", "dpp64", "int", "sel", "mov", "lane", "dpp32"]
tot = collections.Counter()
order = ["looptop", "3", "4", "5", "6", "8", "9", "10", "1", "7", "11", "12", "0", "rowstep", "13", "2", "other", "14"]
for r in order + [r for r in tab if r not in order]:
    if r not in tab:
        continue
    c = tab[r]
    v = sum(c[k] for k in valu_c)
    print("%-38s" % (PH.get(r, r))[:38] + "".join("%7.0f" % c[k] for k in cols) + "%8.0f%8.0f" % (v, nopstates[r]))
    if r != "14":
        tot.update(c); tot["VALU"] += v; tot["nopst"] += nopstates[r]
print("%-38s" % "TOTAL (without the polish)" + "".join("%7.0f" % tot[k] for k in cols) + "%8.0f%8.0f" % (tot["VALU"], tot["nopst"]))
