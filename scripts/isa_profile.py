"""Static profile of a step kernel's gfx950 assembly: instructions per source function of
csrc/mpcqp_bodies.h (needs -gline-tables-only) and per loop.   Usage:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -w -gline-tables-only -Icsrc \
        -DMPCQP_SPEC_DIMS=4,4,16,30,10,1,141u,1 --save-temps csrc/mpcqp_spec.hip -o /tmp/x.o
  python scripts/isa_profile.py mpcqp_spec-hip-amdgcn-amd-amdhsa-gfx950.s [k_step_s]"""
import bisect, collections, re, sys

path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_step_s"
src = sys.argv[3] if len(sys.argv) > 3 else "/root/repo/modelpredictivecontrol.jl_amd/csrc/mpcqp_bodies.h"
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_ZN5mpcqp\d+" + kern, l)][0]
end = [i for i, l in enumerate(lines) if i > start and ".amdhsa_kernel" in l][0]
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
# function table of the source: "MPCQP_HD ... name(" / "__device__ ... name(" at member indentation
funcs = []
for n, l in enumerate(open(src).read().split("\n"), 1):
    m = re.match(r"\s*(?:template <[^>]*>\s*)?(?:MPCQP_HD|__device__ __forceinline__)\s+(?:static\s+)?(?:inline\s+)?[\w:<>,\s\*&]*?\b(\w+)\(", l)
    if m and m.group(1) not in ("if", "for"):
        funcs.append((n, m.group(1)))
starts = [f[0] for f in funcs]
labels, insts, cur = {}, [], None
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r"^(\.L[\w$]+):", s)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if not s or s.startswith((".", ";", "/")):
        continue
    insts.append((s.split()[0], s, cur))


def kind(op):
    if "mfma" in op: return "mfma"
    if op.startswith(("v_fma_f64", "v_fmac_f64")): return "fma64"
    if op.startswith(("v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_rcp_f64", "v_rsq_f64")): return "f64"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if "readlane" in op or "writelane" in op: return "lane"
    if "dpp" in op: return "dpp"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_waitcnt", "s_nop")): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_"): return "salu"
    return "other"


def fn_of(loc):
    if not loc or loc[0] != src.split("/")[-1]:
        return loc[0] if loc else "?"
    i = bisect.bisect_right(starts, loc[1]) - 1
    return funcs[i][1] if i >= 0 else "pre"


print("total", len(insts), dict(collections.Counter(kind(o) for o, _, _ in insts)))
by = collections.defaultdict(collections.Counter)
for op, s, loc in insts:
    by[fn_of(loc)][kind(op)] += 1
print("\n-- static instructions per source function")
for f, c in sorted(by.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-22s %5d  %s" % (f, sum(c.values()), dict(c)))
loops = []
for i, (op, s, loc) in enumerate(insts):
    if op.startswith("s_cbranch") or op == "s_branch":
        t = s.split()[-1]
        if t in labels and labels[t] <= i:
            loops.append((labels[t], i))
loops.sort()
print("\n-- loops (instruction index range, size, kinds, main source functions)")
for a, b in loops:
    seg = insts[a:b + 1]
    c = collections.Counter(kind(o) for o, _, _ in seg)
    fs = collections.Counter(fn_of(loc) + ":" + str(loc[1] if loc else 0) for _, _, loc in seg).most_common(3)
    inner = not any(x >= a and y <= b and (x, y) != (a, b) for x, y in loops)
    print(("inner " if inner else "outer ") + "%5d-%5d n=%4d" % (a, b, b - a + 1), dict(c), fs)
if len(sys.argv) > 5:      # dump a range
    lo, hi = int(sys.argv[4]), int(sys.argv[5])
    inv = collections.defaultdict(list)
    for k, v in labels.items():
        inv[v].append(k)
    for i in range(lo, hi + 1):
        for k in inv.get(i, []):
            print(k + ":")
        print("%5d  %-70s %s" % (i, insts[i][1].split(";")[0], insts[i][2][1] if insts[i][2] else ""))
