"""Root-cause helper for an on-demand specialisation that mpcqp_prepare's comparison flags (VERDICT r3 item 1b:
family seed 2014 = shape nu=3 ny=2 nx̂=5 Hp=27 Hc=19, rows 0xc1 -- "14 iterations vs 6 on the runtime-dimension kernel").

    python scripts/diag_spec.py [seed] [--huge|--large]      (on the GPU box)

runs the family's handle twice, in two processes -- once on its specialisation, once with MPCQP_FORCE_GENERIC=1 -- on the
very inputs of the library's self-test (the LCG of csrc/mpcqp_host.hip: self_test_spec_impl) and prints, per controller:
the iterates after k = 1, 2, 3, 5, 8 interior-point iterations (MPCQP_FLAG_KEEP_ITERATE), the complete solve with and
without the active-set polish (status, factorisations, audit record).  If the capped iterates agree to rounding the
Newton matrix, right-hand sides and row passes of the specialisation are right and a different iteration count of the
complete solve is the polish being accepted at a different attempt.
"""
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def lcg_inputs(n, nxh, nry):
    m = (1 << 64) - 1
    st = 0x9E3779B97F4A7C15
    out = []
    for _ in range(n * nxh + n * nry):
        st = (st * 6364136223846793005 + 1442695040888963407) & m
        out.append((st >> 11) / 9007199254740992.0 * 2.0 - 1.0)
    x = np.array(out[:n * nxh]).reshape(n, nxh)
    ry = 2.0 * np.array(out[n * nxh:]).reshape(n, nry)
    return x, ry


def child(seed, mode):
    import warnings
    import mpcqp
    from mpcqp import api
    from tests import parity_util as pu
    kinds = []
    # build the family's handle exactly like the test does (constructor + setconstraint), stop before the first step
    captured = {}
    orig = mpcqp.BatchLinMPC.moveinput

    def grab(self, *a, **k):
        captured["mpc"] = self
        raise StopIteration

    mpcqp.BatchLinMPC.moveinput = grab
    lib = None
    if os.environ.get("MPCQP_DIAG_EMU"):          # (script check on a box without a GPU: the CPU wave emulator of tests/emu)
        lib = api.load_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "emu", "libmpcqp_emu.so"))
    try:
        pu.run_random_case(seed, lib=lib, B=3, large=(mode == "large"), huge=(mode == "huge"), kinds=kinds)
    except StopIteration:
        pass
    mpcqp.BatchLinMPC.moveinput = orig
    mpc = captured["mpc"]
    hd = mpc.hd
    hd.set_flags(hd.flags | api.FLAG_RY_CONSTANT)
    with warnings.catch_warnings():
        warnings.simplefilter("always")
        kind = hd.prepare()
    n = min(hd.B, 8)
    x, ry = lcg_inputs(n, hd.nxhat, hd.ny)
    B = hd.B
    X = np.zeros((B, hd.nxhat)); X[:n] = x
    RY = np.zeros((B, hd.ny)); RY[:n] = ry
    LU = np.zeros((B, hd.nu))
    d0 = np.zeros((B, hd.nd)) if hd.nd else None
    Dh = np.zeros((B, hd.nD)) if hd.nd else None
    base = (hd.flags | api.FLAG_RY_CONSTANT | api.FLAG_COLD_START) & ~(api.FLAG_KEEP_QP | api.FLAG_WARM_DUAL)
    res = {"kind": kind, "dims": [hd.nu, hd.ny, hd.nxhat, hd.Hp, hd.Hc, hd.nZ], "rows": hex(hd.row_groups()), "runs": {}}

    def run(tag, flags, max_iter):
        hd.set_flags(flags)
        hd.set_iteration_limit(max_iter)
        Z = np.zeros((B, hd.nZ))
        u0, st, it = hd.step(X, LU, RY, Z, d0=d0, Dhat0=Dh)
        au = hd.get(api.GET_AUDIT)
        res["runs"][tag] = {"Z": Z[:n].tolist(), "status": st[:n].tolist(), "iters": it[:n].tolist(), "audit": au[:n].tolist()}

    for k in (1, 2, 3, 5, 8):
        run(f"k{k}", base | api.FLAG_KEEP_ITERATE, k)
    run("full", base, 0)
    run("nopolish", base | api.FLAG_NO_POLISH, 0)
    print("@@" + json.dumps(res))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].lstrip("-").isdigit() else 2014
    mode = "huge" if "--huge" in sys.argv else "small" if "--small" in sys.argv else "large"
    if "--child" in sys.argv:
        return child(seed, mode)
    out = {}
    for name, env in (("spec", {"MPCQP_SELFTEST_VERBOSE": "1"}), ("generic", {"MPCQP_FORCE_GENERIC": "1"})):
        e = dict(os.environ, **env)
        p = subprocess.run([sys.executable, __file__, str(seed), "--" + mode, "--child"], env=e, capture_output=True, text=True)
        for line in p.stderr.splitlines():
            if "[mpcqp]" in line:
                print(f"({name}) {line}")
        js = [l for l in p.stdout.splitlines() if l.startswith("@@")]
        if not js:
            print(p.stdout[-3000:], p.stderr[-3000:])
            raise SystemExit(f"{name} child failed")
        out[name] = json.loads(js[0][2:])
    a, b = out["spec"], out["generic"]
    print(f"family {seed} ({mode}): dims nu,ny,nx̂,Hp,Hc,nZ̃ = {a['dims']} rows {a['rows']}; kernel kinds: spec run {a['kind']}, generic run {b['kind']}")
    for tag in a["runs"]:
        ra, rb = a["runs"][tag], b["runs"][tag]
        Za, Zb = np.array(ra["Z"]), np.array(rb["Z"])
        rel = np.abs(Za - Zb).max(axis=1) / np.maximum(1.0, np.abs(Zb).max(axis=1))
        print(f"  {tag:9s} max rel |Z_spec - Z_generic| per controller {np.array2string(rel, precision=2)}  status {ra['status']} / {rb['status']}  "
              f"iterations {ra['iters']} / {rb['iters']}  polished {[int(x[3]) for x in ra['audit']]} / {[int(x[3]) for x in rb['audit']]}")


if __name__ == "__main__":
    main()
