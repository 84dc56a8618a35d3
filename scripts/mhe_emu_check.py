"""Development check: the MHE kernel bodies on the CPU wave emulator against oracle/mhe.py."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
from mpcqp import mhe as pm
from oracle import estim as es, mhe as om

EMU = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "emu", "libmpcqp_emu.so")
lib = mpcqp.api.load_library(EMU)


def plant(Ts=400.0):
    a1, a2 = np.exp(-Ts / 1800.0), np.exp(-Ts / 800.0)
    A = np.diag([a1, a2])
    B = np.array([[1 - a1, 1 - a1, 1 - a1], [-(1 - a2), 1 - a2, -(1 - a2)]])
    C = np.diag([1.90, 0.74])
    return es.LinModelOracle(A, B[:, :2], C, B[:, 2:], np.zeros((2, 1)), Ts=Ts)


def run(direct, He, bounds, nper, B=5, seed=0):
    model = plant().setop(uop=[10, 50], yop=[50, 30], dop=[5])
    rng = np.random.default_rng(seed)
    ests = [om.MHEOracle(model, He=He, direct=direct) for _ in range(B)]
    for e in ests:
        if bounds:
            e.setconstraint(**bounds)
    e0 = ests[0]
    rep = lambda M: np.repeat(np.asarray(M)[None], B, 0)
    kw = {}
    bm = pm.BatchMHE(rep(e0.Ah), rep(e0.Bhu), rep(e0.Chm), rep(e0.Bhd), rep(e0.Dhdm), He=He, Q̂=rep(e0.Q), R̂=rep(e0.R),
                     P̂_0=rep(e0.cov.P0), direct=direct, uop=model.uop, yop_m=model.yop[e0.i_ym], dop=model.dop,
                     x̂op=e0.xhop, f̂op=e0.fhop, lib=lib)
    if bounds:
        m = {"xhatmin": "x̂min", "xhatmax": "x̂max", "whatmin": "ŵmin", "whatmax": "ŵmax", "vhatmin": "v̂min", "vhatmax": "v̂max"}
        bm.setconstraint(**{m[k]: v for k, v in bounds.items()})
    worst = 0.0
    for k in range(nper):
        y = np.array([53.0, 26.0]) + rng.standard_normal((B, 2))
        u = np.array([12.0, 48.0]) + rng.standard_normal((B, 2))
        d = np.array([5.0]) + 0.3 * rng.standard_normal((B, 1))
        t0 = time.time()
        xg = bm.preparestate(y, d)
        xo = np.array([e.preparestate(y[b], d[b]) for b, e in enumerate(ests)])
        if direct:
            info = bm.getinfo()
            err = np.abs(xg - xo).max()
            Zo = np.array([e.Zt for e in ests])
            zerr = np.abs(info["Ŵ"] - Zo[:, e0.nxh:e0.nxh + info["Nk"] * e0.nxh]).max()
            print(f"  k={k} Nk={info['Nk']} st={info['status'].tolist()} it={info['iters'].tolist()} |x̂ err|={err:.2e} |Ŵ err|={zerr:.2e}  {time.time()-t0:.1f}s")
            worst = max(worst, err, zerr)
        xg = bm.updatestate(u, y, d)
        xo = np.array([e.updatestate(u[b], y[b], d[b]) for b, e in enumerate(ests)])
        if not direct:
            info = bm.getinfo()
            err = np.abs(xg - xo).max()
            print(f"  k={k} Nk={info['Nk']} st={info['status'].tolist()} it={info['iters'].tolist()} |x̂ err|={err:.2e}  {time.time()-t0:.1f}s")
            worst = max(worst, err)
        Pg = bm.handle.get(pm.GET_PBAR)
        Po = np.array([e.Parr_old for e in ests])
        worst = max(worst, np.abs(Pg - Po).max())
    print(f"direct={direct} He={He} bounds={bounds}: worst {worst:.3e}")
    return worst


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "free"):
        run(True, 3, None, 6)
        run(False, 3, None, 6)
    if which in ("all", "x"):
        run(True, 4, dict(xhatmax=[0.1, np.inf, np.inf, np.inf], xhatmin=[-np.inf, -0.5, -np.inf, -np.inf]), 7)
    if which in ("all", "w"):
        run(True, 4, dict(whatmax=[0.05, 0.05, 0.05, 0.05], whatmin=[-0.05, -0.05, -0.05, -0.05]), 7)
    if which in ("all", "v"):
        run(True, 4, dict(vhatmin=[-0.5, -0.5], vhatmax=[0.5, 0.6]), 7)
        run(False, 4, dict(vhatmin=[-0.5, -0.5], xhatmax=[0.1, np.inf, np.inf, np.inf]), 7)
