"""Shared helpers of the MovingHorizonEstimator parity tests: the same seeded batch through the product
(BatchMHE over the C-ABI) and through oracle/mhe.py, estimator by estimator."""
import numpy as np

import mpcqp
from mpcqp import mhe as pm
from mpcqp import synth
from oracle import estim as es
from oracle import mhe as om


def bounds_of(cfg):
    b = {}
    if np.isfinite(cfg.xabs):
        b.update(xhatmin=np.full(cfg.nxh, -cfg.xabs), xhatmax=np.full(cfg.nxh, cfg.xabs))
    if np.isfinite(cfg.wabs):
        b.update(whatmin=np.full(cfg.nxh, -cfg.wabs), whatmax=np.full(cfg.nxh, cfg.wabs))
    if np.isfinite(cfg.vabs):
        b.update(vhatmin=np.full(cfg.nym, -cfg.vabs), vhatmax=np.full(cfg.nym, cfg.vabs))
    return b


PRODUCT_KEYS = {"xhatmin": "x̂min", "xhatmax": "x̂max", "whatmin": "ŵmin", "whatmax": "ŵmax", "vhatmin": "v̂min",
                "vhatmax": "v̂max", "c_xhatmin": "c_x̂min", "c_xhatmax": "c_x̂max", "c_whatmin": "c_ŵmin", "c_whatmax": "c_ŵmax",
                "c_vhatmin": "c_v̂min", "c_vhatmax": "c_v̂max"}


def make_product(cfg, bt, lib=None, bounds=None, **kw):
    nd = cfg.nd
    bm = pm.BatchMHE(bt["Ahat"], bt["Bhu"], bt["Chm"], bt["Bhd"] if nd else None, bt["Dhdm"] if nd else None,
                     He=cfg.He, Q̂=bt["Qhat"], R̂=bt["Rhat"], P̂_0=bt["P0"], direct=cfg.direct, Cwt=getattr(cfg, "Cwt", np.inf),
                     lib=lib, **kw)
    bounds = bounds_of(cfg) if bounds is None else bounds
    if bounds:
        bm.setconstraint(**{PRODUCT_KEYS[k]: v for k, v in bounds.items()})
    return bm


def make_oracles(cfg, bt, members, bounds=None):
    """One MHEOracle per listed member, on the plant + output integrators the batch was built from."""
    bounds = bounds_of(cfg) if bounds is None else bounds
    out = []
    for b in members:
        model = es.LinModelOracle(bt["A"][b], bt["Bu"][b], bt["C"][b], bt["Bd"][b] if cfg.nd else None,
                                  np.zeros((cfg.nym, cfg.nd)) if cfg.nd else None)
        e = om.MHEOracle(model, He=cfg.He, direct=cfg.direct, Cwt=getattr(cfg, "Cwt", np.inf), sigmaQ=np.full(cfg.nx, cfg.sigmaQ),
                         sigmaR=np.full(cfg.nym, cfg.sigmaR), sigmaQint_ym=np.full(cfg.nym, cfg.sigmaQint),
                         sigmaP_0=np.full(cfg.nx, cfg.sigmaP0), sigmaPint_ym_0=np.full(cfg.nym, cfg.sigmaP0),
                         nint_ym=[1] * cfg.nym)
        assert np.allclose(e.Ah, bt["Ahat"][b]) and np.allclose(e.Chm, bt["Chm"][b]) and np.allclose(e.Q, bt["Qhat"][b])
        if bounds:
            e.setconstraint(**bounds)
        out.append(e)
    return out


def run_periods(cfg, bt, nper, members, lib=None, seed=0, bounds=None):
    """Drive product and oracles through nper periods of the same data.  Returns per period the worst
    |x̂ - x̂_oracle|, |Ŵ - Ŵ_oracle|, |P̄ - P̄_oracle| over `members`, and the product's statuses."""
    Y, U, D = synth.make_mhe_data(cfg, bt, nper, seed=seed)
    bm = make_product(cfg, bt, lib=lib, bounds=bounds)
    ors = make_oracles(cfg, bt, members, bounds=bounds)
    nxh = cfg.nxh
    rows = []
    for k in range(nper):
        y, u, d = Y[k], U[k], (D[k] if cfg.nd else None)
        xg = bm.preparestate(y, d)
        xo = np.array([e.preparestate(y[b], d[b] if cfg.nd else ()) for e, b in zip(ors, members)])
        if not cfg.direct:
            xg = bm.updatestate(u, y, d)
            xo = np.array([e.updatestate(u[b], y[b], d[b] if cfg.nd else ()) for e, b in zip(ors, members)])
        info = bm.getinfo()
        Nk = info["Nk"]
        Wo = np.array([e.Zt[e.neps + nxh:e.neps + nxh + Nk * nxh] for e in ors])
        ex = np.abs(xg[members] - xo).max()
        ew = np.abs(info["Ŵ"][members] - Wo).max()
        scale = max(1.0, np.abs(xo).max())
        if cfg.direct:
            bm.updatestate(u, y, d)
            for e, b in zip(ors, members):
                e.updatestate(u[b], y[b], d[b] if cfg.nd else ())
        Po = np.array([e.Parr_old for e in ors])
        ep = np.abs(bm.handle.get(pm.GET_PBAR)[members] - Po).max() / max(1.0, np.abs(Po).max())
        eo = np.array([e.Zt[0] if e.neps else 0.0 for e in ors])
        rows.append(dict(k=k, Nk=Nk, ex=ex / scale, ew=ew / scale, ep=ep, ee=np.abs(info["ϵ"][members] - eo).max(), eps=eo,
                         status=info["status"].copy(),
                         iters=info["iters"].copy(), ostatus=[e.status for e in ors]))
    return rows, bm


def reference_known_answers(lib=None, B=2, forms=(True, False)):
    """"MHE estimation and getinfo (LinModel)" of the reference (test/2_test_state_estim.jl:1034-1075) through the
    PRODUCT: the estimator of the reference's `sys` plant (He = 2, both forms, default integrators) stays at the
    operating point, and its estimated outputs follow a step of the measurements to 1e-3 / 1e-2 after 40 periods."""
    from tests.test_oracle_mhe import _plant
    out = {}
    for direct in forms:
        model = _plant().setop(uop=[10, 50], yop=[50, 30], dop=[5])
        e = om.MHEOracle(model, He=2, direct=direct, sigmaQ=[0.25, 0.25], sigmaP_0=[0.25, 0.25])
        rep = lambda M: np.repeat(np.asarray(M, float)[None], B, 0)
        bm = pm.BatchMHE(rep(e.Ah), rep(e.Bhu), rep(e.Chm), rep(e.Bhd), rep(e.Dhdm), He=2, Q̂=rep(e.Q), R̂=rep(e.R),
                         P̂_0=rep(e.cov.P0), direct=direct, uop=model.uop, yop_m=model.yop[e.i_ym], dop=model.dop,
                         x̂op=e.xhop, f̂op=e.fhop, lib=lib)
        yhat = lambda: np.einsum("ij,bj->bi", e.Ch, bm.x̂0) + model.yop          # evaloutput (D̂d = 0 for this plant)
        bm.preparestate([50, 30], [5])
        x = bm.updatestate([10, 50], [50, 30], [5])
        res = {"x_at_op": np.abs(x - e.xhop).max()}
        for _ in range(40):
            bm.preparestate([50, 30], [5]); bm.updatestate([11, 52], [50, 30], [5])
        bm.preparestate([50, 30], [5])
        res["y_hold"] = np.abs(yhat() - [50, 30]).max()
        for _ in range(40):
            bm.preparestate([51, 32], [5]); bm.updatestate([10, 50], [51, 32], [5])
        bm.preparestate([51, 32], [5])
        res["y_step"] = np.abs(yhat() - [51, 32]).max()
        res["tol"] = 1e-3 if direct else 1e-2
        out[direct] = res
    return out


def _plant_u():
    """LinModel(sys, Ts, i_u=[1,2]) of the reference's test module (test/0_test_module.jl), minimal realisation."""
    Ts = 400.0
    a1, a2 = np.exp(-Ts / 1800.0), np.exp(-Ts / 800.0)
    return es.LinModelOracle(np.diag([a1, a2]), np.array([[1 - a1, 1 - a1], [-(1 - a2), 1 - a2]]), np.diag([1.90, 0.74]),
                             np.zeros((2, 0)), np.zeros((2, 0)), Ts=Ts)


class _OracleMHE:
    """oracle/mhe.py behind the keyword vocabulary of BatchMHE (one estimator, answers with a leading batch axis)."""

    def __init__(self, e):
        self.e = e

    def setconstraint(self, **kw):
        inv = {v: k for k, v in PRODUCT_KEYS.items()}
        self.e.setconstraint(**{inv[k]: v for k, v in kw.items()})

    def preparestate(self, ym, d=()):
        return self.e.preparestate(ym, d)[None]

    def updatestate(self, u, ym, d=()):
        x = self.e.updatestate(u, ym, d)[None]
        assert self.e.status == 0
        return x

    def getinfo(self):
        e = self.e
        return {"Ŵ": e.Zt[e.neps + e.nxh:][:e.nxh * e.Nk][None], "V̂": e.Vhat[None], "status": np.array([e.status])}


def reference_constraint_violation(soft, lib=None, B=2, oracle=False):
    """"MHE constraint violation (LinModel)" of the reference (test/2_test_state_estim.jl:1491-1539), same call
    sequence: He = 1, nint_ym = 0, Cwt = 1e5 with every softness parameter on (soft) or Cwt = Inf (hard); a bound
    that excludes the operating point is set on x̂, then ŵ, then v̂, and the estimate / Ŵ / V̂ must sit on it
    (atol 5e-2 in the reference).  Returns {label: worst |answer - expected|} of the product (or of the oracle)."""
    model = _plant_u().setop(uop=[10, 50], yop=[50, 30])
    Cwt = 1e5 if soft else np.inf
    e = om.MHEOracle(model, He=1, direct=True, Cwt=Cwt, nint_ym=[0, 0])
    if oracle:
        bm = _OracleMHE(e)
    else:
        rep = lambda M: np.repeat(np.asarray(M, float)[None], B, 0)
        bm = pm.BatchMHE(rep(e.Ah), rep(e.Bhu), rep(e.Chm), He=1, Q̂=rep(e.Q), R̂=rep(e.R), P̂_0=rep(e.cov.P0), direct=True,
                         Cwt=Cwt, uop=model.uop, yop_m=model.yop[e.i_ym], x̂op=e.xhop, f̂op=e.fhop, lib=lib)
    big, nbig = [100.0, 100.0], [-100.0, -100.0]
    wide = dict(x̂min=nbig, x̂max=big, ŵmin=nbig, ŵmax=big, v̂min=nbig, v̂max=big)
    bm.setconstraint(**wide)
    if soft:
        bm.setconstraint(c_x̂min=[1, 1], c_x̂max=[1, 1], c_ŵmin=[0.1, 0.1], c_ŵmax=[0.1, 0.1], c_v̂min=[1, 1], c_v̂max=[1, 1])
    out = {}

    def period():
        bm.preparestate([50, 30])
        x = bm.updatestate([10, 50], [50, 30])
        info = bm.getinfo()
        assert np.all(info["status"] == 0)
        return x, info

    bm.setconstraint(x̂min=[1, 1], x̂max=big)
    out["x̂min"] = np.abs(period()[0] - [1, 1]).max()
    bm.setconstraint(x̂min=nbig, x̂max=[-1, -1])
    out["x̂max"] = np.abs(period()[0] - [-1, -1]).max()
    bm.setconstraint(**wide)
    bm.setconstraint(ŵmin=[1, 1], ŵmax=big)
    out["ŵmin"] = np.abs(period()[1]["Ŵ"] - [1, 1]).max()
    bm.setconstraint(ŵmin=nbig, ŵmax=[-1, -1])
    out["ŵmax"] = np.abs(period()[1]["Ŵ"] - [-1, -1]).max()
    bm.setconstraint(**wide)
    bm.setconstraint(v̂min=[1, 1], v̂max=big)
    out["v̂min"] = np.abs(period()[1]["V̂"] - [1, 1]).max()
    bm.setconstraint(v̂min=nbig, v̂max=[-1, -1])
    out["v̂max"] = np.abs(period()[1]["V̂"] - [-1, -1]).max()
    return out


def reference_unfilled_window(direct, lib=None, B=2, oracle=False):
    """"MHE estimation with unfilled window" of the reference (test/2_test_state_estim.jl:1313-1337): the plant
    x+ = 0.5x + u, y = x (linear, so the LinModel estimator applies), an integrator on the input (nint_u = [1]),
    He = 3; the estimator is told u = 0 while the plant receives u = 0.1 for 40 periods: the estimated output must
    equal the plant output (atol 1e-6).  Returns |ŷ - y|."""
    model = es.LinModelOracle(np.array([[0.5]]), np.array([[1.0]]), np.array([[1.0]]), np.zeros((1, 0)), np.zeros((1, 0)), Ts=10.0)
    e = om.MHEOracle(model, He=3, direct=direct, nint_u=[1])
    if oracle:
        prep, upd = (lambda y: e.preparestate(y)), (lambda u, y: e.updatestate(u, y))
        yhat = lambda: e.evaloutput()
    else:
        rep = lambda M: np.repeat(np.asarray(M, float)[None], B, 0)
        bm = pm.BatchMHE(rep(e.Ah), rep(e.Bhu), rep(e.Chm), He=3, Q̂=rep(e.Q), R̂=rep(e.R), P̂_0=rep(e.cov.P0), direct=direct, lib=lib)
        prep, upd = (lambda y: bm.preparestate(y)), (lambda u, y: bm.updatestate(u, y))
        yhat = lambda: np.einsum("ij,bj->bi", e.Ch, bm.x̂0)
    x = np.zeros(1)
    for _ in range(40):
        y = x.copy()
        prep(y)
        upd([0.0], y)
        x = 0.5 * x + 0.1
    prep(x.copy())
    return float(np.abs(yhat() - x).max())


def reference_setmodel(lib=None, B=2, oracle=False):
    """"MHE set model" of the reference (test/2_test_state_estim.jl:1668-1718): He = 5, nint_ym = 0, predictor form,
    x̂ ∈ [-1000, 1000]; setmodel! with a new Â and new (uop, yop), then new (xop, fop); a second estimator goes through
    setmodel! + initstate!.  Returns {label: (value, expected)} from the product (or from the oracle)."""
    def lin(a, b, uop, yop, xop, fop):
        return es.LinModelOracle(np.array([[a]]), np.array([[b]]), np.array([[1.0]]), np.zeros((1, 0)), np.zeros((1, 0)),
                                 Ts=10.0).setop(uop=[uop], yop=[yop], xop=[xop], fop=[fop])
    rep = lambda M: np.repeat(np.asarray(M, float)[None], B, 0)

    def product_of(e, model):
        bm = pm.BatchMHE(rep(e.Ah), rep(e.Bhu), rep(e.Chm), He=5, Q̂=rep(e.Q), R̂=rep(e.R), P̂_0=rep(e.cov.P0), direct=False,
                         uop=model.uop, yop_m=model.yop[e.i_ym], x̂op=e.xhop, f̂op=e.fhop, lib=lib)
        return bm

    def product_setmodel(bm, e, model):
        bm.setmodel(rep(e.Ah), rep(e.Bhu), rep(e.Chm), uop=model.uop, yop_m=model.yop[e.i_ym], x̂op=e.xhop, f̂op=e.fhop)

    out = {}
    m1 = lin(0.5, 0.3, 2.0, 50.0, 3.0, 3.0)
    e = om.MHEOracle(m1, He=5, direct=False, nint_ym=[0])
    e.setconstraint(xhatmin=[-1000], xhatmax=[1000])
    bm = None
    if not oracle:
        bm = product_of(e, m1)
        bm.setconstraint(x̂min=[-1000], x̂max=[1000])
    first = lambda v: float(np.asarray(v).ravel()[0])
    # (the oracle goes through every period too: it provides the augmented matrices of setmodel! and the comparison values)
    def step(u, y):
        xo = first(e.updatestate(u, y))
        return xo if oracle else first(bm.updatestate(u, y))

    def prep(y):
        xo = first(e.preparestate(y))
        return xo if oracle else first(bm.preparestate(y))
    prep([50.0])
    out["x̂ after the first period"] = (step([2.0], [50.0]), 3.0)
    m2 = lin(0.2, 0.3, 3.0, 55.0, 3.0, 3.0)
    e.setmodel(m2)
    if not oracle:
        product_setmodel(bm, e, m2)
        out["ŷ after setmodel!"] = (first(e.Ch @ bm.x̂0[0] + m2.yop), 55.0)
    else:
        out["ŷ after setmodel!"] = (first(e.evaloutput()), 55.0)
        out["lastu0"] = (first(e.lastu0), -1.0)
        out["U0[1]"] = (first(e.U0), -1.0)
        out["Y0m[1]"] = (first(e.Y0m), -5.0)
    out["x̂ with the new model"] = (prep([55.0]), 3.0)
    m3 = lin(0.2, 0.3, 3.0, 55.0, 8.0, 8.0)
    e.setmodel(m3)
    if not oracle:
        product_setmodel(bm, e, m3)
        out["x̂0 after the new operating point"] = (first(bm.x̂0), -5.0)
        out["x̂0min"] = (first(bm._con["xmin"]), -1008.0)
        # the estimator keeps working on the shifted windows: product against oracle over three more periods
        for k in range(3):
            xg = step([3.0 + 0.1 * k], [55.0 + 0.2 * k])
            out[f"x̂ period {k} after setmodel!"] = (xg, first(e.x0 + e.xhop))
    else:
        out["x̂0 after the new operating point"] = (first(e.x0), -5.0)
        out["X̂0_old[1]"] = (first(e.X0_old), -5.0)
        out["x̂0arr_old"] = (first(e.x0arr_old), -5.0)
        out["X̂0min"] = (first(e.con["X0min"]), -1008.0)
    # second estimator: setmodel! followed by initstate!
    m4 = lin(0.5, 0.3, 2.0, 50.0, 3.0, 3.0)
    e2 = om.MHEOracle(m4, He=5, direct=False, nint_ym=[0])
    m5 = lin(0.5, 0.9, 3.0, 55.0, 8.0, 8.0)
    if oracle:
        out["x̂ (second estimator)"] = (first(e2.updatestate([3.0], [50.0])), 3.3)
        e2.setmodel(m5)
        e2.initstate([3.0], [55.0])
        out["x̂ after setmodel! + initstate!"] = (first(e2.updatestate([4.0], [55.0])), 8.9)
    else:
        b2 = product_of(e2, m4)
        out["x̂ (second estimator)"] = (first(b2.updatestate([3.0], [50.0])), 3.3)
        e2.setmodel(m5)
        product_setmodel(b2, e2, m5)
        x_init = e2.initstate([3.0], [55.0])
        b2.initstate(x_init, u=[3.0])
        out["x̂ after setmodel! + initstate!"] = (first(b2.updatestate([4.0], [55.0])), 8.9)
    return out


def window_long_bounds(lib=None, B=3, seed=21, soft=False, nper=9, csoft=False, eps_seen=None, noop_setmodel=False):
    """setconstraint!(estim; X̂min, ..., V̂max): a bound per channel AND stage (construct.jl:858-935), product against
    oracle over a growing and then moving window.  Returns the worst relative errors (x̂, Ŵ) and the number of periods in
    which some stage bound of the oracle's optimum was active."""
    cfg = synth.MheConfig("winlong", nx=3, nu=1, nym=2, nd=0, He=5, **({"Cwt": 1e4} if (soft or csoft) else {}))
    bt = synth.make_mhe_batch(cfg, B, seed=seed)
    Y, U, D = synth.make_mhe_data(cfg, bt, nper, seed=seed)
    rng = np.random.default_rng(seed)
    nx, nym, He = cfg.nxh, cfg.nym, cfg.He
    # stage-dependent boxes: tight on some stages, absent (Inf) on others
    Xw = rng.uniform(0.3, 1.5, nx * (He + 1)); Xw[rng.random(Xw.size) < 0.3] = np.inf
    Ww = rng.uniform(0.05, 0.4, nx * He); Ww[rng.random(Ww.size) < 0.3] = np.inf
    Vw = rng.uniform(0.2, 0.8, nym * He); Vw[rng.random(Vw.size) < 0.3] = np.inf
    bm = make_product(cfg, bt, lib=lib, bounds={})
    ors = make_oracles(cfg, bt, list(range(B)), bounds={})
    if soft:
        bm.setconstraint(c_x̂max=np.full(nx, 0.5), c_v̂min=np.ones(nym))
        for e in ors:
            e.setconstraint(c_xhatmax=np.full(nx, 0.5), c_vhatmin=np.ones(nym))
    if csoft:
        # window-long SOFTNESS (C_x̂min ... C_v̂max, construct.jl:937-1020): a softness per channel and stage, zero (hard) on some
        # rows; a per-channel keyword given first is overwritten where a window-long vector follows
        Cx0 = rng.uniform(0.1, 1.0, nx * (He + 1)); Cx0[rng.random(Cx0.size) < 0.4] = 0.0
        Cx1 = rng.uniform(0.1, 1.0, nx * (He + 1)); Cx1[rng.random(Cx1.size) < 0.4] = 0.0
        Cw1 = rng.uniform(0.1, 1.0, nx * He); Cw1[rng.random(Cw1.size) < 0.4] = 0.0
        Cv0 = rng.uniform(0.1, 1.0, nym * He); Cv0[rng.random(Cv0.size) < 0.4] = 0.0
        bm.setconstraint(c_ŵmin=np.full(nx, 0.3), c_x̂min=np.full(nx, 9.0))
        bm.setconstraint(C_x̂min=Cx0, C_x̂max=Cx1, C_ŵmax=Cw1, C_v̂min=Cv0)
        for e in ors:
            e.setconstraint(c_whatmin=np.full(nx, 0.3), c_xhatmin=np.full(nx, 9.0))
            e.setconstraint(C_xhatmin=Cx0, C_xhatmax=Cx1, C_whatmax=Cw1, C_vhatmin=Cv0)
        Xw = np.where(np.isinf(Xw), Xw, 0.6 * Xw); Ww = np.where(np.isinf(Ww), Ww, 0.6 * Ww)     # tighter: the slack is used
    bm.setconstraint(X̂min=-Xw, X̂max=Xw, Ŵmin=-Ww, Ŵmax=Ww, V̂min=-Vw, V̂max=Vw)
    for e in ors:
        e.setconstraint(Xhatmin=-Xw, Xhatmax=Xw, Whatmin=-Ww, Whatmax=Ww, Vhatmin=-Vw, Vhatmax=Vw)
    if noop_setmodel:
        # setmodel! with nothing changed re-uploads the bounds (new deviation values): the window-long softness set before
        # must survive it (mpcqp_mhe_set_bounds[_window] once dropped CLS_C and the kernel then read the window-long
        # softness arrays as per-channel ones: ADVICE r4)
        bm.setmodel()
    ex = ew = 0.0
    active = 0
    for k in range(nper):
        xg = bm.preparestate(Y[k])
        xo = np.array([e.preparestate(Y[k][b]) for b, e in enumerate(ors)])
        if not cfg.direct:
            xg = bm.updatestate(U[k], Y[k])
            xo = np.array([e.updatestate(U[k][b], Y[k][b]) for b, e in enumerate(ors)])
        info = bm.getinfo()
        assert np.all(info["status"] == 0) and all(e.status == 0 for e in ors), k
        Nk = info["Nk"]
        Wo = np.array([e.Zt[e.neps + nx:e.neps + nx + Nk * nx] for e in ors])
        sc = max(1.0, np.abs(xo).max())
        ex = max(ex, np.abs(xg - xo).max() / sc)
        ew = max(ew, np.abs(info["Ŵ"] - Wo).max() / sc)
        if eps_seen is not None and np.isfinite(cfg.Cwt):
            eps_seen.append(max(float(e.Zt[0]) for e in ors))
        for e in ors:
            Xb = Xw[nx:][(He - Nk) * nx:]
            active += int(np.any(np.abs(np.abs(e.X0[:Nk * nx]) - Xb) <= 1e-6) or np.any(np.abs(np.abs(Wo) - Ww[(He - Nk) * nx:]) <= 1e-6))
        if cfg.direct:
            bm.updatestate(U[k], Y[k])
            for b, e in enumerate(ors):
                e.updatestate(U[k][b], Y[k][b])
    return ex, ew, active


def random_family(seed, lib=None, B=5, **solver):
    """One randomised MovingHorizonEstimator family (dimensions, form, horizon, bound classes, hard / soft) driven
    through He + 3 periods on the product and on oracle/mhe.py, every member compared.  Returns (worst relative
    error over the periods where the oracle's QP was solved, number of compared solves, failures seen)."""
    rng = np.random.default_rng(1000 + seed)
    nx, nym = int(rng.integers(1, 9)), int(rng.integers(1, 5))
    kw = dict(nx=nx, nu=int(rng.integers(0, 4)), nym=nym, nd=int(rng.integers(0, 3)), He=int(rng.integers(1, 11)),
              direct=bool(rng.integers(0, 2)))
    cls = int(rng.integers(0, 6))           # 0 none, 1 x̂, 2 ŵ, 3 v̂, 4 x̂ + v̂, 5 ŵ + v̂
    if cls in (1, 4):
        kw["xabs"] = float(rng.uniform(0.6, 2.0))
    if cls in (2, 5):
        kw["wabs"] = float(rng.uniform(0.1, 0.4))
    if cls in (3, 4, 5):
        kw["vabs"] = float(rng.uniform(0.3, 0.8))
    soft = cls != 0 and rng.random() < 0.4
    if soft:
        kw["Cwt"] = float(10.0 ** rng.uniform(2, 5))
    cfg = synth.MheConfig(f"fam{seed}", **kw)
    bt = synth.make_mhe_batch(cfg, B, seed=seed)
    bounds = bounds_of(cfg)
    if soft:
        nxh = cfg.nxh
        for key, n in (("xhat", nxh), ("what", nxh), ("vhat", cfg.nym)):
            if key + "min" in bounds:
                bounds["c_" + key + "min"] = np.where(rng.random(n) < 0.6, rng.uniform(0.2, 1.5, n), 0.0)
                bounds["c_" + key + "max"] = np.where(rng.random(n) < 0.6, rng.uniform(0.2, 1.5, n), 0.0)
    nper = cfg.He + 3
    Y, U, D = synth.make_mhe_data(cfg, bt, nper, seed=seed)
    bm = make_product(cfg, bt, lib=lib, bounds=bounds, **solver)
    ors = make_oracles(cfg, bt, range(B), bounds=bounds)
    clean = np.ones(B, bool)
    worst, ncmp, nfail = 0.0, 0, 0
    for k in range(nper):
        y, u, d = Y[k], U[k], (D[k] if cfg.nd else None)
        xg = bm.preparestate(y, d)
        if not cfg.direct:
            xg = bm.updatestate(u, y, d)
        for b, e in enumerate(ors):
            xo = e.preparestate(y[b], d[b] if cfg.nd else ())
            if not cfg.direct:
                xo = e.updatestate(u[b], y[b], d[b] if cfg.nd else ())
            if not clean[b]:
                continue
            if e.status != 0:                  # infeasible window (x̂ + v̂ or ŵ + v̂ bounds can contradict the data)
                assert bm.status[b] != 0, (seed, k, b, "the oracle failed, the product reports success")
                clean[b] = False
                nfail += 1
                continue
            assert bm.status[b] == 0, (seed, k, b, "the product failed on a window the oracle solved", kw)
            worst = max(worst, np.abs(xg[b] - xo).max() / max(1.0, np.abs(xo).max()))
            ncmp += 1
        if cfg.direct:
            bm.updatestate(u, y, d)
            for b, e in enumerate(ors):
                e.updatestate(u[b], y[b], d[b] if cfg.nd else ())
    return worst, ncmp, nfail
