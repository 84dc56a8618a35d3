"""GPU parity of the batched linear MovingHorizonEstimator (BASELINE configs[4]) against
oracle/mhe.py, through the C-ABI (BatchMHE -> mpcqp_mhe_*): growing and moving windows, arrival
covariance, both forms, every bound class, every register-column count of the kernel; at the full
batch of config 5 through the reference's own pin -- an unconstrained MHE IS the Kalman filter
(test/2_test_state_estim.jl:1750-1777) -- and through feasibility of all bounds."""
import numpy as np
import pytest

import mpcqp
from mpcqp import mhe as pm
from mpcqp import synth
from tests import mhe_util

pytestmark = pytest.mark.gpu
# Interior-point iterate (complementarity gap 1e-12, no active-set polish in this kernel) against the oracle's
# exact active-set optimum, relative to max(1, |x̂|): a weakly active bound (s ~ lambda ~ 1e-6) leaves ~3e-7; the
# north-star tolerance is 1e-5.
TOL = 2e-6


def _check(rows, tol=TOL):
    for r in rows:
        assert all(s == 0 for s in r["ostatus"]), r
        assert np.all(r["status"] == 0), (r["k"], np.flatnonzero(r["status"]))
        assert r["ex"] <= tol and r["ew"] <= tol and r["ep"] <= 1e-12, r


def test_c5_window_growth_and_motion_match_oracle():
    cfg = synth.C5
    bt = synth.make_mhe_batch(cfg, 64, seed=1)
    rows, bm = mhe_util.run_periods(cfg, bt, cfg.He + 5, [0, 9, 21, 33, 47, 63])
    _check(rows)
    assert bm.handle.register_columns() == 12 and rows[-1]["Nk"] == cfg.He
    assert max(r["iters"].max() for r in rows) > 5         # the bounds are active: a real QP, not a linear solve
    info = bm.getinfo()
    assert np.abs(info["X̂"]).max() <= cfg.xabs + 1e-8 and np.abs(info["x̂arr"]).max() <= cfg.xabs + 1e-8


@pytest.mark.parametrize("kw", [
    dict(nx=1, nu=1, nym=1, nd=0, He=6, xabs=0.7),                          # NX = 4 (nx̂ = 2)
    dict(nx=4, nu=2, nym=3, nd=2, He=8, xabs=1.2),                          # NX = 8, measured disturbances
    dict(nx=4, nu=2, nym=3, nd=1, He=8, wabs=0.25),                         # ŵ bounds
    dict(nx=4, nu=2, nym=3, nd=0, He=8, vabs=0.3),                          # v̂ bounds
    dict(nx=4, nu=2, nym=3, nd=1, He=8, xabs=1.5, direct=False),            # predictor form
    dict(nx=11, nu=3, nym=5, nd=0, He=10, xabs=1.5),                        # NX = 16
    dict(nx=2, nu=0, nym=2, nd=0, He=5, vabs=0.4),                          # no manipulated input
], ids=["NX4", "NX8+d", "what", "vhat", "predictor", "NX16", "nu0"])
def test_families_match_oracle(kw):
    cfg = synth.MheConfig("fam", **kw)
    B = 23                                                  # five full wavefronts and a partial one
    bt = synth.make_mhe_batch(cfg, B, seed=7)
    rows, bm = mhe_util.run_periods(cfg, bt, cfg.He + 3, [0, 5, 11, 22])
    _check(rows)


def test_heterogeneous_bounds_in_one_wavefront():
    """Estimators of one wavefront with different bound sets (one of them unbounded)."""
    cfg = synth.MheConfig("het", nx=4, nu=2, nym=2, nd=0, He=6)
    B = 8
    bt = synth.make_mhe_batch(cfg, B, seed=11)
    xmax = np.full((B, cfg.nxh), np.inf)
    xmin = np.full((B, cfg.nxh), -np.inf)
    xmax[0] = 0.6; xmin[0] = -0.6
    xmax[2, :2] = 0.3
    xmin[5, 1:] = -0.5
    Y, U, D = synth.make_mhe_data(cfg, bt, 9, seed=2)
    bm = mhe_util.make_product(cfg, bt, bounds={})
    bm.setconstraint(x̂min=xmin, x̂max=xmax)
    ors = [mhe_util.make_oracles(cfg, bt, [b], bounds=dict(xhatmin=xmin[b], xhatmax=xmax[b]))[0] for b in range(B)]
    for k in range(9):
        xg = bm.preparestate(Y[k])
        xo = np.array([e.preparestate(Y[k][b]) for b, e in enumerate(ors)])
        assert np.all(bm.status == 0) and np.abs(xg - xo).max() <= TOL * max(1.0, np.abs(xo).max())
        bm.updatestate(U[k], Y[k])
        for b, e in enumerate(ors):
            e.updatestate(U[k][b], Y[k][b])
    it = bm.getinfo()["iters"]
    assert it[1] == 0 and it[0] > 0                        # the unbounded estimator took its single Newton step


def test_infeasible_estimator_fails_alone():
    """x̂ and ŵ bounds that the data of one estimator cannot meet (the oracle's QP is infeasible too): it
    reports status 2 and keeps a finite open-loop estimate; the other estimators of its wavefront are solved."""
    cfg = synth.MheConfig("inf", nx=2, nu=2, nym=2, nd=1, He=3, xabs=0.8, wabs=0.3)
    B = 6
    bt = synth.make_mhe_batch(cfg, B, seed=2)
    Y, U, D = synth.make_mhe_data(cfg, bt, 4)
    bm = mhe_util.make_product(cfg, bt)
    ors = mhe_util.make_oracles(cfg, bt, range(B))
    clean = np.ones(B, bool)
    for k in range(4):
        xg = bm.preparestate(Y[k], D[k])
        for b, e in enumerate(ors):
            xo = e.preparestate(Y[k][b], D[k][b])
            if e.status != 0:
                assert bm.status[b] == 2 and np.all(np.isfinite(xg[b]))
                clean[b] = False
            elif clean[b]:
                assert bm.status[b] == 0 and np.abs(xg[b] - xo).max() <= TOL * max(1.0, np.abs(xo).max())
        bm.updatestate(U[k], Y[k], D[k])
        for b, e in enumerate(ors):
            e.updatestate(U[k][b], Y[k][b], D[k][b])
    assert not clean[4] and clean.sum() == B - 1


def _batched_kf(bt, cfg, Y, U, D):
    """Time-varying KalmanFilter (predictor form) of the whole batch in NumPy: kalman.jl:1235-1290."""
    B, nxh = bt["Ahat"].shape[:2]
    A, Bu, C, Bd, Q, R = bt["Ahat"], bt["Bhu"], bt["Chm"], bt["Bhd"], bt["Qhat"], bt["Rhat"]
    x, P = np.zeros((B, nxh)), bt["P0"].copy()
    out = []
    for k in range(Y.shape[0]):
        M = C @ P @ C.transpose(0, 2, 1) + R
        K = P @ C.transpose(0, 2, 1) @ np.linalg.inv(M)
        x = x + np.einsum("bij,bj->bi", K, Y[k] - np.einsum("bij,bj->bi", C, x))
        P = (np.eye(nxh) - K @ C) @ P
        x = np.einsum("bij,bj->bi", A, x) + np.einsum("bij,bj->bi", Bu, U[k])
        if cfg.nd:
            x = x + np.einsum("bij,bj->bi", Bd, D[k])
        P = A @ P @ A.transpose(0, 2, 1) + Q
        out.append(x.copy())
    return out


def test_config5_batch_unconstrained_is_the_kalman_filter():
    """BASELINE configs[4] at full size (B = 65536, nx̂ = 12, He = 20), predictor form, no bounds: every
    estimate equals the time-varying Kalman filter's (the reference's pin, atol 1e-6 there)."""
    cfg = synth.MheConfig("C5 free", nx=8, nu=4, nym=4, nd=0, He=20, direct=False)
    B = 65536
    bt = synth.make_mhe_batch(cfg, B, seed=3)
    nper = cfg.He + 3
    Y, U, D = synth.make_mhe_data(cfg, bt, nper, seed=5)
    bm = mhe_util.make_product(cfg, bt, bounds={}, keep_windows=False)
    ref = _batched_kf(bt, cfg, Y, U, D)
    worst = 0.0
    for k in range(nper):
        bm.preparestate(Y[k])
        xg = bm.updatestate(U[k], Y[k])
        assert np.all(bm.status == 0)
        worst = max(worst, np.abs(xg - ref[k]).max() / max(1.0, np.abs(ref[k]).max()))
    assert worst <= 1e-9, worst


def test_config5_batch_constrained_full_size():
    """B = 65536 of config 5 with its hard state bounds: every solve succeeds, every bound holds, a
    sample of estimators equals the oracle."""
    cfg = synth.C5
    B = 65536
    bt = synth.make_mhe_batch(cfg, B, seed=1)
    members = [0, 4097, 20000, 65535]
    rows, bm = mhe_util.run_periods(cfg, bt, cfg.He + 2, members)
    _check(rows)
    info = bm.getinfo()
    assert np.abs(info["X̂"]).max() <= cfg.xabs + 1e-8 and np.abs(info["x̂arr"]).max() <= cfg.xabs + 1e-8
    frac_active = np.mean(np.abs(info["X̂"]).max(axis=1) >= cfg.xabs - 1e-6)
    assert frac_active > 0.05                               # the workload does exercise the constraints


def test_closed_loop_mhe_feeds_linmpc():
    """Both device paths in one loop, like `sim!` with a MovingHorizonEstimator inside a LinMPC
    (src/plot_sim.jl, controller/execute.jl:59-80): plant -> ym -> BatchMHE.preparestate -> BatchLinMPC.moveinput ->
    u -> BatchMHE.updatestate, against the same loop on oracle/mhe.py + oracle/condense.py, estimator by estimator."""
    from oracle import condense as cd, qp
    cfg = synth.MheConfig("loop", nx=3, nu=2, nym=2, nd=0, He=5, xabs=2.5)
    B, Hp, Hc = 6, 8, 3
    bt = synth.make_mhe_batch(cfg, B, seed=13)
    rng = np.random.default_rng(5)
    est = mhe_util.make_product(cfg, bt)
    ors = mhe_util.make_oracles(cfg, bt, range(B))
    mpc = mpcqp.BatchLinMPC(bt["Ahat"], bt["Bhu"], bt["Chm"], Hp=Hp, Hc=Hc, Mwt=[1.0, 1.0], Nwt=[0.1, 0.1], Cwt=1e5)
    mpc.setconstraint(umin=[-0.8, -0.8], umax=[0.8, 0.8], ymax=[1.2, 1.2])
    omp = []
    for b in range(B):
        m = cd.LinMPCOracle(bt["Ahat"][b], bt["Bhu"][b], bt["Chm"][b], Hp=Hp, Hc=Hc, Mwt=[1.0, 1.0], Nwt=[0.1, 0.1], Cwt=1e5)
        m.setconstraint(umin=[-0.8, -0.8], umax=[0.8, 0.8], ymax=[1.2, 1.2])
        omp.append(m)
    xg = 0.3 * rng.standard_normal((B, cfg.nx))          # plant states driven by the PRODUCT's inputs
    xo = xg.copy()                                       # ... and by the oracle loop's inputs
    ry = np.array([0.6, -0.4])
    lu_o = np.zeros((B, cfg.nu))
    worst_x = worst_u = 0.0
    for k in range(9):
        v = cfg.sigmaR * rng.standard_normal((B, cfg.nym))
        yg = np.einsum("bij,bj->bi", bt["C"], xg) + v
        yo = np.einsum("bij,bj->bi", bt["C"], xo) + v
        xhat = est.preparestate(yg)
        assert np.all(est.status == 0)
        ug = mpc.moveinput(xhat, ry)
        assert np.all(mpc.status == 0)
        est.updatestate(ug, yg)
        uo = np.zeros_like(ug)
        for b, (e, m) in enumerate(zip(ors, omp)):
            xh = e.preparestate(yo[b])
            m.initpred(xh, lu_o[b], ry)
            m.linconstraint()
            z, st, _ = qp.solve_qp(*m.qp_data(), m.warmstart(), return_info=True)
            m.Zt = z
            uo[b] = z[:cfg.nu] + lu_o[b]
            e.updatestate(uo[b], yo[b])
            worst_x = max(worst_x, np.abs(xhat[b] - xh).max() / max(1.0, np.abs(xh).max()))
        worst_u = max(worst_u, np.abs(ug - uo).max())
        lu_o = uo.copy()
        w = cfg.sigmaQ * rng.standard_normal((B, cfg.nx))
        xg = np.einsum("bij,bj->bi", bt["A"], xg) + np.einsum("bij,bj->bi", bt["Bu"], ug) + w
        xo = np.einsum("bij,bj->bi", bt["A"], xo) + np.einsum("bij,bj->bi", bt["Bu"], uo) + w
    assert worst_x <= 1e-5 and worst_u <= 1e-5, (worst_x, worst_u)


@pytest.mark.parametrize("kw, soft", [
    (dict(nx=2, nu=2, nym=2, nd=0, He=5, xabs=0.5, vabs=0.2, Cwt=1e3),
     dict(c_xhatmax=[1.0, 0.5, 0.0, 0.0], c_xhatmin=[0.0, 1.0, 0.0, 0.0], c_vhatmin=[1.0, 1.0], c_vhatmax=[0.5, 1.0])),
    (dict(nx=4, nu=2, nym=3, nd=1, He=8, xabs=0.8, wabs=0.15, Cwt=1e4),
     dict(c_xhatmin=[1.0] * 7, c_xhatmax=[1.0] * 7, c_whatmin=[0.5] * 7, c_whatmax=[0.0] * 7)),
    (dict(nx=8, nu=4, nym=4, nd=0, He=10, xabs=0.6, Cwt=1e5, direct=False), dict(c_xhatmin=[1.0] * 12, c_xhatmax=[1.0] * 12)),
], ids=["xhat+vhat", "xhat+what+d", "NX12 predictor"])
def test_soft_constraints_match_oracle(kw, soft):
    """Finite Cwt and softness parameters: the slack ε, the estimates and Ŵ equal the oracle's (which solves the
    reference's condensed QP with ε first in Z̃)."""
    cfg = synth.MheConfig("soft", **kw)
    B = 9
    bt = synth.make_mhe_batch(cfg, B, seed=17)
    bounds = mhe_util.bounds_of(cfg)
    bounds.update({k: np.asarray(v, float) for k, v in soft.items()})
    rows, bm = mhe_util.run_periods(cfg, bt, cfg.He + 2, [0, 4, 8], bounds=bounds)
    for r in rows:
        assert all(s == 0 for s in r["ostatus"]) and np.all(r["status"] == 0), r
        assert r["ex"] <= TOL and r["ew"] <= TOL and r["ee"] <= TOL * max(1.0, r["eps"].max()), r
    assert max(r["eps"].max() for r in rows) > 1e-3


def test_reference_known_answers_through_the_product():
    """test/2_test_state_estim.jl:1034-1075 driven through BatchMHE on the GPU (both forms)."""
    for direct, r in mhe_util.reference_known_answers(B=3).items():
        assert r["x_at_op"] <= 1e-9 and r["y_hold"] <= r["tol"] and r["y_step"] <= r["tol"], (direct, r)


@pytest.mark.parametrize("seed", list(range(16)) + [227])
def test_randomised_families_match_oracle(seed):
    """Random dimensions, forms, horizons, bound classes, hard / soft (tests/mhe_util.random_family); a 900-family sweep
    is recorded in profiles/r2c/family_sweeps.txt (seed 227 is its worst case, 3e-6: a weakly active bound)."""
    worst, ncmp, nfail = mhe_util.random_family(seed)
    assert ncmp > 0 and worst <= 1e-5, (worst, ncmp, nfail)         # north-star tolerance


@pytest.mark.parametrize("soft", [True, False], ids=["soft", "hard"])
def test_reference_constraint_violation_through_the_product(soft):
    """"MHE constraint violation (LinModel)", test/2_test_state_estim.jl:1491-1539, through BatchMHE on the GPU:
    x̂, Ŵ, V̂ sit on the violated bound (atol 5e-2 in the reference) and agree with the oracle's answers."""
    got = mhe_util.reference_constraint_violation(soft, B=3)
    ref = mhe_util.reference_constraint_violation(soft, oracle=True)
    for k in ref:
        assert got[k] <= 5e-2, (k, got[k])
        assert abs(got[k] - ref[k]) <= 1e-5, (k, got[k], ref[k])


@pytest.mark.parametrize("direct", [True, False])
def test_reference_unfilled_window_through_the_product(direct):
    """"MHE estimation with unfilled window", test/2_test_state_estim.jl:1313-1337 (atol 1e-6 there)."""
    assert mhe_util.reference_unfilled_window(direct, B=3) <= 1e-6


@pytest.mark.parametrize("soft", [False, True], ids=["hard", "soft"])
def test_window_long_bounds(soft):
    """setconstraint!(estim; X̂min, ..., V̂max) (construct.jl:858-935): a bound per channel and stage; a window that is
    not full uses the last Nk blocks."""
    ex, ew, active = mhe_util.window_long_bounds(B=6, soft=soft, nper=10)
    assert active > 0
    assert ex <= TOL and ew <= TOL, (ex, ew)


def test_window_long_softness():
    """setconstraint!(estim; C_x̂min, ..., C_v̂max) (construct.jl:937-1020): a softness per channel and stage, zero (hard) on
    some rows; the softness column of the constraint matrices is not truncated while the window grows."""
    eps = []
    ex, ew, active = mhe_util.window_long_bounds(B=6, nper=10, csoft=True, eps_seen=eps)
    assert active > 0 and max(eps) > 1e-6
    assert ex <= TOL and ew <= TOL, (ex, ew)


def test_reference_setmodel_through_the_product():
    """setmodel!(::MovingHorizonEstimator, model), test/2_test_state_estim.jl:1668-1718."""
    for k, (v, want) in mhe_util.reference_setmodel(B=3).items():
        assert abs(v - want) <= 1e-3 * max(1.0, abs(want)), (k, v, want)
