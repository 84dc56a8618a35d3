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
