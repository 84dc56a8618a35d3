"""CPU ORACLE (test infrastructure, NOT a product path) -- linear MovingHorizonEstimator
(BASELINE configs[4]) and the time-varying KalmanFilter it is pinned on.

Dense NumPy restatement, "the reference's way" (condensed decision vector Z̃ = [ε; x̂0(k-Nk+p); Ŵ],
SingleShooting, LinModel), of
  * init_predmat_mhe                      src/estimator/mhe/transcription.jl:151-295   (E, G, J, B, ex̄, EX̂, GX̂, JX̂, BX̂)
  * relaxarrival / relaxX̂ / relaxŴ / relaxV̂  src/estimator/mhe/construct.jl:1151-1288   (slack FIRST in Z̃)
  * add_data_windows!, trunc_*            src/estimator/mhe/execute.jl:497-548, 560-574 (growing window Nk < He)
  * initpred!  (F, fx̄, H̃, q̃)               src/estimator/mhe/execute.jl:419-457
  * linconstraint! (FX̂, b)                src/estimator/mhe/transcription.jl:732-782
  * optim_objective! -> the QP optimum     src/estimator/mhe/execute.jl:576-618 (oracle/qp.py instead of JuMP/OSQP)
  * getstate!, predict_mhe!               src/estimator/mhe/execute.jl:629-643, transcription.jl:1111-1133
  * correct_cov! / update_cov!            src/estimator/mhe/execute.jl:727-781 (default covestim = KalmanFilter,
                                          construct.jl:642-649)
  * KalmanFilter correct / predict        src/estimator/kalman.jl:1235-1264, 1275-1290, update_estimate! :520-525

Pin (tests/test_oracle_mhe.py): the reference's own "MHE v.s. Kalman filters" test
(test/2_test_state_estim.jl:1750-1777: He = 3, both forms, 6 periods, atol 1e-6) and the
steady-state answers of "MHE estimation and getinfo (LinModel)" (:1034-1075).
"""
from __future__ import annotations

import numpy as np

from . import estim as es
from . import qp


class KalmanFilterOracle(es.SteadyKalmanFilterOracle):
    """KalmanFilter(model; nint_ym, σP_0, σQ, σR, direct): time-varying gain and covariance."""

    def correct(self, y0m, d0):
        """correct_estimate_kf! -- kalman.jl:1235-1264."""
        Cm = self.Chm
        M = Cm @ self.P @ Cm.T + self.R
        K = self.P @ Cm.T @ np.linalg.inv(M)
        self.x0 = self.x0 + K @ (y0m - Cm @ self.x0 - self.Dhdm @ d0)
        self.P = (np.eye(self.nxh) - K @ Cm) @ self.P
        self.P = 0.5 * (self.P + self.P.T)

    def predict(self, u0, d0):
        """predict_estimate_kf! -- kalman.jl:1275-1290."""
        self.x0 = self.Ah @ self.x0 + self.Bhu @ u0 + self.Bhd @ d0 + self.fhop - self.xhop
        self.P = self.Ah @ self.P @ self.Ah.T + self.Q

    def preparestate(self, ym, d=()):
        if self.direct:
            self.correct(np.asarray(ym, float) - self.model.yop[self.i_ym], np.asarray(d, float) - self.model.dop)
        self.prepared = True
        return self.x0 + self.xhop

    def updatestate(self, u, ym=None, d=()):
        u0, d0 = np.asarray(u, float) - self.model.uop, np.asarray(d, float) - self.model.dop
        if not self.direct:
            self.correct(np.asarray(ym, float) - self.model.yop[self.i_ym], d0)
        self.predict(u0, d0)
        return self.x0 + self.xhop


def _attach_cov(kf, model, sigmaQ, sigmaR, sigmaQint_ym, sigmaQint_u=None, nint_u_tot=0):
    """Q̂, R̂ of the augmented model (the SteadyKalmanFilterOracle keeps only the gain); augmented states in the
    order [x; integrators on u; integrators on ym] (augment_model, src/estimator/construct.jl)."""
    nint_y = kf.nxh - model.nx - nint_u_tot
    sQ = np.full(model.nx, 1.0 / model.nx) if sigmaQ is None else np.asarray(sigmaQ, float)
    sR = np.ones(len(kf.i_ym)) if sigmaR is None else np.asarray(sigmaR, float)
    sQu = np.ones(nint_u_tot) if sigmaQint_u is None else np.asarray(sigmaQint_u, float)
    sQy = np.ones(nint_y) if sigmaQint_ym is None else np.asarray(sigmaQint_ym, float)
    kf.Q = np.diag(np.concatenate([sQ, sQu, sQy]) ** 2)
    kf.R = np.diag(sR ** 2)


def make_kalman_filter(model, direct=True, sigmaQ=None, sigmaR=None, sigmaQint_ym=None, sigmaP_0=None,
                       sigmaPint_ym_0=None, nint_ym=None, P_0=None, nint_u=0, sigmaQint_u=None, sigmaPint_u_0=None):
    kf = KalmanFilterOracle.__new__(KalmanFilterOracle)
    es.SteadyKalmanFilterOracle.__init__(kf, model, sigmaQ=sigmaQ, sigmaR=sigmaR, nint_u=nint_u, nint_ym=nint_ym,
                                         sigmaQint_u=sigmaQint_u, sigmaQint_ym=sigmaQint_ym)
    nu_int = int(np.sum(nint_u))
    _attach_cov(kf, model, sigmaQ, sigmaR, sigmaQint_ym, sigmaQint_u, nu_int)
    kf.direct = direct
    nint = kf.nxh - model.nx - nu_int
    sP = np.full(model.nx, 1.0 / model.nx) if sigmaP_0 is None else np.asarray(sigmaP_0, float)
    sPu = np.ones(nu_int) if sigmaPint_u_0 is None else np.asarray(sigmaPint_u_0, float)
    sPy = np.ones(nint) if sigmaPint_ym_0 is None else np.asarray(sigmaPint_ym_0, float)
    kf.P0 = np.diag((sP if len(sP) == kf.nxh else np.concatenate([sP, sPu, sPy])) ** 2)
    if P_0 is not None:                      # (setstate!(estim, x̂, P̂): a full matrix)
        kf.P0 = np.asarray(P_0, float).copy()
    kf.P = kf.P0.copy()
    kf.prepared = False
    return kf


def init_predmat_mhe(Ah, Bhu, Chm, Bhd, Dhdm, xhop, fhop, He, direct):
    """E, G, J, B, ex̄, EX̂, GX̂, JX̂, BX̂ -- transcription.jl:151-295 (SingleShooting)."""
    nx, nu, nd, nym = Ah.shape[0], Bhu.shape[1], Bhd.shape[1], Chm.shape[0]
    p = 0 if direct else 1
    Apow = [np.eye(nx)]
    for _ in range(He):
        Apow.append(Apow[-1] @ Ah)
    nCApow = [-Chm @ P for P in Apow]                       # -Ĉm Â^j, j = 0..He
    stack = lambda blocks: np.vstack(blocks) if blocks else np.zeros((0, 0))
    nCA = stack(nCApow)
    E = np.zeros((nym * He, nx + nx * He))
    i = 0
    for j in (range(1, He + 1) if p == 0 else range(0, He)):
        rows = slice(i * nym, nym * He)
        E[rows, j * nx:(j + 1) * nx] = nCA[:nym * He - i * nym]
        i += 1
    if p == 0:
        E[:, :nx] = nCA[nym:]
    exbar = np.hstack([-np.eye(nx), np.zeros((nx, nx * He))])
    Apv = stack(Apow)
    EX = np.zeros((nx * He, nx + nx * He))
    for i, j in enumerate(range(1, He + 1)):
        EX[i * nx:, j * nx:(j + 1) * nx] = Apv[:nx * He - i * nx]
    EX[:, :nx] = Apv[nx:]
    nCAB = np.vstack([np.zeros((nym, nu))] + [nCApow[j] @ Bhu for j in range(He)])
    G = np.zeros((nym * He, nu * He))
    i = 0
    for j in (range(1, He) if p == 0 else range(0, He - 1)):
        G[i * nym:, j * nu:(j + 1) * nu] = nCAB[:nym * He - i * nym]
        i += 1
    if p == 0:
        G[:, :nu] = nCAB[nym:]
    AB = stack([Apow[j] @ Bhu for j in range(He)])
    GX = np.zeros((nx * He, nu * He))
    for j in range(He):
        GX[j * nx:, j * nu:(j + 1) * nu] = AB[:nx * He - j * nx]
    nCABd = np.vstack([-Dhdm] + [nCApow[j] @ Bhd for j in range(He)])
    J = np.zeros((nym * He, nd * (He + 1)))
    i = 0
    for j in range(1, He + 1):
        J[i * nym:, j * nd:(j + 1) * nd] = nCABd[:nym * He - i * nym]
        i += 1
    if p == 0:
        J[:, :nd] = nCABd[nym:]
    ABd = stack([Apow[j] @ Bhd for j in range(He)]) if nd else np.zeros((nx * He, 0))
    JX = np.zeros((nx * He, nd * (He + 1)))
    for j in range(He):
        JX[j * nx:, (j + p) * nd:(j + p + 1) * nd] = ABd[:nx * He - j * nx]
    S = np.cumsum(np.array(Apow), axis=0)                   # S(j) = sum_{i<=j} Â^i
    fx = fhop - xhop
    coefB = np.zeros((nym * He, nx))
    for jj, i in enumerate(range(0, He) if p == 0 else range(1, He)):
        coefB[i * nym:(i + 1) * nym] = -Chm @ S[jj]
    Bv = coefB @ fx
    BX = np.vstack([S[j] for j in range(He)]) @ fx
    return E, G, J, Bv, exbar, EX, GX, JX, BX


class MHEOracle:
    """MovingHorizonEstimator(model; He, nint_ym, σP_0, σQ, σR, Cwt, direct) for a LinModel,
    SingleShooting, default arrival covariance estimator (KalmanFilter with the same covariances)."""

    def __init__(self, model, He, direct=True, Cwt=np.inf, sigmaQ=None, sigmaR=None, sigmaQint_ym=None,
                 sigmaP_0=None, sigmaPint_ym_0=None, nint_ym=None, P_0=None, nint_u=0, sigmaQint_u=None,
                 sigmaPint_u_0=None):
        self.model, self.He, self.direct, self.Cwt = model, int(He), direct, float(Cwt)
        self.neps = 0 if np.isinf(Cwt) else 1
        kw = dict(sigmaQ=sigmaQ, sigmaR=sigmaR, sigmaQint_ym=sigmaQint_ym, sigmaP_0=sigmaP_0,
                  sigmaPint_ym_0=sigmaPint_ym_0, nint_ym=nint_ym, P_0=P_0, nint_u=nint_u, sigmaQint_u=sigmaQint_u,
                  sigmaPint_u_0=sigmaPint_u_0)
        self.cov = make_kalman_filter(model, direct=direct, **kw)          # covestim
        c = self.cov
        self.Ah, self.Bhu, self.Ch, self.Bhd, self.Dhd = c.Ah, c.Bhu, c.Ch, c.Bhd, c.Dhd
        self.Chm, self.Dhdm, self.xhop, self.fhop, self.i_ym = c.Chm, c.Dhdm, c.xhop, c.fhop, c.i_ym
        self.nxh, self.nym, self.nu, self.nd = c.nxh, len(c.i_ym), model.nu, model.nd
        self.Q, self.R = c.Q, c.R
        self.invQ, self.invR = np.linalg.inv(c.Q), np.linalg.inv(c.R)
        (self.E, self.G, self.J, self.B, self.exbar, self.EX, self.GX, self.JX, self.BX) = init_predmat_mhe(
            self.Ah, self.Bhu, self.Chm, self.Bhd, self.Dhdm, self.xhop, self.fhop, self.He, direct)
        nx, He = self.nxh, self.He
        inf = np.inf
        self.con = dict(x0min=np.full(nx, -inf), x0max=np.full(nx, inf),            # arrival
                        X0min=np.full(nx * He, -inf), X0max=np.full(nx * He, inf),
                        Wmin=np.full(nx * He, -inf), Wmax=np.full(nx * He, inf),
                        Vmin=np.full(self.nym * He, -inf), Vmax=np.full(self.nym * He, inf))
        z = np.zeros
        self.soft = dict(c_x0min=z(nx), c_x0max=z(nx), C_xmin=z(nx * He), C_xmax=z(nx * He), C_wmin=z(nx * He),
                         C_wmax=z(nx * He), C_vmin=z(self.nym * He), C_vmax=z(self.nym * He))
        self.reset()

    def reset(self):
        """init_estimate_cov! -- execute.jl:2-36."""
        nx, nu, nd, nym, He = self.nxh, self.nu, self.nd, self.nym, self.He
        self.Zt = np.zeros(self.neps + nx + nx * He)
        self.Y0m = np.full(nym * He, np.nan)
        self.U0 = np.full(nu * He, np.nan)
        self.D0 = np.full(nd * (He + 1), np.nan)
        self.D0[:nd] = 0.0                       # d0(-1) (construct.jl:211)
        self.X0_old = np.full(nx * He, np.nan)
        self.Nk = 0
        self.x0 = np.zeros(nx)
        self.lastu0 = np.zeros(nu)
        self.Parr_old = self.cov.P0.copy()
        self.invPbar = np.linalg.inv(self.Parr_old)
        self.x0arr_old = np.zeros(nx)
        self.prepared = False

    def initstate(self, u, ym, d=()):
        """initstate! -- execute.jl:208-220 + init_estimate_cov! (mhe/execute.jl:2-36)."""
        x = es.SteadyKalmanFilterOracle.initstate(self.cov, u, ym, d)
        x0 = self.cov.x0.copy()
        self.reset()
        self.x0 = x0
        u0 = np.asarray(u, float) - self.model.uop
        d0 = np.asarray(d, float) - self.model.dop
        if self.nd:
            self.D0[:self.nd] = d0
        self.lastu0 = u0.copy()
        return x

    def setmodel(self, model):
        """setmodel!(estim, model) -- src/estimator/execute.jl:483-497 + setmodel_estimator! (mhe/execute.jl:943-1046):
        new LinModel (same dimensions), augmented matrices and prediction matrices rebuilt, every deviation variable
        (windows, lastu0, x̂0, x̂0arr, x̂ bounds) moved from the old to the new operating points."""
        old, c = self.model, self.cov
        uop_old, yop_old, dop_old, xhop_old = old.uop.copy(), old.yop.copy(), old.dop.copy(), self.xhop.copy()
        kf = make_kalman_filter(model, direct=self.direct, nint_ym=[0] * self.nym if self.nxh == model.nx else None)
        assert kf.nxh == self.nxh, "setmodel: the stochastic model (integrators) must stay the same"
        kf.Q, kf.R, kf.P0 = c.Q, c.R, c.P0
        self.model, self.cov = model, kf
        self.Ah, self.Bhu, self.Ch, self.Bhd, self.Dhd = kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd
        self.Chm, self.Dhdm, self.xhop, self.fhop = kf.Chm, kf.Dhdm, kf.xhop, kf.fhop
        (self.E, self.G, self.J, self.B, self.exbar, self.EX, self.GX, self.JX, self.BX) = init_predmat_mhe(
            self.Ah, self.Bhu, self.Chm, self.Bhd, self.Dhdm, self.xhop, self.fhop, self.He, self.direct)
        nx, nu, nd, nym, He = self.nxh, self.nu, self.nd, self.nym, self.He
        dx = xhop_old - self.xhop
        self.x0 = self.x0 + dx
        for k in ("x0min", "x0max"):
            self.con[k] = self.con[k] + dx
        for k in ("X0min", "X0max"):
            self.con[k] = self.con[k] + np.tile(dx, He)
        self.Y0m = self.Y0m + np.tile(yop_old[self.i_ym] - model.yop[self.i_ym], He)
        self.U0 = self.U0 + np.tile(uop_old - model.uop, He)
        if nd:
            self.D0 = self.D0 + np.tile(dop_old - model.dop, He + 1)
        self.X0_old = self.X0_old + np.tile(dx, He)
        self.lastu0 = self.lastu0 + (uop_old - model.uop)
        self.Zt[self.neps:self.neps + nx] += dx
        self.x0arr_old = self.x0arr_old + dx
        return self

    def setstate(self, xhat):
        self.x0 = np.asarray(xhat, float) - self.xhop
        return self

    def setconstraint(self, xhatmin=None, xhatmax=None, whatmin=None, whatmax=None, vhatmin=None, vhatmax=None,
                      c_xhatmin=None, c_xhatmax=None, c_whatmin=None, c_whatmax=None, c_vhatmin=None, c_vhatmax=None,
                      Xhatmin=None, Xhatmax=None, Whatmin=None, Whatmax=None, Vhatmin=None, Vhatmax=None,
                      C_xhatmin=None, C_xhatmax=None, C_whatmin=None, C_whatmax=None, C_vhatmin=None, C_vhatmax=None):
        """setconstraint! -- construct.jl:858-1049 (per-channel bounds repeated over the window; the
        arrival state takes the same x̂ bounds)."""
        He, nx = self.He, self.nxh
        if xhatmin is not None:
            v = np.asarray(xhatmin, float) - self.xhop
            self.con["x0min"], self.con["X0min"] = v.copy(), np.tile(v, He)
        if xhatmax is not None:
            v = np.asarray(xhatmax, float) - self.xhop
            self.con["x0max"], self.con["X0max"] = v.copy(), np.tile(v, He)
        for key, v in (("Wmin", whatmin), ("Wmax", whatmax), ("Vmin", vhatmin), ("Vmax", vhatmax)):
            if v is not None:
                self.con[key] = np.tile(np.asarray(v, float), He)
        # window-long vectors (construct.jl:890-935): X̂ = [arrival; the He window states], Ŵ and V̂ He blocks
        for k0, k1, v in (("x0min", "X0min", Xhatmin), ("x0max", "X0max", Xhatmax)):
            if v is not None:
                v = np.asarray(v, float)
                assert v.shape == (nx * (He + 1),)
                self.con[k0], self.con[k1] = v[:nx] - self.xhop, v[nx:] - np.tile(self.xhop, He)
        for key, v, n in (("Wmin", Whatmin, nx), ("Wmax", Whatmax, nx), ("Vmin", Vhatmin, self.nym), ("Vmax", Vhatmax, self.nym)):
            if v is not None:
                v = np.asarray(v, float)
                assert v.shape == (n * He,)
                self.con[key] = v.copy()
        for k0, k1, v in (("c_x0min", "C_xmin", c_xhatmin), ("c_x0max", "C_xmax", c_xhatmax)):
            if v is not None:
                self.soft[k0], self.soft[k1] = np.asarray(v, float).copy(), np.tile(np.asarray(v, float), He)
        for key, v in (("C_wmin", c_whatmin), ("C_wmax", c_whatmax), ("C_vmin", c_vhatmin), ("C_vmax", c_vhatmax)):
            if v is not None:
                self.soft[key] = np.tile(np.asarray(v, float), He)
        # window-long softness (construct.jl:964-1000): C_x̂ = [arrival; He window states], C_ŵ and C_v̂ He blocks
        for k0, k1, v in (("c_x0min", "C_xmin", C_xhatmin), ("c_x0max", "C_xmax", C_xhatmax)):
            if v is not None:
                v = np.asarray(v, float)
                assert v.shape == (nx * (He + 1),)
                if np.any(v < 0):
                    raise ValueError(f"{k1} weights should be non-negative")
                self.soft[k0], self.soft[k1] = v[:nx].copy(), v[nx:].copy()
        for key, v, n in (("C_wmin", C_whatmin, nx), ("C_wmax", C_whatmax, nx), ("C_vmin", C_vhatmin, self.nym), ("C_vmax", C_vhatmax, self.nym)):
            if v is not None:
                v = np.asarray(v, float)
                assert v.shape == (n * He,)
                if np.any(v < 0):
                    raise ValueError(f"{key} weights should be non-negative")
                self.soft[key] = v.copy()
        if any(np.any(self.soft[k] != 0) for k in self.soft) and not self.neps:
            raise ValueError("Slack variable weight Cwt must be finite to set softness parameters")
        return self

    # ---- data windows ---------------------------------------------------------------------------
    def _add_data(self, y0m, d0, u0):
        """add_data_windows! -- execute.jl:497-548."""
        nx, nu, nd, nym, He = self.nxh, self.nu, self.nd, self.nym, self.He
        x0_old = self.x0.copy()
        self.Nk += 1
        Nk = self.Nk
        moving = Nk > He
        if moving:
            self.Y0m[:-nym] = self.Y0m[nym:]; self.Y0m[-nym:] = y0m
            if nd:
                self.D0[:-nd] = self.D0[nd:]; self.D0[-nd:] = d0
            self.U0[:-nu] = self.U0[nu:]; self.U0[-nu:] = u0
            self.X0_old[:-nx] = self.X0_old[nx:]; self.X0_old[-nx:] = x0_old
            self.Nk = He
        else:
            self.Y0m[nym * (Nk - 1):nym * Nk] = y0m
            if nd:
                self.D0[nd * Nk:nd * (Nk + 1)] = d0
            self.U0[nu * (Nk - 1):nu * Nk] = u0
            self.X0_old[nx * (Nk - 1):nx * Nk] = x0_old
        self.x0arr_old = self.X0_old[:nx].copy()
        return moving

    # ---- the QP of one period ---------------------------------------------------------------------
    def qp_data(self):
        """initpred! (execute.jl:419-457) + linconstraint! (transcription.jl:732-782) for the current
        window: H̃, q̃, A, b of  min 1/2 Z̃'H̃Z̃ + q̃'Z̃  s.t.  A Z̃ <= b  (finite rows), Z̃ = [ε; x̂0arr; Ŵ(1..Nk)]."""
        nx, nu, nd, nym, He, Nk, ne = self.nxh, self.nu, self.nd, self.nym, self.He, self.Nk, self.neps
        nYm, nX, nZ = nym * Nk, nx * Nk, nx + nx * Nk
        E, G, J, B = self.E[:nYm, :nZ], self.G[:nYm, :nu * Nk], self.J[:nYm, :nd * (Nk + 1)], self.B[:nYm]
        EX, GX, JX, BX = self.EX[:nX, :nZ], self.GX[:nX, :nu * Nk], self.JX[:nX, :nd * (Nk + 1)], self.BX[:nX]
        exb = self.exbar[:, :nZ]
        U0, D0, Y0m = self.U0[:nu * Nk], self.D0[:nd * (Nk + 1)], self.Y0m[:nYm]
        F = Y0m + B + G @ U0 + (J @ D0 if nd else 0.0)
        fxb = self.x0arr_old
        EZ = np.vstack([exb, E])
        FZ = np.concatenate([fxb, F])
        M = np.zeros((nx + nYm, nx + nYm))
        M[:nx, :nx] = self.invPbar
        M[nx:, nx:] = np.kron(np.eye(Nk), self.invR)
        Tw = np.hstack([np.zeros((nx * Nk, nx)), np.eye(nx * Nk)])            # Ŵ = Tŵ Z (init_ZtoŴ)
        N = Tw.T @ np.kron(np.eye(Nk), self.invQ) @ Tw
        H = 2.0 * (EZ.T @ M @ EZ + N)
        q = 2.0 * (M @ EZ).T @ FZ
        r = FZ @ M @ FZ
        if ne:                              # slack FIRST (relax*, construct.jl:1172-1288)
            H = np.block([[np.array([[2.0 * self.Cwt]]), np.zeros((1, nZ))], [np.zeros((nZ, 1)), H]])
            q = np.concatenate([[0.0], q])
        FX = BX + GX @ U0 + (JX @ D0 if nd else 0.0)
        tr = lambda v, n: v[len(v) - n * Nk:]                        # trunc_bounds: the LAST Nk blocks
        hd = lambda v, n: v[:n * Nk]        # softness = a column of A_X̂min ... A_V̂max, which is NOT truncated: the FIRST Nk blocks
        X0min, X0max = tr(self.con["X0min"], nx), tr(self.con["X0max"], nx)
        Wmin, Wmax = tr(self.con["Wmin"], nx), tr(self.con["Wmax"], nx)
        Vmin, Vmax = tr(self.con["Vmin"], nym), tr(self.con["Vmax"], nym)
        s = self.soft
        col = lambda c: c.reshape(-1, 1)
        ex = -exb                                                     # x̂0arr = ex̂ Z
        blocks = [(-ex, -self.con["x0min"], s["c_x0min"]), (ex, self.con["x0max"], s["c_x0max"]),
                  (-EX, -X0min + FX, hd(s["C_xmin"], nx)), (EX, X0max - FX, hd(s["C_xmax"], nx)),
                  (-Tw, -Wmin, hd(s["C_wmin"], nx)), (Tw, Wmax, hd(s["C_wmax"], nx)),
                  (-E, -Vmin + F, hd(s["C_vmin"], nym)), (E, Vmax - F, hd(s["C_vmax"], nym))]
        A = np.vstack([np.hstack([-col(c), a]) if ne else a for a, _, c in blocks])
        b = np.concatenate([bb for _, bb, _ in blocks])
        fin = np.isfinite(b)
        zmin = np.full(ne + nZ, -np.inf)
        zmax = np.full(ne + nZ, np.inf)
        if ne:
            zmin[0] = 0.0
        self._pred = (E, F, EX, FX, r)
        return H, q, A[fin], b[fin], zmin, zmax

    def _solve(self):
        """optim_objective! + getstate! -- execute.jl:576-643."""
        H, q, A, b, zmin, zmax = self.qp_data()
        nx, Nk, ne = self.nxh, self.Nk, self.neps
        nZt = ne + nx + nx * Nk
        z0 = np.zeros(nZt)
        z, st, info = qp.solve_qp(H, q, A, b, zmin, zmax, z0, return_info=True)
        self.status, self.info = st, info
        self.Zt[:] = 0.0
        self.Zt[:nZt] = z
        E, F, EX, FX, r = self._pred
        Z = z[ne:]
        self.Vhat = E @ Z + F
        self.X0 = EX @ Z + FX
        self.x0 = self.X0[(Nk - 1) * nx:Nk * nx].copy()
        self.x0arr = Z[:nx].copy()
        self.Jopt = 0.5 * z @ H @ z + q @ z + r

    def _cov_step(self, correct_only):
        """correct_cov! / update_cov! -- execute.jl:727-781: the arrival covariance through the
        KalmanFilter's covariance recursion on the oldest data of the window."""
        c = self.cov
        nu, nd, nym = self.nu, self.nd, self.nym
        c.x0 = self.x0arr_old.copy()
        c.P = self.Parr_old.copy()
        y0, d0 = self.Y0m[:nym], self.D0[:nd]
        if correct_only:
            c.correct(y0, d0)
        else:
            if not c.direct:
                c.correct(y0, d0)
            c.predict(self.U0[:nu], d0)
        self.Parr_old = c.P.copy()
        self.invPbar = np.linalg.inv(self.Parr_old)

    # ---- public steps -------------------------------------------------------------------------------
    def preparestate(self, ym, d=()):
        """preparestate! -> correct_estimate! (execute.jl:44-57)."""
        y0m = np.asarray(ym, float) - self.model.yop[self.i_ym]
        d0 = np.asarray(d, float) - self.model.dop
        if self.direct:
            moving = self._add_data(y0m, d0, self.lastu0)
            if moving:
                self._cov_step(correct_only=True)
            self._solve()
        self.prepared = True
        return self.x0 + self.xhop

    def updatestate(self, u, ym, d=()):
        """updatestate! -> update_estimate! (execute.jl:76-88)."""
        u0 = np.asarray(u, float) - self.model.uop
        y0m = np.asarray(ym, float) - self.model.yop[self.i_ym]
        d0 = np.asarray(d, float) - self.model.dop
        if not self.direct:
            self._add_data(y0m, d0, u0)
            self._solve()
        if self.Nk == self.He:
            self._cov_step(correct_only=False)
        self.lastu0 = u0.copy()
        self.prepared = False
        return self.x0 + self.xhop

    def evaloutput(self, d=()):
        d0 = np.asarray(d, float) - self.model.dop
        return self.Ch @ self.x0 + self.Dhd @ d0 + self.model.yop
