"""CPU ORACLE (test infrastructure, NOT a product path) -- the estimator/plant steps that sit on
both sides of the hot path in a closed loop.

Restates, in float64 NumPy/SciPy:
  * LinModel state update / output         src/model/linmodel.jl:284-301, src/sim_model.jl:239-290
  * SteadyKalmanFilter construction        src/estimator/kalman.jl:163-236 (gain from
    ControlSystemsBase.kalman -- third party, not in tree; empirically the *filter-form* gain
    P C'(C P C' + R)^-1 with P the predictor DARE solution, pinned by the doctest golden
    u = 17.577311 of ext/LinearMPCext.jl:255-269, see tests/test_oracle_known_answers.py::test_T8)
  * correct / predict                      src/estimator/kalman.jl:284-309
  * init_estimate!                         src/estimator/execute.jl:246-259
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import solve_discrete_are

from . import condense as cd


class LinModelOracle:
    """Discrete LinModel with operating points (src/model/linmodel.jl:1-66, sim_model.jl:101)."""

    def __init__(self, A, Bu, C, Bd=None, Dd=None, Ts=1.0):
        self.A, self.Bu, self.C = (np.atleast_2d(np.asarray(m, float)) for m in (A, Bu, C))
        self.nx, self.nu = self.Bu.shape
        self.ny = self.C.shape[0]
        self.Bd = np.zeros((self.nx, 0)) if Bd is None else np.atleast_2d(np.asarray(Bd, float))
        self.nd = self.Bd.shape[1]
        self.Dd = np.zeros((self.ny, self.nd)) if Dd is None else np.atleast_2d(np.asarray(Dd, float))
        self.Ts = Ts
        self.uop, self.yop, self.dop = np.zeros(self.nu), np.zeros(self.ny), np.zeros(self.nd)
        self.xop, self.fop = np.zeros(self.nx), np.zeros(self.nx)
        self.x0 = np.zeros(self.nx)

    def setop(self, uop=None, yop=None, dop=None, xop=None, fop=None):
        for name, v in (("uop", uop), ("yop", yop), ("dop", dop), ("xop", xop), ("fop", fop)):
            if v is not None:
                getattr(self, name)[:] = v
        return self

    def evaloutput(self, d=()):
        d0 = np.asarray(d, float) - self.dop
        return self.C @ self.x0 + self.Dd @ d0 + self.yop

    def updatestate(self, u, d=()):
        u0, d0 = np.asarray(u, float) - self.uop, np.asarray(d, float) - self.dop
        self.x0 = self.A @ self.x0 + self.Bu @ u0 + self.Bd @ d0 + self.fop - self.xop
        return self.x0 + self.xop


def tf1_zoh(gain, tau, Ts):
    """ZOH discretisation of gain/(tau s + 1) in the power-of-two balanced realisation the
    reference's LinModel(tf) produces for first-order systems:
    continuous (a, b, c) = (-1/tau, B, C) with B*C = gain/tau and B = 2^round(log2(sqrt(B*C)))."""
    k = gain / tau
    Bc = 2.0 ** np.round(np.log2(np.sqrt(abs(k))))
    Cc = k / Bc
    a = -1.0 / tau
    Ad = np.exp(a * Ts)
    Bd = (Ad - 1.0) / a * Bc
    return np.array([[Ad]]), np.array([[Bd]]), np.array([[Cc]])


class SteadyKalmanFilterOracle:
    """SteadyKalmanFilter(model; nint_u, nint_ym, σQ, σR, σQint_u, σQint_ym, direct=true)."""

    def __init__(self, model, i_ym=None, sigmaQ=None, sigmaR=None, nint_u=0, nint_ym=None,
                 sigmaQint_u=None, sigmaQint_ym=None):
        m = self.model = model
        i_ym = np.arange(m.ny) if i_ym is None else np.asarray(i_ym, int)
        self.i_ym = i_ym
        if nint_ym is None:
            # default_nint (src/estimator/construct.jl:365-376): 1 per measured output when the
            # augmented pair stays observable
            nint_ym = np.zeros(len(i_ym), int)
            for i in range(len(i_ym)):
                nint_ym[i] = 1
                As, Cs_u, Cs_y, _, _ = cd.init_estimstoch(m.nu, m.ny, i_ym, nint_u, nint_ym)
                Ah, _, Ch, *_ = cd.augment_model(m.A, m.Bu, m.C, m.Bd, m.Dd, m.xop, m.fop, As, Cs_u, Cs_y)
                if not _observable(Ah, Ch):
                    nint_ym[i] = 0
        As, Cs_u, Cs_y, nint_u, nint_ym = cd.init_estimstoch(m.nu, m.ny, i_ym, nint_u, nint_ym)
        (self.Ah, self.Bhu, self.Ch, self.Bhd, self.Dhd, self.xhop, self.fhop) = cd.augment_model(
            m.A, m.Bu, m.C, m.Bd, m.Dd, m.xop, m.fop, As, Cs_u, Cs_y)
        self.nxh = self.Ah.shape[0]
        self.Chm, self.Dhdm = self.Ch[i_ym], self.Dhd[i_ym]
        sQ = np.full(m.nx, 1.0 / m.nx) if sigmaQ is None else np.asarray(sigmaQ, float)
        sR = np.ones(len(i_ym)) if sigmaR is None else np.asarray(sigmaR, float)
        sQu = np.ones(int(np.sum(nint_u))) if sigmaQint_u is None else np.asarray(sigmaQint_u, float)
        sQy = np.ones(int(np.sum(nint_ym))) if sigmaQint_ym is None else np.asarray(sigmaQint_ym, float)
        Q = np.diag(np.concatenate([sQ, sQu, sQy]) ** 2)
        R = np.diag(sR ** 2)
        P = solve_discrete_are(self.Ah.T, self.Chm.T, Q, R)
        self.Khat = P @ self.Chm.T @ np.linalg.inv(self.Chm @ P @ self.Chm.T + R)
        self.x0 = np.zeros(self.nxh)

    def preparestate(self, ym, d=()):
        """correct_estimate_obsv! -- src/estimator/kalman.jl:284-295."""
        y0m = np.asarray(ym, float) - self.model.yop[self.i_ym]
        d0 = np.asarray(d, float) - self.model.dop
        self.x0 = self.x0 + self.Khat @ (y0m - self.Chm @ self.x0 - self.Dhdm @ d0)
        return self.x0 + self.xhop

    def updatestate(self, u, ym=None, d=()):
        """predict_estimate_obsv! -- src/estimator/kalman.jl:298-309."""
        u0 = np.asarray(u, float) - self.model.uop
        d0 = np.asarray(d, float) - self.model.dop
        self.x0 = self.Ah @ self.x0 + self.Bhu @ u0 + self.Bhd @ d0 + self.fhop - self.xhop
        return self.x0 + self.xhop

    def initstate(self, u, ym, d=()):
        """init_estimate! -- src/estimator/execute.jl:246-259: steady state consistent with (u, ym)."""
        u0 = np.asarray(u, float) - self.model.uop
        d0 = np.asarray(d, float) - self.model.dop
        y0m = np.asarray(ym, float) - self.model.yop[self.i_ym]
        rhs = np.concatenate([self.fhop - self.xhop + self.Bhu @ u0 + self.Bhd @ d0,
                              y0m - self.Dhdm @ d0])
        M = np.vstack([np.eye(self.nxh) - self.Ah, self.Chm])
        self.x0 = np.linalg.lstsq(M, rhs, rcond=None)[0]
        return self.x0 + self.xhop

    def evaloutput(self, d=()):
        d0 = np.asarray(d, float) - self.model.dop
        return self.Ch @ self.x0 + self.Dhd @ d0 + self.model.yop


def _observable(A, C):
    n = A.shape[0]
    O = np.vstack([C @ np.linalg.matrix_power(A, k) for k in range(n)])
    return np.linalg.matrix_rank(O) == n
