"""CPU ORACLE (test infrastructure, NOT a product path) -- LinMPC with the MultipleShooting transcription,
restated "the reference's way": dense matrices over the decision vector Z = [ΔU; X̂0(k+1..k+Hp)] (+ ϵ), the model as the
equality constraints A_eq Z̃ = b_eq.

    decision vector      src/controller/transcription.jl:5-7   (nZ = nu Hc + nx̂ Hp)
    init_predmat(MS)     src/controller/transcription.jl:196-240   E = [0 diag(Ĉ)], J = diag(D̂d), ex̂ = [0 I], rest 0
    init_defectmat       src/controller/transcription.jl:303-414   Ŝ = E_S Z + G_S d0 + J_S D̂0 + K_S x̂0 + V_S u0(k-1) + B_S
    init_matconstraint   src/controller/transcription.jl:667-703 (+ A_ŝ = [E_S 0] for the equalities, :705-728)
    linconstrainteq!     src/controller/transcription.jl:913-928   b_eq = -F_S
    warm start           src/controller/transcription.jl:1009-1075 (ΔU and X̂0 shifted; the kernel rolls X̂0 out from ΔU)

The weights, bounds, softness parameters, operating points and the `setconstraint!` rules are those of the SingleShooting
oracle (oracle/condense.py: LinMPCOracle), an instance of which is kept as `self.ss` -- the two transcriptions describe
the same optimisation problem, so `ΔU*` of both must agree (the reference's own test, test/3_test_predictive_control.jl:
120-127, 570-579 asserts the same answers from both).  The QP is solved in the null space of A_eq with an ORTHONORMAL
basis (QR of A_eq'), through oracle/qp.py's certified solver, and the KKT conditions of the equality-constrained QP are
checked in the full space afterwards (`info["kkt_full"]`); `solve_hp` does the same with oracle/qp_hp.py (60 digits) for
plants whose condensed Hessian is too ill-conditioned for float64 to adjudicate.
"""
from __future__ import annotations

import numpy as np

from . import condense as cd
from . import qp as qpo

INF = np.inf


def init_predmat_ms(Ch, Dhd, nxh, nu, Hp, Hc):
    """`init_predmat(::LinModel, ::MultipleShooting)` -- transcription.jl:218-240."""
    ny, nd = Ch.shape[0], Dhd.shape[1]
    E = np.hstack([np.zeros((Hp * ny, Hc * nu)), np.kron(np.eye(Hp), Ch)])
    ex = np.hstack([np.zeros((nxh, Hc * nu + (Hp - 1) * nxh)), np.eye(nxh)])
    J = np.kron(np.eye(Hp), Dhd)
    return E, J, ex


def init_defectmat(Ah, Bhu, Bhd, xhop, fhop, Hp, Hc, nb):
    """`init_defectmat(::LinModel, ::MultipleShooting)` -- transcription.jl:373-414."""
    nxh, nu = Bhu.shape
    nd = Bhd.shape[1]
    KS = np.vstack([Ah, np.zeros((nxh * (Hp - 1), nxh))])
    VS = np.tile(Bhu, (Hp, 1))
    ES = np.hstack([np.zeros((nxh * Hp, nu * Hc)), -np.eye(nxh * Hp)])
    for j in range(Hc):
        for i in range(j, Hc):
            r0 = nxh * int(np.sum(nb[:i]))
            for l in range(nb[i]):
                ES[r0 + l * nxh:r0 + (l + 1) * nxh, j * nu:(j + 1) * nu] = Bhu
    for j in range(1, Hp):
        ES[j * nxh:(j + 1) * nxh, nu * Hc + (j - 1) * nxh:nu * Hc + j * nxh] = Ah
    GS = np.vstack([Bhd, np.zeros((nxh * (Hp - 1), nd))])
    JS = np.zeros((nxh * Hp, nd * Hp))
    for j in range(1, Hp):
        JS[j * nxh:(j + 1) * nxh, (j - 1) * nd:j * nd] = Bhd
    BS = np.tile(fhop - xhop, Hp)
    return ES, GS, JS, KS, VS, BS


class LinMPCOracleMS:
    """One `LinMPC(...; transcription=MultipleShooting())` on the augmented model.  Same constructor, `setconstraint`,
    `initpred`, `linconstraint`, `moveinput`, `getinfo` as oracle/condense.py: LinMPCOracle."""

    def __init__(self, Ah, Bhu, Ch, Bhd=None, Dhd=None, **kw):
        if any(kw.get(k) is not None for k in ("Wy", "Wu", "Wd", "Wr")):
            raise NotImplementedError("custom linear constraints: SingleShooting oracle only")
        self.ss = ss = cd.LinMPCOracle(Ah, Bhu, Ch, Bhd, Dhd, **kw)
        nu, ny, nxh, nd, Hp, Hc = ss.nu, ss.ny, ss.nxh, ss.nd, ss.Hp, ss.Hc
        self.nu, self.ny, self.nxh, self.nd, self.Hp, self.Hc, self.nb = nu, ny, nxh, nd, Hp, Hc, ss.nb
        self.nDU, self.nX = nu * Hc, nxh * Hp
        self.nZ = self.nDU + self.nX
        self.neps = ss.neps
        self.nZt = self.nZ + self.neps
        self.PDu = cd.init_ZtoDU(nu, Hc, self.nZ)                       # [I 0]
        self.Pu, self.Tu = cd.init_ZtoU(nu, Hp, Hc, ss.nb, self.nZ)     # [Pu* 0]
        self.E, self.J, self.ex = init_predmat_ms(ss.Ch, ss.Dhd, nxh, nu, Hp, Hc)
        self.ES, self.GS, self.JS, self.KS, self.VS, self.BS = init_defectmat(
            ss.Ah, ss.Bhu, ss.Bhd, ss.xhop, ss.fhop, Hp, Hc, ss.nb)
        self.Zt = np.zeros(self.nZt)
        self.lastu0 = np.zeros(nu)
        self.solved_once = False
        self._rebuild()

    # ---- constraints (same relax* / i_b rules, matrices over the MultipleShooting Z) --------------
    def _rebuild(self):
        ss, ne = self.ss, self.neps
        self.A_Umin, self.A_Umax, self.Put = cd.relaxU(self.Pu, ss.C_umin, ss.C_umax, ne)
        self.A_DUmin, self.A_DUmax, self.PDut = cd.relaxDU(self.PDu, ss.C_dumin, ss.C_dumax, ne)
        self.A_Ymin, self.A_Ymax, self.Et = cd.relaxY(self.E, ss.C_ymin, ss.C_ymax, ne)
        self.A_xmin, self.A_xmax, self.ext = cd.relaxterminal(self.ex, ss.c_xmin, ss.c_xmax, ne)
        # init_boxconstraint_mpc (construct.jl:1209-1234): hard ΔU bounds are variable bounds, X̂0 is free, ϵ >= 0
        Zmin, Zmax = np.full(self.nZt, -INF), np.full(self.nZt, INF)
        nDU = self.nDU
        if ne:
            Zmin[-1] = 0.0
            hard_min, hard_max = self.A_DUmin[:, -1] == 0, self.A_DUmax[:, -1] == 0
            Zmin[:nDU][hard_min] = ss.DUmin[hard_min]
            Zmax[:nDU][hard_max] = ss.DUmax[hard_max]
        else:
            Zmin[:nDU], Zmax[:nDU] = ss.DUmin, ss.DUmax
        self.Zmin, self.Zmax = Zmin, Zmax
        self.A, self.i_b = cd.init_matconstraint(
            Zmin, Zmax, ss.U0min, ss.U0max, ss.DUmin, ss.DUmax, ss.Y0min, ss.Y0max, ss.x0min, ss.x0max,
            self.A_Umin, self.A_Umax, self.A_DUmin, self.A_DUmax, self.A_Ymin, self.A_Ymax, self.A_xmin, self.A_xmax)
        # A_ŝ = [E_S 0]  (transcription.jl:705-728: the slack does not relax the model)
        self.Aeq = np.hstack([self.ES, np.zeros((self.nX, ne))])
        self.Ht = cd.init_quadprog(self.Et, self.PDut, self.Put, ss.M_Hp, ss.Nt_Hc, ss.L_Hp)

    def setconstraint(self, **kw):
        self.ss.solved_once = self.solved_once
        self.ss.setconstraint(**kw)
        self._rebuild()
        return self

    # ---- per step ----------------------------------------------------------------------------------
    def initpred(self, xhat0, lastu, ry=None, d=None, Dhat=None, Rhaty=None, Rhatu=None):
        """`initpred!` (execute.jl:247-314) with the MultipleShooting prediction matrices: F = J D̂0 (K, V, G, B are 0)."""
        ss, Hp = self.ss, self.Hp
        self.lastu0 = np.asarray(lastu, float) - ss.uop
        self.Tu_lastu0 = self.Tu @ self.lastu0
        ry = ss.yop if ry is None else np.asarray(ry, float)
        Rhaty = np.tile(ry, Hp) if Rhaty is None else np.asarray(Rhaty, float)
        Rhatu = ss.Uop if Rhatu is None else np.asarray(Rhatu, float)
        F = np.zeros(self.ny * Hp)
        self.d0, self.D0 = np.zeros(self.nd), np.zeros(self.nd * Hp)
        if self.nd > 0:
            d = np.asarray(d, float)
            Dhat = np.tile(d, Hp) if Dhat is None else np.asarray(Dhat, float)
            self.d0, self.D0 = d - ss.dop, Dhat - ss.Dop
            F = F + self.J @ self.D0
        q = np.zeros(self.nZt)
        r = 0.0
        Cy = F + ss.Yop - Rhaty
        q += (ss.M_Hp @ self.Et).T @ Cy
        r += Cy @ ss.M_Hp @ Cy
        Cu = self.Tu_lastu0 + ss.Uop - Rhatu
        q += (ss.L_Hp @ self.Put).T @ Cu
        r += Cu @ ss.L_Hp @ Cu
        self.F, self.qt, self.r = F, 2.0 * q, r
        self.xhat0 = np.asarray(xhat0, float)
        return F, self.qt, r

    def linconstraint(self):
        """`linconstraint!` (transcription.jl:811-848: fx̂ = 0 here) and `linconstrainteq!` (:913-928)."""
        ss = self.ss
        fx = np.zeros(self.nxh)
        self.b = np.concatenate([
            -ss.U0min + self.Tu_lastu0, ss.U0max - self.Tu_lastu0, -ss.DUmin, ss.DUmax,
            -ss.Y0min + self.F, ss.Y0max - self.F, -ss.x0min + fx, ss.x0max - fx])
        FS = self.BS + self.KS @ self.xhat0 + self.VS @ self.lastu0
        if self.nd > 0:
            FS = FS + self.GS @ self.d0 + self.JS @ self.D0
        self.FS, self.beq = FS, -FS
        return self.b, self.beq

    def qp_data(self):
        """(H̃, q̃, A[i_b], b[i_b], Z̃min, Z̃max, A_eq, b_eq): what the reference hands to JuMP (linmpc.jl:323-345)."""
        return self.Ht, self.qt, self.A[self.i_b], self.b[self.i_b], self.Zmin, self.Zmax, self.Aeq, self.beq

    def rollout(self, DU):
        """X̂0(k+1..k+Hp) of the model for a given ΔU (the defect equations solved forward)."""
        ss = self.ss
        x, out = self.xhat0.copy(), []
        U0 = (self.Pu[:, :self.nDU] @ DU + self.Tu_lastu0).reshape(self.Hp, self.nu)
        for t in range(self.Hp):
            dt = (self.d0 if t == 0 else self.D0[(t - 1) * self.nd:t * self.nd]) if self.nd else np.zeros(0)
            x = ss.Ah @ x + ss.Bhu @ U0[t] + (ss.Bhd @ dt if self.nd else 0.0) + (ss.fhop - ss.xhop)
            out.append(x)
        return np.concatenate(out)

    def warmstart(self):
        """ΔU shifted like SingleShooting; X̂0 rolled out from it (a feasible point of the equalities; the reference
        shifts last period's X̂0, transcription.jl:1009-1075 -- a starting point, not part of the answer)."""
        nu, nDU = self.nu, self.nDU
        Zs = np.zeros(self.nZt)
        Zs[:nDU - nu] = self.Zt[nu:nDU]
        Zs[nDU:nDU + self.nX] = self.rollout(Zs[:nDU])
        if self.neps:
            Zs[-1] = self.Zt[-1]
        return Zs

    # ---- solve ---------------------------------------------------------------------------------------
    def _reduce(self):
        H, q, A, b, zmin, zmax, Aeq, beq = self.qp_data()
        G, h = qpo.stack_constraints(A, b, zmin, zmax)
        # Z = Zp + N w: Zp the minimum-norm solution of A_eq Z = b_eq, N an orthonormal null-space basis (QR of A_eq')
        Q, R = np.linalg.qr(Aeq.T, mode="complete")
        m = Aeq.shape[0]
        Zp = Q[:, :m] @ np.linalg.solve(R[:m, :m].T, beq)
        N = Q[:, m:]
        return (H, q, G, h, Aeq, beq), Zp, N

    def solve(self, return_info=False, hp=False, digits=60):
        """Optimum of the MultipleShooting QP.  Returns (Z̃, status[, info])."""
        (H, q, G, h, Aeq, beq), Zp, N = self._reduce()
        Hr, qr, Gr, hr = N.T @ H @ N, N.T @ (H @ Zp + q), G @ N, h - G @ Zp
        Hr = 0.5 * (Hr + Hr.T)
        n = N.shape[1]
        if hp:
            from . import qp_hp
            w, info = qp_hp.solve(Hr, qr, Gr, hr, digits=digits)
            st, lam = info["status"], info["lam"]
            info = dict(info, certificate="extended-precision")
        else:
            w, st, info = qpo.solve_qp(Hr, qr, Gr, hr, np.full(n, -INF), np.full(n, INF), None, return_info=True)
            lam = info["lam"]
        Z = Zp + N @ w
        # KKT of the equality-constrained QP in the full space: H Z + q + G'lam + A_eq'nu = 0 (nu by least squares)
        g = H @ Z + q + G.T @ lam
        nu_, *_ = np.linalg.lstsq(Aeq.T, -g, rcond=None)
        sc = 1.0 + max(np.abs(q).max(), np.abs(H @ Z).max(), np.abs(G.T @ lam).max(initial=0.0))
        info["kkt_full"] = {"stationarity": float(np.abs(g + Aeq.T @ nu_).max() / sc),
                            "defect": float(np.abs(Aeq @ Z - beq).max() / (1.0 + np.abs(Z).max())),
                            "primal": float(np.maximum(G @ Z - h, 0.0).max(initial=0.0) / (1.0 + np.abs(h).max(initial=0.0)))}
        info["cond_reduced"] = float(np.linalg.cond(Hr))
        return (Z, st, info) if return_info else (Z, st)

    def moveinput(self, xhat0, ry=None, d=None, *, lastu=None, Dhat=None, Rhaty=None, Rhatu=None, hp=False):
        lastu = self.lastu0 + self.ss.uop if lastu is None else lastu
        self.initpred(xhat0, lastu, ry, d, Dhat, Rhaty, Rhatu)
        self.linconstraint()
        Zs = self.warmstart()
        Z, st, info = self.solve(return_info=True, hp=hp)
        self.status, self.info = st, info
        self.Zt = Zs if st == 2 else Z
        self.solved_once = True
        u = self.Zt[:self.nu] + self.lastu0 + self.ss.uop
        self.lastu0 = u - self.ss.uop
        return u

    def getinfo(self):
        ss = self.ss
        Y0 = self.Et @ self.Zt + self.F
        U0 = self.Put @ self.Zt + self.Tu_lastu0
        X0 = self.Zt[self.nDU:self.nDU + self.nX]
        J = 0.5 * self.Zt @ self.Ht @ self.Zt + self.qt @ self.Zt + self.r
        return {"ΔU": self.Zt[:self.nDU].copy(), "ϵ": self.Zt[-1] if self.neps else 0.0, "Ŷ": Y0 + ss.Yop,
                "U": U0 + ss.Uop, "X̂0": X0.copy(), "x̂end": X0[-self.nxh:] + ss.xhop, "J": J,
                "u": self.lastu0 + ss.uop}
