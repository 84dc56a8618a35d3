"""CPU ORACLE (test infrastructure, NOT a product path) -- condensation of a LinMPC.

Float64 NumPy restatement of the LinModel + SingleShooting branch of
JuliaControl/ModelPredictiveControl.jl v2.11.0 (reference paths are relative to
/root/reference).  Everything here is written "the dense way" on purpose: it
materialises E, Pu, A ... exactly like the reference does, so that it is an independent
check on the HIP kernels (which never materialise any of those).

Parity pin: the reference is Julia (no toolchain in this image, JuMP/OSQP not vendored), so it
cannot be executed here.  This oracle is pinned instead against the reference's own
known-answer tests (tests/test_oracle_known_answers.py: T1..T9, SURVEY.md section 8c and 8(f3),
including the 6-digit doctest golden u = 17.577311 of ext/LinearMPCext.jl:255-269 and the
LQR-equivalence test at atol 1e-5 of test/3_test_predictive_control.jl:498-527), and against the
40-sample closed-loop series of the reference's README example taken from its own result figure
(docs/src/assets/readme_result.svg -> tests/golden/readme_result_series.json).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import numpy as np

INF = np.inf


# --------------------------------------------------------------------------------------
# estimator side: stochastic-integrator augmentation (data contract of the hot path)
# --------------------------------------------------------------------------------------
def init_integrators(nint, ny):
    """`init_integrators` -- src/estimator/construct.jl:226-252.

    nint[i] integrators in series on channel i; A is block lower-bidiagonal of ones and C picks
    the last state of every chain.
    """
    nint = np.zeros(ny, dtype=int) if np.isscalar(nint) and nint == 0 else np.asarray(nint, int)
    if nint.size != ny:
        raise ValueError("nint length mismatch")
    if np.any(nint < 0):
        raise ValueError("nint values should be >= 0")
    nx = int(nint.sum())
    A, C = np.zeros((nx, nx)), np.zeros((ny, nx))
    pos = 0
    for i in range(ny):
        k = int(nint[i])
        if k:
            blk = np.eye(k) + np.eye(k, k=-1)
            A[pos:pos + k, pos:pos + k] = blk
            C[i, pos + k - 1] = 1.0
            pos += k
    return A, C, nint


def init_estimstoch(nu, ny, i_ym, nint_u, nint_ym):
    """`init_estimstoch` + `stoch_ym2y` -- src/estimator/construct.jl:172-210."""
    i_ym = np.asarray(i_ym, int)
    As_u, Cs_u, nint_u = init_integrators(nint_u, nu)
    As_ym, Cs_ym, nint_ym = init_integrators(nint_ym, len(i_ym))
    Cs_y = np.zeros((ny, Cs_ym.shape[1]))
    Cs_y[i_ym, :] = Cs_ym
    nxs_u, nxs_y = As_u.shape[0], As_ym.shape[0]
    As = np.block([[As_u, np.zeros((nxs_u, nxs_y))], [np.zeros((nxs_y, nxs_u)), As_ym]])
    Cs_u = np.hstack([Cs_u, np.zeros((nu, nxs_y))])
    Cs_y = np.hstack([np.zeros((ny, nxs_u)), Cs_y])
    return As, Cs_u, Cs_y, nint_u, nint_ym


def augment_model(A, Bu, C, Bd, Dd, xop, fop, As, Cs_u, Cs_y):
    """`augment_model` -- src/estimator/construct.jl:305-323.

    Ahat = [A Bu*Cs_u; 0 As], Bhat_u = [Bu; 0], Chat = [C Cs_y], Bhat_d = [Bd; 0], Dhat_d = Dd,
    operating points padded with zeros.
    """
    nx, nu = Bu.shape
    nd = Bd.shape[1]
    nxs = As.shape[0]
    Ah = np.block([[A, Bu @ Cs_u], [np.zeros((nxs, nx)), As]])
    Bhu = np.vstack([Bu, np.zeros((nxs, nu))])
    Ch = np.hstack([C, Cs_y])
    Bhd = np.vstack([Bd, np.zeros((nxs, nd))])
    xhop = np.concatenate([xop, np.zeros(nxs)])
    fhop = np.concatenate([fop, np.zeros(nxs)])
    return Ah, Bhu, Ch, Bhd, Dd.copy(), xhop, fhop


# --------------------------------------------------------------------------------------
# controller side: horizons and conversion matrices
# --------------------------------------------------------------------------------------
def move_blocking(Hp, Hc):
    """`move_blocking` -- src/controller/construct.jl:629-660 (vector and integer methods)."""
    if np.isscalar(Hc):
        nb = [1] * int(Hc)
        if Hc > 0:
            nb[-1] = Hp - int(Hc) + 1
        return nb
    nb = [int(v) for v in Hc]
    if not all(v > 0 for v in nb):
        raise ValueError("Move blocking vector must be strictly positive integers.")
    if sum(nb) < Hp:
        nb = nb + [Hp - sum(nb)]
    elif sum(nb) > Hp:
        cs = np.cumsum(nb)
        last = int(np.argmax(cs >= Hp))
        nb = nb[:last + 1]
        if sum(nb) > Hp:
            nb[-1] = Hp - sum(nb[:-1])
    return nb


def init_ZtoDU(nu, Hc, nZ):
    """`init_ZtoΔU` -- src/controller/construct.jl:733-741: PΔu = [I 0]."""
    return np.hstack([np.eye(nu * Hc), np.zeros((nu * Hc, nZ - nu * Hc))])


def init_ZtoU(nu, Hp, Hc, nb, nZ):
    """`init_ZtoU` -- src/controller/construct.jl:792-809.

    Row-block of interval i holds i identity blocks (held cumulative sum), Tu = [I; ...; I].
    """
    Pu = np.zeros((nu * Hp, nZ))
    row = 0
    for i in range(Hc):
        for _ in range(nb[i]):
            for j in range(i + 1):
                Pu[row:row + nu, j * nu:(j + 1) * nu] = np.eye(nu)
            row += nu
    Tu = np.tile(np.eye(nu), (Hp, 1))
    return Pu, Tu


def init_predmat(Ah, Bhu, Ch, Bhd, Dhd, xhop, fhop, Hp, Hc, nb):
    """`init_predmat(::LinModel, ::SingleShooting)` -- src/controller/transcription.jl:115-194.

    Returns E, G, J, K, V, B, ex, gx, jx, kx, vx, bx (dense).
    """
    nxh, nu = Bhu.shape
    ny = Ch.shape[0]
    nd = Bhd.shape[1]
    # Apow[j] = Ah^j, j = 0..Hp  (:122-126), csum S(m) = sum_{l<=m} Ah^l (:128)
    Apow = np.empty((Hp + 1, nxh, nxh))
    Apow[0] = np.eye(nxh)
    for j in range(1, Hp + 1):
        Apow[j] = Apow[j - 1] @ Ah
    S = np.cumsum(Apow, axis=0)
    jl = np.concatenate([[0], np.cumsum(nb)]).astype(int)
    # K (:142-147)
    kx = Apow[Hp].copy()
    K = np.vstack([Ch @ Apow[j] for j in range(1, Hp + 1)])
    # V (:149-151): Q!(V, 0, Hp, 0)
    vx = S[Hp - 1] @ Bhu
    V = np.vstack([Ch @ S[l] @ Bhu for l in range(Hp)])
    # E, ex (:153-165)
    nZ = nu * Hc
    ex = np.empty((nxh, nZ))
    E = np.zeros((Hp * ny, nZ))
    for j in range(Hc):
        cols = slice(nu * j, nu * (j + 1))
        for i in range(j, Hc):
            i_Q, m_Q, b_Q = jl[i], jl[i + 1], jl[j]
            for l in range(m_Q - i_Q):
                r0 = ny * (i_Q + l)
                E[r0:r0 + ny, cols] = Ch @ S[i_Q - b_Q + l] @ Bhu
        ex[:, cols] = S[Hp - jl[j] - 1] @ Bhu
    # G, J, gx, jx (:167-182)
    gx = Apow[Hp - 1] @ Bhd
    G = np.zeros((Hp * ny, nd))
    jx = np.zeros((nxh, Hp * nd))
    J = np.kron(np.eye(Hp), Dhd) if nd > 0 else np.zeros((Hp * ny, 0))
    if nd > 0:
        for j in range(1, Hp + 1):
            G[ny * (j - 1):ny * j, :] = Ch @ Apow[j - 1] @ Bhd
        for j in range(1, Hp + 1):
            r = slice(ny * j, ny * Hp)
            c = slice(nd * (j - 1), nd * j)
            J[r, c] = G[0:ny * (Hp - j), :]
            jx[:, c] = Apow[Hp - j - 1] @ Bhd if j < Hp else 0.0
    # B, bx (:184-192)
    coefB = np.vstack([Ch @ S[j - 1] for j in range(1, Hp + 1)])
    dop = fhop - xhop
    bx = S[Hp - 1] @ dop
    B = coefB @ dop
    return E, G, J, K, V, B, ex, gx, jx, kx, vx, bx


# --------------------------------------------------------------------------------------
# constraint softening, box bounds, stacked A and i_b
# --------------------------------------------------------------------------------------
def relaxU(Pu, C_umin, C_umax, neps):
    """`relaxU` -- src/controller/construct.jl:999-1010."""
    if neps == 1:
        A_Umin = -np.hstack([Pu, C_umin[:, None]])
        A_Umax = np.hstack([Pu, -C_umax[:, None]])
        Put = np.hstack([Pu, np.zeros((Pu.shape[0], 1))])
    else:
        A_Umin, A_Umax, Put = -Pu, Pu.copy(), Pu.copy()
    return A_Umin, A_Umax, Put


def relaxDU(PDu, C_dumin, C_dumax, neps):
    """`relaxΔU` -- src/controller/construct.jl:1034-1044."""
    if neps == 1:
        A_min = -np.hstack([PDu, C_dumin[:, None]])
        A_max = np.hstack([PDu, -C_dumax[:, None]])
        n0, n1 = PDu.shape
        PDut = np.block([[PDu, np.zeros((n0, 1))], [np.zeros((1, n1)), np.ones((1, 1))]])
    else:
        A_min, A_max, PDut = -PDu, PDu.copy(), PDu.copy()
    return A_min, A_max, PDut


def relaxY(E, C_ymin, C_ymax, neps):
    """`relaxŶ` -- src/controller/construct.jl:1068-1083."""
    if neps == 1:
        A_min = -np.hstack([E, C_ymin[:, None]])
        A_max = np.hstack([E, -C_ymax[:, None]])
        Et = np.hstack([E, np.zeros((E.shape[0], 1))])
    else:
        A_min, A_max, Et = -E, E.copy(), E.copy()
    return A_min, A_max, Et


def relaxterminal(ex, c_xmin, c_xmax, neps):
    """`relaxterminal` -- src/controller/construct.jl:1183-1199."""
    if neps == 1:
        A_min = -np.hstack([ex, c_xmin[:, None]])
        A_max = np.hstack([ex, -c_xmax[:, None]])
        ext = np.hstack([ex, np.zeros((ex.shape[0], 1))])
    else:
        A_min, A_max, ext = -ex, ex.copy(), ex.copy()
    return A_min, A_max, ext


def relaxW(E, Pu, Hp, Wby, Wbu, C_wmin, C_wmax, neps):
    """`relaxW` -- src/controller/construct.jl:1138-1160: E_w = W̄y [0; E] + W̄u [Pu; pu] with
    pu the last nu rows of Pu; A_Wmin = -[E_w C_wmin], A_Wmax = [E_w -C_wmax], Ẽ_w = [E_w 0]."""
    nW = Wby.shape[0]
    ny = Wby.shape[1] // (Hp + 1)
    nu = Pu.shape[0] // Hp
    Ew = Wby @ np.vstack([np.zeros((ny, E.shape[1])), E]) + Wbu @ np.vstack([Pu, Pu[-nu:, :]])
    if neps == 1:
        return -np.hstack([Ew, C_wmin[:, None]]), np.hstack([Ew, -C_wmax[:, None]]), np.hstack([Ew, np.zeros((nW, 1))])
    return -Ew, Ew, Ew


def init_boxconstraint(nDU, neps, DUmin, DUmax, A_DUmin, A_DUmax):
    """`init_boxconstraint_mpc` -- src/controller/construct.jl:1209-1234 (SingleShooting).

    Hard (softness 0) ΔU bounds become variable bounds; eps >= 0.
    """
    nZt = nDU + neps
    Zmin, Zmax = np.full(nZt, -INF), np.full(nZt, INF)
    if neps > 0:
        Zmin[-1] = 0.0
        for i in range(nDU):
            if A_DUmin[i, -1] == 0:
                Zmin[i] = DUmin[i]
            if A_DUmax[i, -1] == 0:
                Zmax[i] = DUmax[i]
    else:
        Zmin[:nDU] = DUmin
        Zmax[:nDU] = DUmax
    return Zmin, Zmax


def init_matconstraint(Zmin, Zmax, U0min, U0max, DUmin, DUmax, Y0min, Y0max, x0min, x0max,
                       A_Umin, A_Umax, A_DUmin, A_DUmax, A_Ymin, A_Ymax, A_xmin, A_xmax,
                       Wmin=None, Wmax=None, A_Wmin=None, A_Wmax=None):
    """`init_matconstraint_mpc(::LinModel)` + `deleteΔU_lincon!`.

    src/controller/transcription.jl:667-703 and :783-789.  Row order
    [Umin; Umax; ΔUmin; ΔUmax; Ymin; Ymax; Wmin; Wmax; x̂min; x̂max] -- the x̂ blocks in the *correct* order
    (SURVEY 9.4 item 2).
    """
    if Wmin is None:
        Wmin = Wmax = np.zeros(0)
        A_Wmin = A_Wmax = np.zeros((0, A_Umin.shape[1]))
    A = np.vstack([A_Umin, A_Umax, A_DUmin, A_DUmax, A_Ymin, A_Ymax, A_Wmin, A_Wmax, A_xmin, A_xmax])
    fin = lambda v: ~np.isinf(v)
    i_DUmin, i_DUmax = fin(DUmin), fin(DUmax)
    nDU = len(DUmin)
    i_DUmin &= np.isinf(Zmin[:nDU])
    i_DUmax &= np.isinf(Zmax[:nDU])
    i_b = np.concatenate([fin(U0min), fin(U0max), i_DUmin, i_DUmax, fin(Y0min), fin(Y0max),
                          fin(Wmin), fin(Wmax), fin(x0min), fin(x0max)])
    return A, i_b


def init_quadprog(Et, PDut, Put, M_Hp, Nt_Hc, L_Hp):
    """`init_quadprog` -- src/controller/construct.jl:837-845 (Hermitian lower of the sum)."""
    Ht = 2.0 * (Et.T @ M_Hp @ Et + PDut.T @ Nt_Hc @ PDut + Put.T @ L_Hp @ Put)
    L = np.tril(Ht)
    return L + np.tril(Ht, -1).T


class LinMPCOracle:
    """One logical `LinMPC` (src/controller/linmpc.jl:3-111) on the augmented model.

    Inputs are the *augmented* estimator matrices (the data contract of the hot path,
    src/controller/transcription.jl:118) and operating points.  Weights may be dense
    (M_Hp, N_Hc, L_Hp) or per-channel vectors repeated over the horizon like the keyword
    constructor does (src/controller/linmpc.jl:229-253).
    """

    def __init__(self, Ah, Bhu, Ch, Bhd=None, Dhd=None, *, Hp, Hc=2, Mwt=None, Nwt=None, Lwt=None,
                 Cwt=1e5, M_Hp=None, N_Hc=None, L_Hp=None, uop=None, yop=None, dop=None,
                 xhop=None, fhop=None, Wy=None, Wu=None, Wd=None, Wr=None):
        Ah, Bhu, Ch = (np.atleast_2d(np.asarray(m, float)) for m in (Ah, Bhu, Ch))
        self.nxh, self.nu = Bhu.shape
        self.ny = Ch.shape[0]
        self.Bhd = np.zeros((self.nxh, 0)) if Bhd is None else np.atleast_2d(np.asarray(Bhd, float))
        self.nd = self.Bhd.shape[1]
        self.Dhd = np.zeros((self.ny, self.nd)) if Dhd is None else np.atleast_2d(np.asarray(Dhd, float))
        self.Ah, self.Bhu, self.Ch = Ah, Bhu, Ch
        nu, ny, nxh, nd = self.nu, self.ny, self.nxh, self.nd
        z = lambda v, n: np.zeros(n) if v is None else np.asarray(v, float).reshape(n)
        self.uop, self.yop, self.dop = z(uop, nu), z(yop, ny), z(dop, nd)
        self.xhop, self.fhop = z(xhop, nxh), z(fhop, nxh)
        self.Hp = int(Hp)
        self.nb = move_blocking(self.Hp, Hc)
        self.Hc = len(self.nb)
        Hp, Hc = self.Hp, self.Hc
        # weights -- defaults of src/general.jl:3-6
        Mwt = np.ones(ny) if Mwt is None else np.asarray(Mwt, float)
        Nwt = np.full(nu, 0.1) if Nwt is None else np.asarray(Nwt, float)
        Lwt = np.zeros(nu) if Lwt is None else np.asarray(Lwt, float)
        self.M_Hp = np.diag(np.tile(Mwt, Hp)) if M_Hp is None else np.asarray(M_Hp, float)
        self.N_Hc = np.diag(np.tile(Nwt, Hc)) if N_Hc is None else np.asarray(N_Hc, float)
        self.L_Hp = np.diag(np.tile(Lwt, Hp)) if L_Hp is None else np.asarray(L_Hp, float)
        self.Cwt = float(Cwt)
        self.neps = 0 if np.isinf(self.Cwt) else 1  # src/controller/construct.jl:903
        nDU = nu * Hc
        self.nDU, self.nZ = nDU, nDU
        self.nZt = nDU + self.neps
        # Ñ_Hc = blkdiag(N_Hc, C) -- src/controller/construct.jl:70-79
        if self.neps:
            self.Nt_Hc = np.block([[self.N_Hc, np.zeros((nDU, 1))],
                                   [np.zeros((1, nDU)), np.array([[self.Cwt]])]])
        else:
            self.Nt_Hc = self.N_Hc
        self.Uop, self.Yop, self.Dop = np.tile(self.uop, Hp), np.tile(self.yop, Hp), np.tile(self.dop, Hp)
        self.PDu = init_ZtoDU(nu, Hc, self.nZ)
        self.Pu, self.Tu = init_ZtoU(nu, Hp, Hc, self.nb, self.nZ)
        (self.E, self.G, self.J, self.K, self.V, self.B,
         self.ex, self.gx, self.jx, self.kx, self.vx, self.bx) = init_predmat(
            Ah, Bhu, Ch, self.Bhd, self.Dhd, self.xhop, self.fhop, Hp, Hc, self.nb)
        # default constraints -- src/controller/construct.jl:887-961
        self.U0min, self.U0max = np.full(nu * Hp, -INF), np.full(nu * Hp, INF)
        self.DUmin, self.DUmax = np.full(nDU, -INF), np.full(nDU, INF)
        self.Y0min, self.Y0max = np.full(ny * Hp, -INF), np.full(ny * Hp, INF)
        self.x0min, self.x0max = np.full(nxh, -INF), np.full(nxh, INF)
        self.C_umin, self.C_umax = np.zeros(nu * Hp), np.zeros(nu * Hp)
        self.C_dumin, self.C_dumax = np.zeros(nDU), np.zeros(nDU)
        self.C_ymin, self.C_ymax = np.ones(ny * Hp), np.ones(ny * Hp)
        self.c_xmin, self.c_xmax = np.ones(nxh), np.ones(nxh)
        # custom linear constraints -- validate_custom_lincon, src/controller/construct.jl:666-694
        given = [np.atleast_2d(np.asarray(W, float)) for W in (Wy, Wu, Wd, Wr) if W is not None]
        self.nw = given[0].shape[0] if given else 0
        nw = self.nw
        mat = lambda W, n: np.zeros((nw, n)) if W is None else np.atleast_2d(np.asarray(W, float))
        self.Wy, self.Wu, self.Wd, self.Wr = mat(Wy, ny), mat(Wu, nu), mat(Wd, nd), mat(Wr, ny)
        for W, n, name in ((self.Wy, ny, "Wy"), (self.Wu, nu, "Wu"), (self.Wd, nd, "Wd"), (self.Wr, ny, "Wr")):
            if W.shape != (nw, n):
                raise ValueError(f"{name} must have {nw} rows and {n} columns")
        rd = lambda W: np.kron(np.eye(Hp + 1), W)                   # repeatdiag, src/general.jl:226
        self.Wby, self.Wbu, self.Wbd, self.Wbr = rd(self.Wy), rd(self.Wu), rd(self.Wd), rd(self.Wr)
        self.nW = nw * (Hp + 1)
        self.Wmin, self.Wmax = np.full(self.nW, -INF), np.full(self.nW, INF)
        self.C_wmin, self.C_wmax = np.ones(self.nW), np.ones(self.nW)
        self._rebuild_constraints()
        self.Ht = init_quadprog(self.Et, self.PDut, self.Put, self.M_Hp, self.Nt_Hc, self.L_Hp)
        # state carried between calls
        self.Zt = np.zeros(self.nZt)
        self.lastu0 = np.zeros(nu)
        self.solved_once = False

    # ---- constraints -----------------------------------------------------------------
    def _rebuild_constraints(self):
        ne = self.neps
        self.A_Umin, self.A_Umax, self.Put = relaxU(self.Pu, self.C_umin, self.C_umax, ne)
        self.A_DUmin, self.A_DUmax, self.PDut = relaxDU(self.PDu, self.C_dumin, self.C_dumax, ne)
        self.A_Ymin, self.A_Ymax, self.Et = relaxY(self.E, self.C_ymin, self.C_ymax, ne)
        self.A_xmin, self.A_xmax, self.ext = relaxterminal(self.ex, self.c_xmin, self.c_xmax, ne)
        self.Zmin, self.Zmax = init_boxconstraint(self.nDU, ne, self.DUmin, self.DUmax,
                                                  self.A_DUmin if ne else None,
                                                  self.A_DUmax if ne else None)
        self.A_Wmin, self.A_Wmax, self.Ewt = relaxW(self.E, self.Pu, self.Hp, self.Wby, self.Wbu,
                                                    self.C_wmin, self.C_wmax, ne)
        self.A, self.i_b = init_matconstraint(
            self.Zmin, self.Zmax, self.U0min, self.U0max, self.DUmin, self.DUmax,
            self.Y0min, self.Y0max, self.x0min, self.x0max,
            self.A_Umin, self.A_Umax, self.A_DUmin, self.A_DUmax,
            self.A_Ymin, self.A_Ymax, self.A_xmin, self.A_xmax,
            self.Wmin, self.Wmax, self.A_Wmin, self.A_Wmax)

    def setconstraint(self, *, umin=None, umax=None, dumin=None, dumax=None, ymin=None, ymax=None,
                      xhatmin=None, xhatmax=None, Umin=None, Umax=None, DUmin=None, DUmax=None,
                      Ymin=None, Ymax=None, c_umin=None, c_umax=None, c_dumin=None, c_dumax=None,
                      c_ymin=None, c_ymax=None, c_xhatmin=None, c_xhatmax=None,
                      wmin=None, wmax=None, Wmin=None, Wmax=None, c_wmin=None, c_wmax=None,
                      C_wmin=None, C_wmax=None, C_umin=None, C_umax=None, C_dumin=None, C_dumax=None,
                      C_ymin=None, C_ymax=None):
        """`setconstraint!` -- src/controller/construct.jl:324-559 (horizon-long softness `C_umin` ... `C_ymax`: :446-483)."""
        Hp, Hc = self.Hp, self.Hc
        f = lambda v: np.asarray(v, float).ravel()
        for v, name in ((wmin, "wmin"), (wmax, "wmax"), (c_wmin, "c_wmin"), (c_wmax, "c_wmax")):
            if v is not None and f(v).shape != (self.nw,):
                raise ValueError(f"{name} size must be ({self.nw},)")       # construct.jl:411,420
        if Wmin is None and wmin is not None:
            self.Wmin = np.tile(f(wmin), Hp + 1)
        elif Wmin is not None:
            self.Wmin = f(Wmin)
        if Wmax is None and wmax is not None:
            self.Wmax = np.tile(f(wmax), Hp + 1)
        elif Wmax is not None:
            self.Wmax = f(Wmax)
        if Umin is None and umin is not None:
            self.U0min = np.tile(f(umin), Hp) - self.Uop
        elif Umin is not None:
            self.U0min = f(Umin) - self.Uop
        if Umax is None and umax is not None:
            self.U0max = np.tile(f(umax), Hp) - self.Uop
        elif Umax is not None:
            self.U0max = f(Umax) - self.Uop
        if DUmin is None and dumin is not None:
            self.DUmin = np.tile(f(dumin), Hc)
        elif DUmin is not None:
            self.DUmin = f(DUmin)
        if DUmax is None and dumax is not None:
            self.DUmax = np.tile(f(dumax), Hc)
        elif DUmax is not None:
            self.DUmax = f(DUmax)
        if Ymin is None and ymin is not None:
            self.Y0min = np.tile(f(ymin), Hp) - self.Yop
        elif Ymin is not None:
            self.Y0min = f(Ymin) - self.Yop
        if Ymax is None and ymax is not None:
            self.Y0max = np.tile(f(ymax), Hp) - self.Yop
        elif Ymax is not None:
            self.Y0max = f(Ymax) - self.Yop
        if xhatmin is not None:
            self.x0min = f(xhatmin) - self.xhop
        if xhatmax is not None:
            self.x0max = f(xhatmax) - self.xhop
        ecrs = (c_umin, c_umax, c_dumin, c_dumax, c_ymin, c_ymax, c_xhatmin, c_xhatmax,
                c_wmin, c_wmax, C_wmin, C_wmax, C_umin, C_umax, C_dumin, C_dumax, C_ymin, C_ymax)
        if any(e is not None for e in ecrs):
            if self.neps != 1:
                raise ValueError("Slack variable weight Cwt must be finite to set softness parameters")
            if self.solved_once:
                raise RuntimeError("Cannot set softness parameters after calling moveinput!")
        if c_umin is not None:
            self.C_umin = np.tile(f(c_umin), Hp)
        if c_umax is not None:
            self.C_umax = np.tile(f(c_umax), Hp)
        if c_dumin is not None:
            self.C_dumin = np.tile(f(c_dumin), Hc)
        if c_dumax is not None:
            self.C_dumax = np.tile(f(c_dumax), Hc)
        if c_ymin is not None:
            self.C_ymin = np.tile(f(c_ymin), Hp)
        if c_ymax is not None:
            self.C_ymax = np.tile(f(c_ymax), Hp)
        for whole, name, n in ((C_umin, "C_umin", self.nu * Hp), (C_umax, "C_umax", self.nu * Hp), (C_dumin, "C_dumin", self.nu * Hc),
                               (C_dumax, "C_dumax", self.nu * Hc), (C_ymin, "C_ymin", self.ny * Hp), (C_ymax, "C_ymax", self.ny * Hp)):
            if whole is not None:                       # construct.jl:454-483
                if f(whole).shape != (n,):
                    raise ValueError(f"{name} size must be ({n},)")
                if np.any(f(whole) < 0):
                    raise ValueError(f"{name} weights should be non-negative")
                setattr(self, name, f(whole))
        if c_xhatmin is not None:
            self.c_xmin = f(c_xhatmin)
        if c_xhatmax is not None:
            self.c_xmax = f(c_xhatmax)
        for v in (c_wmin, c_wmax, C_wmin, C_wmax):
            if v is not None and np.any(f(v) < 0):
                raise RuntimeError("softness parameters must be nonnegative")    # construct.jl:495-509
        if C_wmin is None and c_wmin is not None:
            self.C_wmin = np.tile(f(c_wmin), Hp + 1)
        elif C_wmin is not None:
            self.C_wmin = f(C_wmin)
        if C_wmax is None and c_wmax is not None:
            self.C_wmax = np.tile(f(c_wmax), Hp + 1)
        elif C_wmax is not None:
            self.C_wmax = f(C_wmax)
        old_ib, old_zmin, old_zmax = self.i_b.copy(), self.Zmin.copy(), self.Zmax.copy()
        self._rebuild_constraints()
        if self.solved_once:
            # src/controller/construct.jl:541-551: only finite values may change
            if (np.any(old_ib != self.i_b) or np.any(np.isinf(old_zmin) != np.isinf(self.Zmin))
                    or np.any(np.isinf(old_zmax) != np.isinf(self.Zmax))):
                raise RuntimeError("Cannot modify ±Inf constraints after calling moveinput!")
        return self

    # ---- per-step ---------------------------------------------------------------------
    def initpred(self, xhat0, lastu, ry=None, d=None, Dhat=None, Rhaty=None, Rhatu=None):
        """`initpred!(::LinModel)` + `initpred_common!` -- src/controller/execute.jl:247-314.

        Returns (F, qt, r).  Also stores lastu0 and Tu*lastu0 like the reference.
        """
        Hp = self.Hp
        self.lastu0 = np.asarray(lastu, float) - self.uop
        self.Tu_lastu0 = self.Tu @ self.lastu0
        ry = self.yop if ry is None else np.asarray(ry, float)
        Rhaty = np.tile(ry, Hp) if Rhaty is None else np.asarray(Rhaty, float)
        Rhatu = self.Uop if Rhatu is None else np.asarray(Rhatu, float)
        F = self.B + self.K @ xhat0 + self.V @ self.lastu0
        if self.nd > 0:
            d = np.asarray(d, float)
            Dhat = np.tile(d, Hp) if Dhat is None else np.asarray(Dhat, float)
            self.d0 = d - self.dop
            self.D0 = Dhat - self.Dop
            F = F + self.G @ self.d0 + self.J @ self.D0
        q = np.zeros(self.nZt)
        r = 0.0
        if np.any(self.M_Hp != 0):
            Cy = F + self.Yop - Rhaty
            q += (self.M_Hp @ self.Et).T @ Cy
            r += Cy @ self.M_Hp @ Cy
        if np.any(self.L_Hp != 0):
            Cu = self.Tu_lastu0 + self.Uop - Rhatu
            q += (self.L_Hp @ self.Put).T @ Cu
            r += Cu @ self.L_Hp @ Cu
        self.F, self.qt, self.r = F, 2.0 * q, r
        self.xhat0 = np.asarray(xhat0, float)
        self.ry, self.Rhaty = ry, Rhaty
        # ŷ(k) = evaloutput(estim, d) -- src/controller/execute.jl:304 (initpred_common!)
        self.yhat = self.Ch @ self.xhat0 + self.yop + (self.Dhd @ self.d0 if self.nd > 0 else 0.0)
        return self.F, self.qt, self.r

    def linconstraint(self):
        """`linconstraint!(::LinModel)` -- src/controller/transcription.jl:811-848."""
        fx = self.bx + self.kx @ self.xhat0 + self.vx @ self.lastu0
        if self.nd > 0:
            fx = fx + self.gx @ self.d0 + self.jx @ self.D0
        self.fx = fx
        # F_w -- linconstraint_custom!, src/controller/execute.jl:337-364 (engineering values)
        Fw = np.zeros(self.nW)
        if self.nw > 0:
            Fw += self.Wbu @ np.concatenate([self.Tu_lastu0 + self.Uop, self.lastu0 + self.uop])
            if self.nd > 0:
                Fw += self.Wbd @ np.concatenate([self.d0 + self.dop, self.D0 + self.Dop])
            Fw += self.Wbr @ np.concatenate([self.ry, self.Rhaty])
            Fw += self.Wby @ np.concatenate([self.yhat, self.F + self.Yop])
        self.Fw = Fw
        self.b = np.concatenate([
            -self.U0min + self.Tu_lastu0, self.U0max - self.Tu_lastu0,
            -self.DUmin, self.DUmax,
            -self.Y0min + self.F, self.Y0max - self.F,
            -self.Wmin + Fw, self.Wmax - Fw,
            -self.x0min + fx, self.x0max - fx])
        return self.b

    def warmstart(self):
        """`set_warmstart_mpc!(::SingleShooting)` -- src/controller/transcription.jl:997-1007."""
        nu, nDU = self.nu, self.nDU
        Zs = np.zeros(self.nZt)
        Zs[:nDU - nu] = self.Zt[nu:nDU]
        if self.neps == 1:
            Zs[-1] = self.Zt[-1]
        return Zs

    def qp_data(self):
        """The QP handed to the solver: (H, q, A[i_b], b[i_b], Zmin, Zmax).

        src/controller/linmpc.jl:323-339 and src/controller/execute.jl:796-799.
        """
        return self.Ht, self.qt, self.A[self.i_b], self.b[self.i_b], self.Zmin, self.Zmax

    def moveinput(self, xhat0, ry=None, d=None, *, lastu=None, Dhat=None, Rhaty=None, Rhatu=None,
                  solver=None):
        """`moveinput!` -- src/controller/execute.jl:59-80 with `optim_objective!` (:466-505)
        and `getinput!` (:536-546).  `solver(H,q,A,b,zmin,zmax,z0) -> (z, status)`; status 2 =
        error => warm start returned (:499-500)."""
        from . import qp as _qp
        lastu = self.lastu0 + self.uop if lastu is None else lastu
        self.initpred(xhat0, lastu, ry, d, Dhat, Rhaty, Rhatu)
        self.linconstraint()
        Zs = self.warmstart()
        solve = _qp.solve_qp if solver is None else solver
        z, status = solve(*self.qp_data(), Zs)
        self.status = status
        self.Zt = Zs if status == 2 else z
        self.solved_once = True
        u = self.Zt[:self.nu] + self.lastu0 + self.uop
        self.lastu0 = u - self.uop
        return u

    def getinfo(self):
        """`getinfo` subset -- src/controller/execute.jl:145-198 and `predict!`
        (src/controller/transcription.jl:1136-1145)."""
        Y0 = self.Et @ self.Zt + self.F
        # lastu0 was already advanced by getinput!; U uses the pre-step value via Tu_lastu0
        U0 = self.Put @ self.Zt + self.Tu_lastu0
        xend = self.ext @ self.Zt + self.fx
        J = 0.5 * self.Zt @ self.Ht @ self.Zt + self.qt @ self.Zt + self.r
        return {"ΔU": self.Zt[:self.nDU].copy(), "ϵ": self.Zt[-1] if self.neps else 0.0,
                "Ŷ": Y0 + self.Yop, "U": U0 + self.Uop, "x̂end": xend + self.xhop, "J": J,
                "W": self.Ewt @ self.Zt + self.Fw,                     # execute.jl:221
                "u": self.lastu0 + self.uop}
