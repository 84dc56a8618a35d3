"""CPU ORACLE (test infrastructure, NOT a product path) -- the QP the reference hands to JuMP.

    min_z  1/2 z'Hz + q'z   s.t.  A z <= b,  zmin <= z <= zmax        (src/general.jl:107,
                                         src/controller/linmpc.jl:323-339)

The reference solves it with OSQP (C library, third party, not vendored under /root/reference;
compat "0.8", Project.toml:44) at its default ~1e-3 tolerance, or with DAQP (exact active-set).
What is pinned by the reference's tests is the QP *optimum* (LQR equivalence at 1e-5,
test/3_test_predictive_control.jl:498-527; DAQP vs LinearMPC at 1e-10, test/5_test_extensions.jl:41),
not OSQP's iterates.  This oracle therefore returns the optimum to ~1e-10 and *certifies* it:

  1. dense float64 Mehrotra predictor-corrector interior point driven to the float64 floor of
     the complementarity gap (absolute mu ~1e-14; soft-constraint multipliers reach 2*Cwt*eps ~1e5,
     so a *relative* residual test is far too loose here -- measured: 1e-3 errors in ΔU),
  2. active-set polish (equality-constrained KKT solve on the detected active rows, with a few
     add/drop corrections),
  3. a rigorous a-posteriori bound on |z - z*|: for a (numerically) feasible z, lam >= 0,
     r = Hz+q+G'lam and sigma = lambda_min(H) > 0, strong convexity gives
         sigma*e^2 <= |r|*e + lam's   =>   e <= (|r| + sqrt(|r|^2 + 4 sigma lam's)) / (2 sigma),
     independent of how (z, lam) was found.  `info["err_bound"]` carries it; `info["kkt"]` the
     individual residuals; tests assert on them.

Status codes mirror SURVEY 8(b): 0 OPTIMAL, 1 ITERATION_LIMIT (solution kept, like the @warn
branch src/controller/execute.jl:491-496), 2 INFEASIBLE/NUMERICAL_ERROR (caller returns the
shifted warm start, :499-500).
"""
from __future__ import annotations

import numpy as np

OPTIMAL, ITERATION_LIMIT, INFEASIBLE = 0, 1, 2


def stack_constraints(A, b, zmin, zmax):
    """Append finite variable bounds as rows: G z <= h."""
    n = len(zmin)
    I = np.eye(n)
    lo, hi = ~np.isinf(zmin), ~np.isinf(zmax)
    G = np.vstack([A.reshape(-1, n), -I[lo], I[hi]])
    h = np.concatenate([b, -zmin[lo], zmax[hi]])
    return G, h


def kkt_residuals(H, q, G, h, z, lam):
    """Certificate of optimality for min 1/2 z'Hz+q'z s.t. Gz<=h."""
    scale = 1.0 + max(np.abs(q).max(initial=0.0), np.abs(H @ z).max(initial=0.0))
    slack = h - G @ z
    return {
        "stationarity": float(np.abs(H @ z + q + G.T @ lam).max(initial=0.0) / scale),
        "primal": float(np.maximum(-slack, 0.0).max(initial=0.0) / (1.0 + np.abs(h).max(initial=0.0))),
        "dual": float(np.maximum(-lam, 0.0).max(initial=0.0) / scale),
        "complementarity": float(np.abs(lam * slack).max(initial=0.0) / scale),
    }


def ipm(H, q, G, h, z0=None, mu_tol=1e-14, res_tol=1e-10, maxit=200):
    """Mehrotra predictor-corrector (Nocedal & Wright, Alg. 16.4).

    The Newton systems are solved on the quasi-definite reduced KKT matrix
    [[H, G'], [G, -diag(s/lam)]] by LU with partial pivoting (accuracy over speed: the normal
    equations H + G' diag(lam/s) G -- what the HIP kernel factorises -- lose positive
    definiteness in float64 once lam/s spans ~20 decades, which soft constraints with
    Cwt = 1e5 do reach; the oracle must keep converging there to be a reference).

    Stops when the ABSOLUTE mean complementarity mu = s'lam/m <= mu_tol and both residuals are at
    res_tol relative to the terms they are differences of -- or when mu stalls at the float64
    floor.  Returns z, lam, s, iters, status.
    """
    n, m = len(q), len(h)
    if m == 0:
        return np.linalg.solve(H, -q), np.zeros(0), np.zeros(0), 0, OPTIMAL
    z = np.zeros(n) if z0 is None else np.array(z0, float)
    K = np.zeros((n + m, n + m))
    K[:n, :n], K[:n, n:], K[n:, :n] = H, G.T, G
    idx = np.arange(n, n + m)

    def newton(s, lam, rd, rp, rc):
        # H dz + G'dl = -rd ; G dz + ds = -rp ; lam ds + s dl = -rc   (ds eliminated)
        K[idx, idx] = -s / lam
        rhs = np.concatenate([-rd, -rp + rc / lam])
        try:
            sol = np.linalg.solve(K, rhs)
        except np.linalg.LinAlgError:
            # s/lam underflows to ~1e-20 on more active rows than variables (a degenerate vertex reached with mu ~ 1e-9):
            # the KKT matrix is then singular in float64 although the iterate is an optimum to 1e-9.  A dual
            # regularisation of 1e-13 -- the same device as the kernels' delta -- makes the system solvable; it multiplies
            # dlam, which vanishes at the optimum.  (Met by tests/parity_util.offset_tables_case, member 1: a terminal
            # bound violated by 3.5 at the start, 12 iterations to mu = 1e-9, then LinAlgError.)
            K[idx, idx] = -s / lam - 1e-13
            sol = np.linalg.solve(K, rhs)
        dz, dl = sol[:n], sol[n:]
        ds = -rp - G @ dz
        return dz, ds, dl

    # starting point: affine step from (z, s=1, lam=1) then push into the interior
    s, lam = np.ones(m), np.ones(m)
    dz, ds, dl = newton(s, lam, H @ z + q + G.T @ lam, G @ z + s - h, s * lam)
    s = np.maximum(np.abs(s + ds), 1.0)
    lam = np.maximum(np.abs(lam + dl), 1.0)
    z = z + dz
    nh = 1.0 + np.abs(h).max()
    status = ITERATION_LIMIT
    it = 0
    best = None
    stall = 0
    for it in range(1, maxit + 1):
        Hz, Gl = H @ z, G.T @ lam
        rd = Hz + q + Gl
        rp = G @ z + s - h
        mu = s @ lam / m
        nd = 1.0 + max(np.abs(q).max(), np.abs(Hz).max(), np.abs(Gl).max())
        if np.abs(rd).max() <= res_tol * nd and np.abs(rp).max() <= res_tol * nh:
            if best is None or mu < 0.5 * best[0]:
                stall = 0
            else:
                stall += 1
            if best is None or mu < best[0]:
                best = (mu, z.copy(), lam.copy(), s.copy())
            if mu <= mu_tol or (stall >= 3 and best[0] <= 1e-9):   # stalled at the float64 floor
                status = OPTIMAL
                it -= 1
                break
        # jammed well above the floor (products s_i lam_i spread over many decades, steps cut short
        # by the boundary): a pure centring step (sigma = 1) restores the neighbourhood
        centre = stall >= 3 and mu > 1e-9
        if centre:
            stall = 0
        try:
            if centre:
                dz, ds, dl = newton(s, lam, rd, rp, s * lam - mu)
            else:
                dz, ds, dl = newton(s, lam, rd, rp, s * lam)          # predictor
                a = _maxstep(s, ds, lam, dl)
                mu_aff = (s + a * ds) @ (lam + a * dl) / m
                sigma = (mu_aff / mu) ** 3
                dz, ds, dl = newton(s, lam, rd, rp, s * lam + ds * dl - sigma * mu)   # corrector
        except np.linalg.LinAlgError:
            break
        a = min(1.0, 0.995 * _maxstep(s, ds, lam, dl))
        z, s, lam = z + a * dz, s + a * ds, lam + a * dl
        if not (np.all(np.isfinite(z)) and np.all(np.isfinite(lam))):
            break
    if status == OPTIMAL:
        _, z, lam, s = best
    elif best is not None and best[0] <= 1e-9:
        _, z, lam, s = best        # stalled at the numerical floor after convergence
        status = OPTIMAL
    elif it >= maxit and np.all(np.isfinite(z)) and np.abs(G @ z + s - h).max() <= 1e-6 * nh:
        status = ITERATION_LIMIT
    else:
        status = INFEASIBLE        # primal residual never vanished: the reference's error branch
    return z, lam, s, it, status


def _maxstep(s, ds, lam, dl):
    a = 1.0
    neg = ds < 0
    if neg.any():
        a = min(a, float(np.min(-s[neg] / ds[neg])))
    neg = dl < 0
    if neg.any():
        a = min(a, float(np.min(-lam[neg] / dl[neg])))
    return a


def polish(H, q, G, h, z, lam, s, rounds=6, act=None):
    """Active-set refinement: rows with lam_i > s_i (or the given set `act`) are taken active and the
    equality-constrained KKT system is solved exactly (least-squares: duplicated active rows are
    legal, e.g. a saturated input held over a move-blocking interval); rows with a negative
    multiplier are dropped and violated rows added, a few times."""
    n = len(q)
    act = (lam > s) if act is None else act.copy()
    tolh = 1e-11 * (1.0 + np.abs(h).max())
    zp, lp = z, lam
    for _ in range(rounds):
        k = int(act.sum())
        lp = np.zeros_like(lam)
        if k == 0:
            zp = np.linalg.solve(H, -q)
        else:
            zp, la = _eqp(H, q, G[act], h[act])
            lp[act] = la
        drop = act & (lp < 0)
        add = (~act) & (G @ zp - h > tolh)
        if not drop.any() and not add.any():
            break
        act = (act & ~drop) | add
    return zp, lp


def _eqp(H, q, Ga, ha):
    """min 1/2 z'Hz+q'z s.t. Ga z = ha (Ga may have dependent rows): symmetric diagonal
    equilibration of the KKT matrix (H carries 2*Cwt = 2e5 next to O(1) entries), minimum-norm
    least-squares solve, then iterative refinement with the residual in extended precision."""
    n, k = len(q), len(ha)
    dz = 1.0 / np.sqrt(np.diag(H))
    Hs, Gs = H * np.outer(dz, dz), Ga * dz
    dr = 1.0 / np.maximum(np.linalg.norm(Gs, axis=1), 1e-300)
    Gs = Gs * dr[:, None]
    KKT = np.block([[Hs, Gs.T], [Gs, np.zeros((k, k))]])
    rhs = np.concatenate([-q * dz, ha * dr])
    pinv = np.linalg.pinv(KKT, rcond=1e-13)
    x = pinv @ rhs
    KL, rl = KKT.astype(np.longdouble), rhs.astype(np.longdouble)
    for _ in range(3):
        res = (rl - KL @ x.astype(np.longdouble)).astype(float)
        x = x + pinv @ res
    return x[:n] * dz, x[n:] * dr


def error_bound(H, q, G, h, z, lam, sigma=None):
    """Rigorous bound on |z - z*|_2 (see module docstring); inf if z is infeasible beyond
    rounding or lam has a negative entry beyond rounding."""
    sigma = float(np.linalg.eigvalsh(H)[0]) if sigma is None else sigma
    slack = h - G @ z
    tolh = 1e-9 * (1.0 + np.abs(h).max(initial=0.0))
    if sigma <= 0 or slack.min(initial=0.0) < -tolh or lam.min(initial=0.0) < -1e-9 * (1 + np.abs(lam).max(initial=0.0)):
        return np.inf
    lam = np.maximum(lam, 0.0)
    r = np.linalg.norm(H @ z + q + G.T @ lam)
    gap = float(lam @ np.maximum(slack, 0.0))
    return float((r + np.sqrt(r * r + 4.0 * sigma * gap)) / (2.0 * sigma))


def active_set_certificate(H, q, G, h, z, lam):
    """Exact-KKT check of an active-set point: lam >= 0 with lam_i = 0 off the working set,
    working-set rows tight, all rows feasible, stationarity at rounding level.  When it holds,
    z is the optimum of a QP whose data differ from (H, q, G, h) by rounding-size perturbations,
    i.e. it is the unique optimum to ~1e-10 (strong convexity: |dz| <= |dq| / lambda_min(H))."""
    nh = 1.0 + np.abs(h).max(initial=0.0)
    nl = 1.0 + np.abs(lam).max(initial=0.0)
    # evaluated in extended precision: the terms of the stationarity sum reach 1e6 (multipliers of
    # violated soft rows, 2 Cwt eps), and a residual of 1e-11 of that scale -- float64 evaluation
    # noise -- would still allow an error of 1e-5 / lambda_min(H) in z.  The targets below are
    # 1e-13 of the scale for stationarity and 1e-12 for the row conditions.
    L = np.longdouble
    zl, ll = z.astype(L), lam.astype(L)
    slack = (h.astype(L) - G.astype(L) @ zl).astype(float)
    Hz, Gl = (H.astype(L) @ zl), (G.T.astype(L) @ ll)
    nd = 1.0 + max(np.abs(q).max(), float(np.abs(Hz).max()), float(np.abs(Gl).max()))
    stat = float(np.abs(Hz + q.astype(L) + Gl).max())
    w = lam != 0
    return bool(lam.min(initial=0.0) >= -1e-12 * nl and slack.min(initial=0.0) >= -1e-12 * nh
                and np.abs(slack[w]).max(initial=0.0) <= 1e-12 * nh
                and stat <= 1e-13 * nd)


def solve_qp(H, q, A, b, zmin, zmax, z0=None, return_info=False):
    """Optimum of the reference QP.  Returns (z, status[, info]).

    info["certificate"] is "active-set" when the polished point passes the exact KKT check
    (accuracy ~1e-10), else "ipm-bound" with the rigorous but pessimistic info["err_bound"]."""
    G, h = stack_constraints(A, b, zmin, zmax)
    z, lam, s, it, status = ipm(H, q, G, h, z0)
    info = {"iters": it, "ipm_status": status, "polished": False, "certificate": "none"}
    if status != INFEASIBLE:
        info["certificate"] = "ipm-bound"
        info["err_bound"] = error_bound(H, q, G, h, z, lam)
        if len(h):
            # working sets tried in turn: the interior-point partition lam_i > s_i, then (degenerate
            # vertices, where weakly active rows have s_i ~ lam_i) the rows that are tight at the
            # interior-point optimum to 1e-7, 1e-6, 1e-8 of the bound scale; each with add/drop rounds
            nh = 1.0 + np.abs(h).max()
            slack = h - G @ z
            starts = [None] + [slack < t * nh for t in (1e-7, 1e-6, 1e-8)]
            for act0 in starts:
                zp, lp = polish(H, q, G, h, z, lam, s, rounds=12 if act0 is not None else 6, act=act0)
                if np.all(np.isfinite(zp)) and active_set_certificate(H, q, G, h, zp, lp):
                    info["ipm_vs_polish"] = float(np.abs(z - zp).max())
                    z, lam, info["polished"], info["certificate"] = zp, lp, True, "active-set"
                    break
        else:
            info["certificate"] = "active-set"
        info["kkt"] = kkt_residuals(H, q, G, h, z, lam)
    info["lam"] = lam
    if return_info:
        return z, status, info
    return z, status
