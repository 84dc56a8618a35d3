"""CPU ORACLE (test infrastructure, NOT a product path) -- the QP optimum in extended precision.

    min_z  1/2 z'Hz + q'z   s.t.  G z <= h          (the QP of oracle/qp.py; src/general.jl:107)

`oracle/qp.py` certifies its optimum with an exact-KKT check of an active-set point; on long horizons with
ill-conditioned H̃ (the instances the reference's documentation sends to MultipleShooting,
src/controller/construct.jl:855-866) its float64 polish sometimes finds no working set and the point stays
"ipm-bound": nobody can then say which of two float64 answers 1e-4 apart is right (VERDICT round 3, item 1d:
instance 99 of shape nx=8 nu=2 ny=2 Hp=60 Hc=40).  This module adjudicates: the SAME primal-dual interior-point
iteration (Mehrotra predictor-corrector on the normal equations H + G'(lam/s)G) run in `mpmath` arithmetic of `digits`
decimal digits, with the float64 problem data taken as exact numbers.  At 60 digits the normal equations keep ~35
digits through a condition number of 1e25, so the iteration is driven to a complementarity gap of 1e-36 and the
returned z is the optimum of the float64-data QP to far better than 1e-15 -- whatever its active set, degenerate or not.
`certify()` evaluates the KKT residuals of that point in the same arithmetic (stationarity, feasibility, sign,
complementarity) and the strong-convexity error bound of oracle/qp.py:error_bound, so the claim is checked and not
assumed.

Cost: O(m n^2) multiprecision operations per iteration in pure Python -- seconds per iteration at n = 81, m = 361.
It is used offline (scripts/adjudicate_instances.py writes tests/golden/hp_optima.json) and by a CPU test on a small
problem; the GPU tests compare the kernel with the committed optima.
"""
from __future__ import annotations

import numpy as np

try:
    import mpmath as mp
except ImportError:                                    # pragma: no cover
    mp = None


def _vec(a):
    return [mp.mpf(float(v)) for v in np.asarray(a, float).ravel()]


def _chol_solve(P, rhs_list):
    """In-place Cholesky of the symmetric positive definite list-of-lists P, then solves for every right-hand side."""
    n = len(P)
    for j in range(n):
        Pj = P[j]
        d = Pj[j] - mp.fsum(Pj[k] * Pj[k] for k in range(j))
        if d <= 0:
            raise ZeroDivisionError("normal equations not positive definite at the working precision")
        d = mp.sqrt(d)
        Pj[j] = d
        for i in range(j + 1, n):
            Pi = P[i]
            Pi[j] = (Pi[j] - mp.fsum(Pi[k] * Pj[k] for k in range(j))) / d
    out = []
    for r in rhs_list:
        y = list(r)
        for i in range(n):
            y[i] = (y[i] - mp.fsum(P[i][k] * y[k] for k in range(i))) / P[i][i]
        for i in range(n - 1, -1, -1):
            y[i] = (y[i] - mp.fsum(P[k][i] * y[k] for k in range(i + 1, n))) / P[i][i]
        out.append(y)
    return out


def solve(H, q, G, h, z0=None, lam0=None, s0=None, digits=60, mu_tol=None, maxit=200, verbose=False):
    """Optimum of the QP in `digits`-digit arithmetic.  Returns (z as float64 array, info).  `z0, lam0, s0`: an interior
    float64 iterate to start from (e.g. oracle/qp.py's final one: a dozen iterations instead of 60)."""
    if mp is None:
        raise ImportError("mpmath is not installed")
    H, q, G, h = (np.asarray(a, float) for a in (H, q, G, h))
    n, m = len(q), len(h)
    with mp.workdps(digits):
        mu_tol = mp.mpf(10) ** (-(digits * 3) // 5) if mu_tol is None else mp.mpf(mu_tol)
        Hm = [[mp.mpf(float(H[i, j])) for j in range(n)] for i in range(n)]
        qm, hm = _vec(q), _vec(h)
        # rows of G as sparse lists (the U rows and the box rows have a few non-zeros)
        rows = []
        for i in range(m):
            nz = np.flatnonzero(G[i])
            rows.append([(int(k), mp.mpf(float(G[i, k]))) for k in nz])
        Gz = lambda v: [mp.fsum(c * v[k] for k, c in r) for r in rows]

        def Gt(w):
            out = [mp.mpf(0)] * n
            for r, wi in zip(rows, w):
                if wi != 0:
                    for k, c in r:
                        out[k] += c * wi
            return out

        Hz = lambda v: [mp.fsum(Hm[i][k] * v[k] for k in range(n)) for i in range(n)]
        z = _vec(np.zeros(n) if z0 is None else z0)
        if lam0 is not None and s0 is not None:
            tiny = mp.mpf(10) ** (-digits // 2)
            s = [max(mp.mpf(float(v)), tiny) for v in s0]
            lam = [max(mp.mpf(float(v)), tiny) for v in lam0]
        else:
            gz = Gz(z)
            s = [max(hm[i] - gz[i], mp.mpf(1)) for i in range(m)]
            lam = [mp.mpf(10) / s[i] for i in range(m)]
        status, it = 1, 0
        for it in range(1, maxit + 1):
            hz, gl, gz = Hz(z), Gt(lam), Gz(z)
            rd = [hz[i] + qm[i] + gl[i] for i in range(n)]
            rp = [gz[i] + s[i] - hm[i] for i in range(m)]
            mu = mp.fsum(s[i] * lam[i] for i in range(m)) / m
            rdn, rpn = max(abs(v) for v in rd), max(abs(v) for v in rp)
            if verbose:
                print(f"  hp it {it:3d} mu {mp.nstr(mu, 4)} rd {mp.nstr(rdn, 4)} rp {mp.nstr(rpn, 4)}", flush=True)
            if mu <= mu_tol and rdn <= mu_tol * 10 and rpn <= mu_tol * 10:
                status = 0
                break
            D = [lam[i] / s[i] for i in range(m)]
            P = [row[:] for row in Hm]
            for r, di in zip(rows, D):
                for a_, (ka, ca) in enumerate(r):
                    t = ca * di
                    Pk = P[ka]
                    for kb, cb in r:
                        Pk[kb] += t * cb
            # predictor and corrector share the factor: factor once (in place), keep it
            rhs1 = Gt([D[i] * rp[i] - lam[i] for i in range(m)])       # G'(D rp - rc/s), rc = s lam
            rhs1 = [-rd[i] - rhs1[i] for i in range(n)]
            # factor + first solve
            (dz,) = _chol_solve(P, [rhs1])
            gdz = Gz(dz)
            ds = [-rp[i] - gdz[i] for i in range(m)]
            dl = [-(lam[i] * s[i] + lam[i] * ds[i]) / s[i] for i in range(m)]
            a = _maxstep(s, ds, lam, dl)
            mu_aff = mp.fsum((s[i] + a * ds[i]) * (lam[i] + a * dl[i]) for i in range(m)) / m
            sigma = (mu_aff / mu) ** 3
            rc = [s[i] * lam[i] + ds[i] * dl[i] - sigma * mu for i in range(m)]
            rhs2 = Gt([D[i] * rp[i] - rc[i] / s[i] for i in range(m)])
            rhs2 = [-rd[i] - rhs2[i] for i in range(n)]
            # second solve with the stored factor (P now holds L)
            y = list(rhs2)
            for i in range(n):
                y[i] = (y[i] - mp.fsum(P[i][k] * y[k] for k in range(i))) / P[i][i]
            for i in range(n - 1, -1, -1):
                y[i] = (y[i] - mp.fsum(P[k][i] * y[k] for k in range(i + 1, n))) / P[i][i]
            dz = y
            gdz = Gz(dz)
            ds = [-rp[i] - gdz[i] for i in range(m)]
            dl = [-(rc[i] + lam[i] * ds[i]) / s[i] for i in range(m)]
            a = min(mp.mpf(1), mp.mpf("0.995") * _maxstep(s, ds, lam, dl))
            z = [z[i] + a * dz[i] for i in range(n)]
            s = [s[i] + a * ds[i] for i in range(m)]
            lam = [lam[i] + a * dl[i] for i in range(m)]
        info = certify_mp(Hm, qm, rows, hm, z, lam, n, m)
        info.update(status=status, iters=it, digits=digits, lam=np.array([float(v) for v in lam]),
                    s=np.array([float(v) for v in s]))
        return np.array([float(v) for v in z]), info


def _maxstep(s, ds, lam, dl):
    a = mp.mpf(1)
    for x, dx in ((s, ds), (lam, dl)):
        for xi, di in zip(x, dx):
            if di < 0:
                t = -xi / di
                if t < a:
                    a = t
    return a


def certify_mp(Hm, qm, rows, hm, z, lam, n, m):
    """KKT residuals of (z, lam) in the working precision and the strong-convexity bound on |z - z*|_2:
    sigma e^2 <= |r| e + lam's  (oracle/qp.py: error_bound); sigma = lambda_min(H) from float64 (a lower bound is what
    matters: 0.9 of it is used)."""
    hz = [mp.fsum(Hm[i][k] * z[k] for k in range(n)) for i in range(n)]
    gl = [mp.mpf(0)] * n
    for r, wi in zip(rows, lam):
        for k, c in r:
            gl[k] += c * wi
    r_ = [hz[i] + qm[i] + gl[i] for i in range(n)]
    slack = [hm[i] - mp.fsum(c * z[k] for k, c in rows[i]) for i in range(m)]
    Hf = np.array([[float(v) for v in row] for row in Hm])
    sigma = mp.mpf(0.9 * float(np.linalg.eigvalsh(Hf)[0]))
    rn = mp.sqrt(mp.fsum(v * v for v in r_))
    gap = mp.fsum(lam[i] * max(slack[i], 0) for i in range(m))
    viol = max([mp.mpf(0)] + [-v for v in slack])
    lneg = max([mp.mpf(0)] + [-v for v in lam])
    bound = (rn + mp.sqrt(rn * rn + 4 * sigma * gap)) / (2 * sigma) if sigma > 0 else mp.inf
    return {"stationarity": float(rn), "gap": float(gap), "violation": float(viol), "lam_neg": float(lneg),
            "err_bound": float(bound)}


def solve_reference_qp(H, q, A, b, zmin, zmax, z0=None, digits=60, warm=True, verbose=False):
    """The reference QP (rows + variable bounds, oracle/qp.py: stack_constraints) in extended precision, started from
    oracle/qp.py's float64 interior-point iterate when `warm`."""
    from . import qp as qpo
    G, h = qpo.stack_constraints(A, b, zmin, zmax)
    z0_, lam0, s0 = z0, None, None
    if warm:
        zf, lamf, sf, _, st = qpo.ipm(H, q, G, h, z0, mu_tol=1e-9, res_tol=1e-8)
        if st != qpo.INFEASIBLE and np.all(np.isfinite(zf)):
            # back off into the interior: the float64 iterate is feasible to ~1e-9 only
            z0_, lam0, s0 = zf, np.maximum(lamf, 1e-6), np.maximum(h - G @ zf, 1e-6)
    return solve(H, q, G, h, z0=z0_, lam0=lam0, s0=s0, digits=digits, verbose=verbose)
