"""Shared helpers of the parity tests: run a synthetic batch through the C-ABI (HIP library, or
the CPU wave emulator when `lib` is given) and through the CPU oracle."""
from __future__ import annotations

import numpy as np

import mpcqp
from mpcqp import synth
from oracle import condense as cd, qp


def constraint_kwargs(cfg, oracle=False):
    names = (("umin", "umin"), ("umax", "umax"), ("dumin", "Δumin"), ("dumax", "Δumax"),
             ("ymin", "ymin"), ("ymax", "ymax"))
    kw = {}
    for k, kk in names:
        v = getattr(cfg, k)
        if np.isfinite(v):
            kw[k if oracle else kk] = np.full(cfg.nu if "u" in k else cfg.ny, v)
    return kw


def make_controller(cfg, bt, lib=None, **kw):
    mpc = mpcqp.BatchLinMPC(bt["Ahat"], bt["Bhu"], bt["Chat"], Hp=cfg.Hp, Hc=cfg.Hc, Cwt=cfg.Cwt,
                            Mwt=np.full(cfg.ny, cfg.Mwt), Nwt=np.full(cfg.nu, cfg.Nwt),
                            Lwt=np.full(cfg.nu, cfg.Lwt), lib=lib, **kw)
    mpc.setconstraint(**constraint_kwargs(cfg))
    return mpc


def run_batch(cfg, bt, lib=None, **kw):
    mpc = make_controller(cfg, bt, lib=lib, **kw)
    mpc.lastu0 = bt["lastu0"].copy()
    u = mpc.moveinput(bt["xhat0"], bt["ry"], want_info=True)
    info = mpc.getinfo()
    return {"Z": mpc.Z.copy(), "u": u, "status": mpc.status.copy(), "iters": mpc.iters.copy(),
            "Yhat": info["Ŷ"], "mpc": mpc}


def make_oracle(cfg, bt, i):
    m = cd.LinMPCOracle(bt["Ahat"][i], bt["Bhu"][i], bt["Chat"][i], Hp=cfg.Hp, Hc=cfg.Hc,
                        Cwt=cfg.Cwt, Mwt=np.full(cfg.ny, cfg.Mwt), Nwt=np.full(cfg.nu, cfg.Nwt),
                        Lwt=np.full(cfg.nu, cfg.Lwt))
    m.setconstraint(**constraint_kwargs(cfg, oracle=True))
    return m


def oracle_batch(cfg, bt):
    B = bt["xhat0"].shape[0]
    Z, U, cert, F, Q, H = [], [], [], [], [], []
    for i in range(B):
        m = make_oracle(cfg, bt, i)
        m.initpred(bt["xhat0"][i], bt["lastu0"][i], bt["ry"][i])
        m.linconstraint()
        z, st, info = qp.solve_qp(*m.qp_data(), m.warmstart(), return_info=True)
        Z.append(z)
        U.append(z[:cfg.nu] + bt["lastu0"][i])
        cert.append(info["certificate"] == "active-set")
        F.append(m.F)
        Q.append(m.qt)
        H.append(m.Ht)
    return {"Z": np.array(Z), "u": np.array(U), "certified": np.array(cert), "F": np.array(F),
            "q": np.array(Q), "H": np.array(H)}


def rel_err(Zg, Zo, nDU):
    """max_b ‖ΔU_gpu − ΔU_oracle‖∞ / max(1, ‖ΔU_oracle‖∞)  (BASELINE.md §4 'Parity')."""
    return np.max(np.abs(Zg[:, :nDU] - Zo[:, :nDU]), axis=1) / np.maximum(
        1.0, np.max(np.abs(Zo[:, :nDU]), axis=1))


def lqr_terminal_cost_case():
    """T6 (test/3_test_predictive_control.jl:498-527): terminal cost = DARE solution => LQR."""
    from scipy.linalg import solve_discrete_are
    A = np.array([[0.5, -0.4], [0.6, 0.5]]); Bu = np.eye(2); C = np.eye(2)
    Q, R = np.eye(2), 0.5 * np.eye(2)
    P = solve_discrete_are(A, Bu, Q, R)
    K = np.linalg.solve(R + Bu.T @ P @ Bu, Bu.T @ P @ A)
    M_Hp = np.block([[np.eye(4), np.zeros((4, 2))], [np.zeros((2, 4)), P]])
    return A, Bu, C, K, M_Hp


def run_lqr_terminal_cost(lib=None, B=3, steps=20):
    """Closed loop of T6 through the C-ABI (nint_ym = 0: the state is measured); returns the MPC
    and the LQR state trajectories, (2, steps) each."""
    A, Bu, C, K, M_Hp = lqr_terminal_cost_case()
    rep = lambda a: np.broadcast_to(a, (B,) + a.shape).copy()
    mpc = mpcqp.BatchLinMPC(rep(A), rep(Bu), rep(C), Hp=3, Hc=3, M_Hp=M_Hp, Nwt=[0, 0], Lwt=[0.5, 0.5], lib=lib)
    X_mpc, X_lqr = np.zeros((2, steps)), np.zeros((2, steps))
    x = np.array([1.0, 1.0])
    for i in range(steps):
        u = mpc.moveinput(np.tile(x, (B, 1)), [0.0, 0.0])
        assert np.all(mpc.status == 0) and np.abs(u - u[0]).max() == 0.0
        X_mpc[:, i] = x
        x = A @ x + Bu @ u[B - 1]
    x = np.array([1.0, 1.0])
    for i in range(steps):
        X_lqr[:, i] = x
        x = A @ x + Bu @ (-K @ x)
    return X_mpc, X_lqr
