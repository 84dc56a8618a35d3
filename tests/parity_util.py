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
