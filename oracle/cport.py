"""CPU ORACLE (test infrastructure) -- ctypes binding of oracle/linmpc_ref.c (`make -C oracle`).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liblinmpc_ref.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(path)
        L.linmpc_ref_create.restype = C.c_void_p
        L.linmpc_ref_create.argtypes = [C.c_int] * 6 + [C.c_void_p, C.c_int] + [C.c_void_p] * 13
        L.linmpc_ref_destroy.argtypes = [C.c_void_p]
        L.linmpc_ref_step.restype = C.c_int
        L.linmpc_ref_step.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_double, C.c_double,
                                                        C.c_double, C.c_int]
        L.linmpc_ref_threads.restype = C.c_int
        L.linmpc_ref_polish_params.argtypes = [C.c_double, C.c_double, C.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RefBatch:
    """B dense CPU controllers; arrays in the C-ABI layout (column-major per problem)."""

    def __init__(self, Ahat, Bu, Cm, Mdiag, Ndiag, Ldiag, Cwt, Hp, Hc, nb=None, neps=1,
                 U0min=None, U0max=None, DUmin=None, DUmax=None, Y0min=None, Y0max=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        self.B, self.nu, self.nxh = Bu.shape[0], Bu.shape[1], Bu.shape[2]     # (B, nu, nxh) ABI
        self.ny = Cm.shape[2]
        self.Hp, self.Hc, self.neps = Hp, Hc, neps
        self.nZ = self.nu * Hc + neps
        self._keep = [f(a) for a in (Ahat, Bu, Cm, Mdiag, Ndiag, Ldiag, Cwt, U0min, U0max, DUmin,
                                     DUmax, Y0min, Y0max)]
        nbv = None if nb is None else np.ascontiguousarray(nb, dtype=np.int32)
        self._nb = nbv
        self.h = lib().linmpc_ref_create(self.B, self.nxh, self.nu, self.ny, Hp, Hc, _p(nbv), neps,
                                         *[_p(a) for a in self._keep])

    def step(self, xhat0, lastu0, ry, Z=None, cold=True, nthreads=0, gap_tol=1e-12, res_tol=1e-11,
             delta=1e-12, max_iter=100):
        B = self.B
        x, lu, r = (np.ascontiguousarray(a, dtype=np.float64) for a in (xhat0, lastu0, ry))
        Z = np.zeros((B, self.nZ)) if Z is None else Z
        u0 = np.empty((B, self.nu))
        st = np.empty(B, np.int32)
        it = np.empty(B, np.int32)
        lib().linmpc_ref_step(self.h, _p(x), _p(lu), _p(r), _p(Z), _p(u0), _p(st), _p(it),
                              int(cold), int(nthreads), gap_tol, res_tol, delta, max_iter)
        return Z, u0, st, it

    def __del__(self):
        try:
            lib().linmpc_ref_destroy(self.h)
        except Exception:
            pass


def from_synth(cfg, bt):
    """RefBatch for a synthetic batch of modelpredictivecontrol.jl_amd/synth.py."""
    B = bt["Ahat"].shape[0]
    T = lambda M: np.ascontiguousarray(M.transpose(0, 2, 1))
    full = lambda v, n: None if not np.isfinite(v) else np.full((B, n), float(v))
    nu, ny, Hp, Hc = cfg.nu, cfg.ny, cfg.Hp, cfg.Hc
    return RefBatch(T(bt["Ahat"]), T(bt["Bhu"]), T(bt["Chat"]), np.full((B, ny * Hp), cfg.Mwt),
                    np.full((B, nu * Hc), cfg.Nwt), np.full((B, nu * Hp), cfg.Lwt),
                    np.full(B, cfg.Cwt), Hp, Hc, neps=0 if np.isinf(cfg.Cwt) else 1,
                    U0min=full(cfg.umin, nu * Hp), U0max=full(cfg.umax, nu * Hp),
                    DUmin=full(cfg.dumin, nu * Hc), DUmax=full(cfg.dumax, nu * Hc),
                    Y0min=full(cfg.ymin, ny * Hp), Y0max=full(cfg.ymax, ny * Hp))
