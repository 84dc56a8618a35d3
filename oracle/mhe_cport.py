"""CPU ORACLE (test infrastructure) -- ctypes binding of oracle/mhe_ref.c (`make -C oracle`): the linear
MovingHorizonEstimator period in the state-sequence (block-tridiagonal) form, OpenMP over estimators.
Only tests/ and bench.py's cpu_baseline leg may use it."""
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
        path = os.path.join(_HERE, "_build", "libmhe_ref.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "mhe_ref.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(path)
        L.mhe_ref_run.restype = C.c_int
        L.mhe_ref_run.argtypes = [C.c_int] * 5 + [C.c_void_p] * 8 + [C.c_int, C.c_double] + [C.c_void_p] * 3 + [C.c_int]
        L.mhe_ref_threads.restype = C.c_int
        _LIB = L
    return _LIB


def run(bt, Y, U, He, xabs=np.inf, nthreads=0):
    """`periods` estimator periods (current form) of the batch `bt` (synth.make_mhe_batch arrays: Ahat (B,nx̂,nx̂), Bhu,
    Chm, Qhat, Rhat, P0) on the data Y (periods,B,nym), U (periods,B,nu).  Returns xhat (periods,B,nx̂), iters, status."""
    T = lambda M: np.ascontiguousarray(np.asarray(M, float).transpose(0, 2, 1))       # column-major per problem
    B, nx, _ = bt["Ahat"].shape
    nu, nym = bt["Bhu"].shape[2], bt["Chm"].shape[1]
    periods = Y.shape[0]
    arrs = [T(bt["Ahat"]), T(bt["Bhu"]), T(bt["Chm"]), T(bt["Qhat"]), T(bt["Rhat"]), T(bt["P0"]),
            np.ascontiguousarray(Y, dtype=np.float64), np.ascontiguousarray(U, dtype=np.float64)]
    xhat = np.zeros((periods, B, nx))
    it = np.zeros((periods, B), np.int32)
    st = np.zeros((periods, B), np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().mhe_ref_run(B, nx, nu, nym, int(He), *[p(a) for a in arrs], periods, float(xabs), p(xhat), p(it), p(st), int(nthreads))
    if rc != 0:
        raise ValueError("mhe_ref_run: dimensions beyond the port's limits")
    return xhat, it, st


def threads():
    return lib().mhe_ref_threads()
