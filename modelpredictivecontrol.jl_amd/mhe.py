"""Host-side mirror of the reference's linear MovingHorizonEstimator over the C-ABI of
include/mpcqp_mhe.h.

* `MheHandle` -- 1:1 ctypes binding of the `mpcqp_mhe_*` entry points.
* `BatchMHE`  -- B logical `MovingHorizonEstimator` objects on augmented models (Â, B̂u, Ĉm, B̂d, D̂dm) with
  the reference's vocabulary: constructor keywords of `MovingHorizonEstimator(model; He, σP_0, σQ, σR, Cwt,
  direct)` (/root/reference/src/estimator/mhe/construct.jl:255-460), `setconstraint!` keywords
  (construct.jl:858-1049), `preparestate!` / `updatestate!` (execute.jl:44-88), `getinfo` keys
  (execute.jl:116-200).  Operating points are kept on the host, as the reference's estimator fields do.

There is no CPU fallback: the shared library is the HIP build and every compute call needs a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api
from .api import MpcqpError, _chk, _f64, _ptr, colmajor

EXPORTS = ("mpcqp_mhe_create", "mpcqp_mhe_destroy", "mpcqp_mhe_set_model", "mpcqp_mhe_set_bounds", "mpcqp_mhe_set_bounds_window", "mpcqp_mhe_set_softness", "mpcqp_mhe_set_softness_window", "mpcqp_mhe_init", "mpcqp_mhe_set_state", "mpcqp_mhe_shift_windows",
           "mpcqp_mhe_prepare", "mpcqp_mhe_update", "mpcqp_mhe_prepare_device", "mpcqp_mhe_update_device",
           "mpcqp_mhe_sync", "mpcqp_mhe_get", "mpcqp_mhe_device_ptr", "mpcqp_mhe_nk", "mpcqp_mhe_last_ms",
           "mpcqp_mhe_register_columns")
KEEP_WINDOWS = 1
GET_XHAT0, GET_ZTILDE, GET_STATUS, GET_ITERS, GET_PBAR, GET_VHAT, GET_XHATWIN, GET_EPSILON = range(8)


class MheDims(C.Structure):
    _fields_ = [("batch", C.c_int32), ("nxhat", C.c_int32), ("nu", C.c_int32), ("nym", C.c_int32),
                ("nd", C.c_int32), ("He", C.c_int32), ("direct", C.c_int32), ("device", C.c_int32),
                ("flags", C.c_uint32), ("max_iter", C.c_int32), ("gap_tol", C.c_double),
                ("res_tol", C.c_double), ("dual_reg", C.c_double)]


def _bind(lib):
    if getattr(lib, "_mhe_bound", False):
        return lib
    lib.mpcqp_mhe_create.argtypes = [C.POINTER(MheDims), C.POINTER(C.c_void_p)]
    lib.mpcqp_mhe_destroy.argtypes = [C.c_void_p]
    lib.mpcqp_mhe_set_model.argtypes = [C.c_void_p] * 9
    lib.mpcqp_mhe_set_bounds.argtypes = [C.c_void_p] * 7
    lib.mpcqp_mhe_set_bounds_window.argtypes = [C.c_void_p] * 7
    lib.mpcqp_mhe_set_softness.argtypes = [C.c_void_p] * 8
    lib.mpcqp_mhe_set_softness_window.argtypes = [C.c_void_p] * 8
    lib.mpcqp_mhe_init.argtypes = [C.c_void_p] * 5
    lib.mpcqp_mhe_set_state.argtypes = [C.c_void_p] * 2
    lib.mpcqp_mhe_shift_windows.argtypes = [C.c_void_p] * 5
    lib.mpcqp_mhe_prepare.argtypes = [C.c_void_p] * 3
    lib.mpcqp_mhe_update.argtypes = [C.c_void_p] * 4
    lib.mpcqp_mhe_prepare_device.argtypes = [C.c_void_p] * 3
    lib.mpcqp_mhe_update_device.argtypes = [C.c_void_p] * 4
    lib.mpcqp_mhe_sync.argtypes = [C.c_void_p]
    lib.mpcqp_mhe_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.mpcqp_mhe_device_ptr.restype = C.c_void_p
    lib.mpcqp_mhe_device_ptr.argtypes = [C.c_void_p, C.c_int]
    lib.mpcqp_mhe_nk.argtypes = [C.c_void_p]
    lib.mpcqp_mhe_last_ms.restype = C.c_double
    lib.mpcqp_mhe_last_ms.argtypes = [C.c_void_p]
    lib.mpcqp_mhe_register_columns.argtypes = [C.c_void_p]
    lib._mhe_bound = True
    return lib


class MheHandle:
    """ctypes binding of include/mpcqp_mhe.h; arrays are (B, n) NumPy (= the ABI's (n,B))."""

    def __init__(self, B, nxhat, nu, nym, nd, He, direct=True, device=0, flags=0, max_iter=0, gap_tol=0.0,
                 res_tol=0.0, dual_reg=0.0, lib=None):
        self.lib = _bind(lib or api.load_library())
        self.B, self.nx, self.nu, self.nym, self.nd, self.He, self.direct = B, nxhat, nu, nym, nd, He, bool(direct)
        dims = MheDims(B, nxhat, nu, nym, nd, He, 1 if direct else 0, device, flags, max_iter, gap_tol, res_tol, dual_reg)
        self._h = C.c_void_p()
        _chk(self.lib, self.lib.mpcqp_mhe_create(C.byref(dims), C.byref(self._h)))
        self.flags = flags

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.mpcqp_mhe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_model(self, Ahat, Bhu, Chm, Bhd, Dhdm, fx, Qhat, Rhat):
        """(B,nx̂,nx̂) (B,nx̂,nu) (B,nym,nx̂) (B,nx̂,nd)|None (B,nym,nd)|None (B,nx̂)|None (B,nx̂,nx̂) (B,nym,nym)"""
        arrs = [colmajor(a) if a is not None else None for a in (Ahat, Bhu, Chm, Bhd, Dhdm)]
        arrs += [_f64(fx) if fx is not None else None, colmajor(Qhat), colmajor(Rhat)]
        _chk(self.lib, self.lib.mpcqp_mhe_set_model(self._h, *[_ptr(a) for a in arrs]))

    def set_bounds(self, xmin=None, xmax=None, wmin=None, wmax=None, vmin=None, vmax=None):
        arrs = [None if a is None else _f64(a) for a in (xmin, xmax, wmin, wmax, vmin, vmax)]
        _chk(self.lib, self.lib.mpcqp_mhe_set_bounds(self._h, *[_ptr(a) for a in arrs]))

    def set_bounds_window(self, Xmin=None, Xmax=None, Wmin=None, Wmax=None, Vmin=None, Vmax=None):
        """Window-long bounds: (B, (He+1) nx̂) [arrival; window oldest first], (B, He nx̂), (B, He nym); None = absent."""
        arrs = [None if a is None else _f64(a) for a in (Xmin, Xmax, Wmin, Wmax, Vmin, Vmax)]
        _chk(self.lib, self.lib.mpcqp_mhe_set_bounds_window(self._h, *[_ptr(a) for a in arrs]))

    def set_softness(self, Cwt=None, c_xmin=None, c_xmax=None, c_wmin=None, c_wmax=None, c_vmin=None, c_vmax=None):
        """Cwt (B,) finite or None (= Inf, hard only); softness arrays (B, n) >= 0 or None."""
        arrs = [None if a is None else _f64(a) for a in (Cwt, c_xmin, c_xmax, c_wmin, c_wmax, c_vmin, c_vmax)]
        _chk(self.lib, self.lib.mpcqp_mhe_set_softness(self._h, *[_ptr(a) for a in arrs]))

    def init(self, xhat0, P0, d0_prev=None, lastu0=None):
        arrs = [None if xhat0 is None else _f64(xhat0), colmajor(P0), None if d0_prev is None else _f64(d0_prev),
                None if lastu0 is None else _f64(lastu0)]
        _chk(self.lib, self.lib.mpcqp_mhe_init(self._h, *[_ptr(a) for a in arrs]))

    def set_state(self, xhat0):
        _chk(self.lib, self.lib.mpcqp_mhe_set_state(self._h, _ptr(_f64(xhat0))))

    def set_softness_window(self, Cwt, C_xmin=None, C_xmax=None, C_wmin=None, C_wmax=None, C_vmin=None, C_vmax=None):
        """Window-long softness: Cwt (B,) finite; (B, (He+1) nx̂) [arrival; window oldest first], (B, He nx̂), (B, He nym); None = 0."""
        arrs = [None if a is None else _f64(a) for a in (Cwt, C_xmin, C_xmax, C_wmin, C_wmax, C_vmin, C_vmax)]
        _chk(self.lib, self.lib.mpcqp_mhe_set_softness_window(self._h, *[_ptr(a) for a in arrs]))

    def shift_windows(self, dy0m=None, du0=None, dd0=None, dx0=None):
        arrs = [None if a is None else _f64(a) for a in (dy0m, du0, dd0, dx0)]
        _chk(self.lib, self.lib.mpcqp_mhe_shift_windows(self._h, *[_ptr(a) for a in arrs]))

    def _failed(self, rc):
        if rc < 0:
            _chk(self.lib, rc)
        return rc

    def prepare(self, y0m, d0=None):
        y, d = _f64(y0m), (None if d0 is None or self.nd == 0 else _f64(d0))
        return self._failed(self.lib.mpcqp_mhe_prepare(self._h, _ptr(y), _ptr(d)))

    def update(self, u0, y0m=None, d0=None):
        u = None if self.nu == 0 else _f64(u0)
        y = None if y0m is None else _f64(y0m)
        d = None if d0 is None or self.nd == 0 else _f64(d0)
        return self._failed(self.lib.mpcqp_mhe_update(self._h, _ptr(u), _ptr(y), _ptr(d)))

    def prepare_device(self, y0m, d0=0):
        _chk(self.lib, self.lib.mpcqp_mhe_prepare_device(self._h, y0m, d0))

    def update_device(self, u0, y0m=0, d0=0):
        _chk(self.lib, self.lib.mpcqp_mhe_update_device(self._h, u0, y0m, d0))

    def sync(self):
        _chk(self.lib, self.lib.mpcqp_mhe_sync(self._h))

    def get(self, what):
        B, nx, He, nym = self.B, self.nx, self.He, self.nym
        shape, dt = {GET_XHAT0: ((B, nx), np.float64), GET_ZTILDE: ((B, nx + He * nx), np.float64),
                     GET_STATUS: ((B,), np.int32), GET_ITERS: ((B,), np.int32), GET_PBAR: ((B, nx, nx), np.float64),
                     GET_VHAT: ((B, He * nym), np.float64), GET_XHATWIN: ((B, He * nx), np.float64),
                     GET_EPSILON: ((B,), np.float64)}[what]
        out = np.zeros(shape, dt)
        _chk(self.lib, self.lib.mpcqp_mhe_get(self._h, what, _ptr(out)))
        return out.transpose(0, 2, 1).copy() if what == GET_PBAR else out

    def device_ptr(self, what):
        return self.lib.mpcqp_mhe_device_ptr(self._h, what)

    @property
    def Nk(self):
        return self.lib.mpcqp_mhe_nk(self._h)

    def last_ms(self):
        return self.lib.mpcqp_mhe_last_ms(self._h)

    def register_columns(self):
        return self.lib.mpcqp_mhe_register_columns(self._h)


def _diag_cov(sig, B, n, name):
    s = np.asarray(sig, float)
    if s.ndim == 1:
        s = np.broadcast_to(s, (B, n))
    if s.shape != (B, n):
        raise ValueError(f"{name} size {s.shape} ≠ (B, n) = ({B}, {n})")
    if np.any(s < 0):
        raise ValueError(f"{name}: standard deviations must be ≥ 0")
    out = np.zeros((B, n, n))
    idx = np.arange(n)
    out[:, idx, idx] = s ** 2
    return out


class BatchMHE:
    """B independent `MovingHorizonEstimator`s (LinModel) of identical dimensions on one GPU.

    Ahat (B,nx̂,nx̂), Bhu (B,nx̂,nu), Chm (B,nym,nx̂) [rows i_ym of Ĉ], Bhd (B,nx̂,nd), Dhdm (B,nym,nd): the
    augmented model of `augment_model` (src/estimator/construct.jl).  Covariances as in the reference:
    `σP_0`, `σQ`, `σR` standard deviations of the AUGMENTED state / measured outputs (vectors, shared or
    (B,n)), or full matrices `P̂_0`, `Q̂`, `R̂` ((B,n,n)).  `Cwt` = Inf (default): hard constraints only; a finite
    `Cwt` adds the slack ε and enables the softness keywords `c_x̂min … c_v̂max` of `setconstraint`."""

    def __init__(self, Ahat, Bhu, Chm, Bhd=None, Dhdm=None, *, He, σP_0=None, σQ=None, σR=None, P̂_0=None, Q̂=None,
                 R̂=None, Cwt=np.inf, direct=True, uop=None, yop_m=None, dop=None, x̂op=None, f̂op=None, device=0,
                 keep_windows=True, lib=None, **solver):
        Ahat = np.asarray(Ahat, float)
        if Ahat.ndim != 3 or Ahat.shape[1] != Ahat.shape[2]:
            raise ValueError("Ahat must be (B, nx̂, nx̂)")
        B, nx = Ahat.shape[:2]
        Bhu = np.zeros((B, nx, 0)) if Bhu is None else np.asarray(Bhu, float)
        Chm = np.asarray(Chm, float)
        nu, nym = Bhu.shape[2], Chm.shape[1]
        Bhd = np.zeros((B, nx, 0)) if Bhd is None else np.asarray(Bhd, float)
        nd = Bhd.shape[2]
        Dhdm = np.zeros((B, nym, nd)) if Dhdm is None else np.asarray(Dhdm, float)
        if Bhu.shape != (B, nx, nu) or Chm.shape != (B, nym, nx) or Bhd.shape != (B, nx, nd) or Dhdm.shape != (B, nym, nd):
            raise ValueError("model matrices have inconsistent sizes")
        if not isinstance(He, (int, np.integer)) or He < 1:
            raise ValueError("Estimation horizon He should be ≥ 1")          # construct.jl:436
        if Cwt < 0:
            raise ValueError("Cwt weight should be ≥ 0")                       # construct.jl:437
        self.Cwt = float(Cwt)
        self.nϵ = 0 if np.isinf(Cwt) else 1
        self.B, self.nx̂, self.nu, self.nym, self.nd, self.He, self.direct = B, nx, nu, nym, nd, int(He), bool(direct)
        self.Q̂ = np.asarray(Q̂, float) if Q̂ is not None else _diag_cov(np.ones(nx) if σQ is None else σQ, B, nx, "σQ")
        self.R̂ = np.asarray(R̂, float) if R̂ is not None else _diag_cov(np.ones(nym) if σR is None else σR, B, nym, "σR")
        self.P̂_0 = np.asarray(P̂_0, float) if P̂_0 is not None else _diag_cov(np.ones(nx) if σP_0 is None else σP_0, B, nx, "σP_0")
        for M, n, name in ((self.Q̂, nx, "Q̂"), (self.R̂, nym, "R̂"), (self.P̂_0, nx, "P̂_0")):
            if M.shape != (B, n, n):
                raise ValueError(f"{name} size {M.shape} ≠ ({B}, {n}, {n})")
        z = lambda v, n: np.zeros((B, n)) if v is None else np.broadcast_to(np.asarray(v, float), (B, n)).copy()
        self.uop, self.yop_m, self.dop, self.x̂op, self.f̂op = z(uop, nu), z(yop_m, nym), z(dop, nd), z(x̂op, nx), z(f̂op, nx)
        self._Ahat, self._Bhu, self._Chm, self._Bhd, self._Dhdm = Ahat, Bhu, Chm, Bhd, Dhdm
        self.handle = MheHandle(B, nx, nu, nym, nd, self.He, direct=direct, device=device,
                                flags=KEEP_WINDOWS if keep_windows else 0, lib=lib, **solver)
        self.handle.set_model(Ahat, Bhu if nu else None, Chm, Bhd if nd else None, Dhdm if nd else None,
                              self.f̂op - self.x̂op, self.Q̂, self.R̂)
        self._con, self._soft = {}, {}
        if self.nϵ:
            self.handle.set_softness(np.full(B, self.Cwt))
        self.x̂0 = np.zeros((B, nx))
        self.handle.init(self.x̂0, self.P̂_0)
        self.status = np.zeros(B, np.int32)

    # -- setconstraint! (construct.jl:858-1049): per-channel hard bounds --------------------------------
    def setconstraint(self, *, x̂min=None, x̂max=None, ŵmin=None, ŵmax=None, v̂min=None, v̂max=None, c_x̂min=None, c_x̂max=None,
                      c_ŵmin=None, c_ŵmax=None, c_v̂min=None, c_v̂max=None, **other):
        cwin = {k: other.pop(k) for k in ("C_x̂min", "C_x̂max", "C_ŵmin", "C_ŵmax", "C_v̂min", "C_v̂max") if k in other}
        win = {k: other.pop(k) for k in ("X̂min", "X̂max", "Ŵmin", "Ŵmax", "V̂min", "V̂max") if k in other}
        if other:
            raise TypeError(f"unknown setconstraint keywords {sorted(other)}")
        B = self.B
        if win or getattr(self, "_win", None):
            self._setconstraint_window(win, dict(x̂min=x̂min, x̂max=x̂max, ŵmin=ŵmin, ŵmax=ŵmax, v̂min=v̂min, v̂max=v̂max))
            x̂min = x̂max = ŵmin = ŵmax = v̂min = v̂max = None
        con = dict(self._con)              # (validated before it replaces the current set)
        for key, val, n, shift in (("xmin", x̂min, self.nx̂, self.x̂op), ("xmax", x̂max, self.nx̂, self.x̂op),
                                   ("wmin", ŵmin, self.nx̂, None), ("wmax", ŵmax, self.nx̂, None),
                                   ("vmin", v̂min, self.nym, None), ("vmax", v̂max, self.nym, None)):
            if val is None:
                continue
            v = np.asarray(val, float)
            if v.shape not in ((n,), (B, n)):
                raise ValueError(f"{key} size {v.shape} ≠ ({n},) or ({B}, {n})")       # DimensionMismatch
            v = np.broadcast_to(v, (B, n)).copy()
            con[key] = v - shift if shift is not None else v
        for lo, hi in (("xmin", "xmax"), ("wmin", "wmax"), ("vmin", "vmax")):
            if lo in con and hi in con and np.any(con[lo] > con[hi]):
                raise ValueError(f"{lo} > {hi}: infeasible bounds")
        self._con = con
        if not getattr(self, "_win", None):
            self.handle.set_bounds(**self._con)
        # softness parameters (construct.jl:960-1020): nonnegative, and only with a finite Cwt
        for key, val, n in (("c_xmin", c_x̂min, self.nx̂), ("c_xmax", c_x̂max, self.nx̂), ("c_wmin", c_ŵmin, self.nx̂),
                            ("c_wmax", c_ŵmax, self.nx̂), ("c_vmin", c_v̂min, self.nym), ("c_vmax", c_v̂max, self.nym)):
            if val is None:
                continue
            v = np.asarray(val, float)
            if v.shape not in ((n,), (B, n)):
                raise ValueError(f"{key} size {v.shape} ≠ ({n},) or ({B}, {n})")
            if np.any(v < 0):
                raise ValueError(f"{key} weights should be non-negative")
            if not self.nϵ:
                raise ValueError("Slack variable weight Cwt must be finite to set softness parameters")
            self._soft[key] = np.broadcast_to(v, (B, n)).copy()
        if cwin or getattr(self, "_cwin", None):
            self._setconstraint_softness_window(cwin, dict(c_x̂min=c_x̂min, c_x̂max=c_x̂max, c_ŵmin=c_ŵmin, c_ŵmax=c_ŵmax,
                                                           c_v̂min=c_v̂min, c_v̂max=c_v̂max))
        elif self.nϵ:
            self.handle.set_softness(np.full(B, self.Cwt), **self._soft)
        return self

    def _setconstraint_softness_window(self, cwin, chan):
        """Window-long softness C_x̂min ... C_v̂max (construct.jl:937-1020): (n (He+1),) / (n He,) vectors (or (B, .)) >= 0; a
        per-channel keyword of the same or a later call fills its whole vector (`repeat(c_x̂min, He+1)`, construct.jl:958-963).
        Once a window-long vector has been given the estimator stays on window-long softness."""
        B, nx, nym, He = self.B, self.nx̂, self.nym, self.He
        if not self.nϵ:
            raise ValueError("Slack variable weight Cwt must be finite to set softness parameters")
        cur = dict(getattr(self, "_cwin", None) or {})
        spec = {"C_x̂min": (nx, He + 1, "c_x̂min", "c_xmin"), "C_x̂max": (nx, He + 1, "c_x̂max", "c_xmax"),
                "C_ŵmin": (nx, He, "c_ŵmin", "c_wmin"), "C_ŵmax": (nx, He, "c_ŵmax", "c_wmax"),
                "C_v̂min": (nym, He, "c_v̂min", "c_vmin"), "C_v̂max": (nym, He, "c_v̂max", "c_vmax")}
        for K, (n, nblk, k, old) in spec.items():
            if K in cwin and cwin[K] is not None:
                v = np.asarray(cwin[K], float)
                if v.shape not in ((n * nblk,), (B, n * nblk)):
                    raise ValueError(f"{K} size must be ({n * nblk},)")                     # DimensionMismatch
                if np.any(v < 0):
                    raise ValueError(f"{K} weights should be non-negative")
                cur[K] = np.broadcast_to(v, (B, n * nblk)).copy()
            elif chan.get(k) is not None:
                cur[K] = np.tile(self._soft[old], (1, nblk))
            elif K not in cur:          # first window-long call: start from the per-channel softness in force
                cur[K] = np.tile(self._soft[old], (1, nblk)) if old in self._soft else np.zeros((B, n * nblk))
        self._cwin = cur
        self.handle.set_softness_window(np.full(B, self.Cwt), cur["C_x̂min"], cur["C_x̂max"], cur["C_ŵmin"], cur["C_ŵmax"],
                                        cur["C_v̂min"], cur["C_v̂max"])

    def _setconstraint_window(self, win, chan):
        """Window-long bounds X̂min ... V̂max (construct.jl:858-935): (n (He+1),) / (n He,) vectors (or (B, .)), a bound per
        channel and stage; a per-channel keyword given in the same or a later call fills its whole vector like the
        reference does.  Once a window-long vector has been given the estimator stays on window-long bounds."""
        B, nx, nym, He = self.B, self.nx̂, self.nym, self.He
        cur = dict(getattr(self, "_win", None) or {})
        spec = {"X̂min": (nx, He + 1, -np.inf, "x̂min", "xmin"), "X̂max": (nx, He + 1, np.inf, "x̂max", "xmax"),
                "Ŵmin": (nx, He, -np.inf, "ŵmin", "wmin"), "Ŵmax": (nx, He, np.inf, "ŵmax", "wmax"),
                "V̂min": (nym, He, -np.inf, "v̂min", "vmin"), "V̂max": (nym, He, np.inf, "v̂max", "vmax")}
        for K, (n, nblk, dflt, k, old) in spec.items():
            shift = np.tile(self.x̂op, (1, nblk)) if K[0] == "X" else 0.0
            if K in win and win[K] is not None:
                v = np.asarray(win[K], float)
                if v.shape not in ((n * nblk,), (B, n * nblk)):
                    raise ValueError(f"{K} size {v.shape} ≠ ({n * nblk},)")                 # DimensionMismatch
                cur[K] = np.broadcast_to(v, (B, n * nblk)) - shift
            elif chan.get(k) is not None:
                v = np.asarray(chan[k], float)
                if v.shape not in ((n,), (B, n)):
                    raise ValueError(f"{k} size {v.shape} ≠ ({n},) or ({B}, {n})")
                cur[K] = np.tile(np.broadcast_to(v, (B, n)), (1, nblk)) - shift
            elif K not in cur:          # first window-long call: start from the per-channel bounds in force
                cur[K] = np.tile(self._con[old], (1, nblk)) if old in self._con else np.full((B, n * nblk), dflt)
        for lo, hi in (("X̂min", "X̂max"), ("Ŵmin", "Ŵmax"), ("V̂min", "V̂max")):
            if np.any(cur[lo] > cur[hi]):
                raise ValueError(f"{lo} > {hi}: infeasible bounds")
        self._win = cur
        self.handle.set_bounds_window(cur["X̂min"], cur["X̂max"], cur["Ŵmin"], cur["Ŵmax"], cur["V̂min"], cur["V̂max"])

    def setstate(self, x̂, P̂=None):
        """setstate!(estim, x̂) (src/estimator/execute.jl:424-429): only the current estimate changes; the data windows
        and the arrival covariance stay.  A covariance is an error, as in the reference (mhe/execute.jl:938-941)."""
        if P̂ is not None:
            raise MpcqpError("MovingHorizonEstimator does not compute an estimation covariance matrix P̂.")
        x = np.asarray(x̂, float)
        if x.shape not in ((self.nx̂,), (self.B, self.nx̂)):
            raise ValueError(f"x̂ size must be ({self.nx̂},)")
        self.x̂0 = np.broadcast_to(x, (self.B, self.nx̂)) - self.x̂op
        self.handle.set_state(self.x̂0)
        return self

    def setmodel(self, Ahat=None, Bhu=None, Chm=None, Bhd=None, Dhdm=None, *, uop=None, yop_m=None, dop=None, x̂op=None,
                 f̂op=None, Q̂=None, R̂=None):
        """setmodel!(estim, model; Q̂, R̂) (src/estimator/execute.jl:483-497, mhe/execute.jl:943-1046): new augmented model
        matrices and / or operating points and / or covariances.  The data windows, lastu0, x̂0, x̂0arr and the x̂ bounds
        are deviation variables: they keep their ENGINEERING values, i.e. move by (old − new) operating point."""
        B, nx = self.B, self.nx̂
        z = lambda v, n, old: old if v is None else np.broadcast_to(np.asarray(v, float), (B, n)).copy()
        uop, yop_m, dop = z(uop, self.nu, self.uop), z(yop_m, self.nym, self.yop_m), z(dop, self.nd, self.dop)
        xop, fop = z(x̂op, nx, self.x̂op), z(f̂op, nx, self.f̂op)
        for M, n, name in ((Q̂, nx, "Q̂"), (R̂, self.nym, "R̂")):
            if M is not None:
                M = np.asarray(M, float)
                if M.shape != (B, n, n):
                    raise ValueError(f"{name} size {M.shape} ≠ ({B}, {n}, {n})")
                if np.any(np.linalg.eigvalsh(M) <= 0):
                    raise MpcqpError(f"{name} is not positive definite")
        pick = lambda new, old: old if new is None else np.asarray(new, float)
        self._Ahat, self._Bhu, self._Chm = pick(Ahat, self._Ahat), pick(Bhu, self._Bhu), pick(Chm, self._Chm)
        self._Bhd, self._Dhdm = pick(Bhd, self._Bhd), pick(Dhdm, self._Dhdm)
        self.Q̂, self.R̂ = pick(Q̂, self.Q̂), pick(R̂, self.R̂)
        dx = self.x̂op - xop
        self.handle.shift_windows(self.yop_m - yop_m, (self.uop - uop) if self.nu else None,
                                  (self.dop - dop) if self.nd else None, dx)
        self.x̂0 = self.x̂0 + dx
        self.uop, self.yop_m, self.dop, self.x̂op, self.f̂op = uop, yop_m, dop, xop, fop
        self.handle.set_model(self._Ahat, self._Bhu if self.nu else None, self._Chm, self._Bhd if self.nd else None,
                              self._Dhdm if self.nd else None, self.f̂op - self.x̂op, self.Q̂, self.R̂)
        for k in ("xmin", "xmax"):                   # x̂ bounds: same engineering values, new deviation values
            if k in self._con:
                self._con[k] = self._con[k] + dx
        if getattr(self, "_win", None):
            for K in ("X̂min", "X̂max"):
                self._win[K] = self._win[K] + np.tile(dx, (1, self.He + 1))
            w = self._win
            self.handle.set_bounds_window(w["X̂min"], w["X̂max"], w["Ŵmin"], w["Ŵmax"], w["V̂min"], w["V̂max"])
        elif self._con:
            self.handle.set_bounds(**self._con)
        return self

    def initstate(self, x̂, u=None, d=None):
        """init_estimate_cov! (execute.jl:2-36) with the estimate x̂: windows emptied, P̄ = P̂_0, lastu0 = u - uop."""
        self.x̂0 = np.broadcast_to(np.asarray(x̂, float), (self.B, self.nx̂)) - self.x̂op
        lu = None if u is None else np.broadcast_to(np.asarray(u, float), (self.B, self.nu)) - self.uop
        d0 = None if d is None or self.nd == 0 else np.broadcast_to(np.asarray(d, float), (self.B, self.nd)) - self.dop
        self.handle.init(self.x̂0, self.P̂_0, d0, lu)
        return self

    def _bc(self, v, n, name):
        a = np.asarray(v, float)
        if a.shape not in ((n,), (self.B, n)):
            raise ValueError(f"{name} size {a.shape} ≠ ({n},) or ({self.B}, {n})")
        return np.broadcast_to(a, (self.B, n))

    def _after_solve(self, nbad):
        self.status = self.handle.get(GET_STATUS)
        self.x̂0 = self.handle.get(GET_XHAT0)
        return nbad

    def preparestate(self, ym, d=None):
        """preparestate!(estim, ym, d) -> x̂ (B,nx̂)"""
        y0 = self._bc(ym, self.nym, "ym") - self.yop_m
        d0 = None if self.nd == 0 else self._bc(d, self.nd, "d") - self.dop
        if self.direct:
            self._after_solve(self.handle.prepare(y0, d0))
        return self.x̂0 + self.x̂op

    def updatestate(self, u, ym, d=None):
        """updatestate!(estim, u, ym, d) -> x̂ (B,nx̂) of the next period"""
        u0 = self._bc(u, self.nu, "u") - self.uop
        y0 = self._bc(ym, self.nym, "ym") - self.yop_m
        d0 = None if self.nd == 0 else self._bc(d, self.nd, "d") - self.dop
        nbad = self.handle.update(u0, y0, d0)
        if not self.direct:
            self._after_solve(nbad)
        return self.x̂0 + self.x̂op

    def getinfo(self):
        """Keys of getinfo(estim) (execute.jl:116-200) that the batched path holds."""
        h = self.handle
        Nk, nx, He = h.Nk, self.nx̂, self.He
        Zt = h.get(GET_ZTILDE)
        info = {"Nk": Nk, "Ŵ": Zt[:, nx:nx + Nk * nx], "x̂arr": Zt[:, :nx] + self.x̂op,
                "ϵ": h.get(GET_EPSILON) if self.nϵ else np.zeros(self.B),
                "status": h.get(GET_STATUS), "iters": h.get(GET_ITERS), "P̄": h.get(GET_PBAR)}
        if h.flags & KEEP_WINDOWS:
            info["V̂"] = h.get(GET_VHAT)[:, :Nk * self.nym]
            info["X̂"] = h.get(GET_XHATWIN)[:, :Nk * nx] + np.tile(self.x̂op, (1, Nk))
        for k, alias in (("Ŵ", "What"), ("x̂arr", "xhatarr"), ("V̂", "Vhat"), ("X̂", "Xhat"), ("P̄", "Pbar"), ("ϵ", "epsilon")):
            if k in info:
                info[alias] = info[k]
        return info
