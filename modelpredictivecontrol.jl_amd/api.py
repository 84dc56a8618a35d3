"""Host-side mirror of the reference's LinMPC interface over the C-ABI of include/mpcqp.h.

Two layers:

* `Handle` -- 1:1 ctypes binding of the C entry points (what a Julia `ccall` shim binds, see
  INTEGRATION.md).  Works with host (NumPy) pointers or raw device pointers.
* `BatchLinMPC` -- B logical `LinMPC` controllers of identical dimensions with the reference's
  vocabulary: constructor keywords of `LinMPC(estim; Hp, Hc, Mwt, Nwt, Lwt, Cwt)`
  (/root/reference/src/controller/linmpc.jl:288-316), `setconstraint!` keywords
  (src/controller/construct.jl:324-350), `moveinput!` arguments (src/controller/execute.jl:59-70),
  `getinfo` keys (src/controller/execute.jl:145-198), and the same error behaviour
  (DimensionMismatch -> ValueError, ArgumentError -> ValueError, status policy of
  src/controller/execute.jl:482-503).

There is no CPU fallback: the shared library is the HIP build and every compute call needs a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libmpcqp.so")

# flags / codes of include/mpcqp.h
FLAG_RY_CONSTANT, FLAG_COLD_START, FLAG_KEEP_QP, FLAG_WARM_DUAL, FLAG_NO_POLISH, FLAG_KEEP_ITERATE = 1, 2, 4, 8, 16, 32
KERNEL_GENERIC, KERNEL_AOT, KERNEL_ONDEMAND, KERNEL_SMALL, KERNEL_MS = 0, 1, 2, 3, 4
SINGLE_SHOOTING, MULTIPLE_SHOOTING = 0, 1
STATUS_OPTIMAL, STATUS_ITERATION_LIMIT, STATUS_ERROR = 0, 1, 2
GET_HESSIAN, GET_STEPRESP, GET_KMAT, GET_BVEC, GET_QTILDE, GET_FVEC, GET_AUDIT, GET_XHAT_MS, GET_MS_DEFECT = 1, 2, 3, 4, 5, 6, 7, 8, 9
EXPORTS = ("mpcqp_version", "mpcqp_strerror", "mpcqp_last_hip_error", "mpcqp_create",
           "mpcqp_destroy", "mpcqp_get_sizes", "mpcqp_set_model", "mpcqp_set_weights",
           "mpcqp_set_bounds", "mpcqp_step", "mpcqp_step_device", "mpcqp_loop_device", "mpcqp_recondense_device",
           "mpcqp_get", "mpcqp_last_step_ms", "mpcqp_last_condense_ms", "mpcqp_last_predmat_ms", "mpcqp_kf_set",
           "mpcqp_kf_correct", "mpcqp_kf_predict", "mpcqp_kf_correct_device", "mpcqp_kf_predict_device",
           "mpcqp_set_output_weight_blocks", "mpcqp_set_dense_weights", "mpcqp_set_custom_constraints", "mpcqp_set_custom_bounds",
           "mpcqp_set_flags", "mpcqp_set_iteration_limit", "mpcqp_set_transcription", "mpcqp_transcription_supported", "mpcqp_set_current_setpoint", "mpcqp_prepare", "mpcqp_kernel_kind", "mpcqp_lds_bytes", "mpcqp_row_groups", "mpcqp_prebuild",
           "mpcqp_last_build_error", "mpcqp_multi_create", "mpcqp_multi_destroy", "mpcqp_multi_ndev",
           "mpcqp_multi_handle", "mpcqp_multi_shard", "mpcqp_multi_set_model", "mpcqp_multi_set_weights",
           "mpcqp_multi_set_bounds", "mpcqp_multi_prepare", "mpcqp_multi_step", "mpcqp_multi_gather_device",
           "mpcqp_multi_scatter_device")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class Dims(C.Structure):
    _fields_ = [("batch", C.c_int32), ("nxhat", C.c_int32), ("nu", C.c_int32), ("ny", C.c_int32),
                ("nd", C.c_int32), ("Hp", C.c_int32), ("Hc", C.c_int32), ("nb", _ip),
                ("neps", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("max_iter", C.c_int32), ("gap_tol", C.c_double), ("res_tol", C.c_double),
                ("dual_reg", C.c_double)]


class Sizes(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("nZ", "nDU", "nU", "nY", "nD")]


BOUND_FIELDS = ("U0min", "U0max", "DUmin", "DUmax", "Y0min", "Y0max", "x0min", "x0max",
                "C_umin", "C_umax", "C_dumin", "C_dumax", "C_ymin", "C_ymax", "c_x0min", "c_x0max")


class Bounds(C.Structure):
    _fields_ = [(k, _dp) for k in BOUND_FIELDS]


class MpcqpError(RuntimeError):
    pass


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so.7 /
    libhsa-runtime64.so.1 (same SONAMEs as /opt/rocm/lib, RPATH $ORIGIN); the dynamic loader keeps
    whichever copy is mapped first.  If this library pulls in /opt/rocm's copy first, a later
    `import torch` binds to it and reports "No HIP GPUs are available" (torch 2.10+rocm7.0 against
    the 7.2 runtime).  When torch is installed but not imported yet, map its copy first so that both
    orders of import give the same process image as `import torch; import mpcqp`.
    MPCQP_SYSTEM_HIP=1 keeps the system runtime (a process that never imports torch)."""
    import sys
    if "torch" in sys.modules or os.environ.get("MPCQP_SYSTEM_HIP") == "1":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)


def _check_hip_runtime(lib):
    """The library is built against the system ROCm; the process may run on another HIP runtime (the PyTorch wheel's copy,
    see _share_hip_runtime_with_torch).  A different MAJOR version is worth a warning."""
    import re
    import warnings
    m = re.search(r"HIP (\d+)\.(\d+)", lib.mpcqp_version().decode())
    try:
        rt = C.CDLL("libamdhip64.so")              # (already mapped: resolves to the copy this process uses)
        v = C.c_int(0)
        if not m or int(m.group(1)) == 0 or rt.hipRuntimeGetVersion(C.byref(v)) != 0:       # (0.0: the CPU emulator build)
            return
    except (OSError, AttributeError):
        return
    if v.value // 10000000 != int(m.group(1)):
        warnings.warn(f"mpcqp: built against HIP {m.group(1)}.{m.group(2)}, running on HIP runtime {v.value // 10000000}."
                      f"{v.value // 100000 % 100} (MPCQP_SYSTEM_HIP=1 keeps the system runtime in processes that never import torch)",
                      RuntimeWarning, stacklevel=3)


def load_library(path: str | None = None):
    """Load (once) the HIP shared library.  `path` is for tests only; the environment variable MPCQP_LIB names another build
    of the same library (developer A/B runs of bench.py, scripts/build_variant.sh)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("MPCQP_LIB") or DEFAULT_LIB
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built (run `python -c 'import "
            f"__graft_entry__ as g; g.build()'` or `make -C modelpredictivecontrol.jl_amd/csrc`). "
            "There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    lib.mpcqp_version.restype = C.c_char_p
    _check_hip_runtime(lib)
    lib.mpcqp_strerror.restype = C.c_char_p
    lib.mpcqp_strerror.argtypes = [C.c_int]
    lib.mpcqp_last_hip_error.restype = C.c_char_p
    lib.mpcqp_create.argtypes = [C.POINTER(Dims), C.POINTER(C.c_void_p)]
    lib.mpcqp_destroy.argtypes = [C.c_void_p]
    lib.mpcqp_get_sizes.argtypes = [C.c_void_p, C.POINTER(Sizes)]
    lib.mpcqp_set_model.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    lib.mpcqp_set_weights.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    lib.mpcqp_set_bounds.argtypes = [C.c_void_p, C.POINTER(Bounds)]
    lib.mpcqp_set_output_weight_blocks.argtypes = [C.c_void_p, C.c_void_p]
    lib.mpcqp_set_dense_weights.argtypes = [C.c_void_p] * 4
    lib.mpcqp_set_flags.argtypes = [C.c_void_p, C.c_uint32]
    lib.mpcqp_set_iteration_limit.argtypes = [C.c_void_p, C.c_int32]
    lib.mpcqp_set_transcription.argtypes = [C.c_void_p, C.c_int32]
    lib.mpcqp_transcription_supported.argtypes = [C.c_void_p]
    lib.mpcqp_set_custom_constraints.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    lib.mpcqp_set_custom_bounds.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    lib.mpcqp_step.argtypes = [C.c_void_p] + [C.c_void_p] * 11
    lib.mpcqp_step_device.argtypes = [C.c_void_p] + [C.c_void_p] * 12
    lib.mpcqp_loop_device.argtypes = [C.c_void_p] + [C.c_void_p] * 13
    lib.mpcqp_recondense_device.argtypes = [C.c_void_p, C.c_void_p]
    lib.mpcqp_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.mpcqp_last_step_ms.restype = C.c_double
    lib.mpcqp_last_step_ms.argtypes = [C.c_void_p]
    lib.mpcqp_last_condense_ms.restype = C.c_double
    lib.mpcqp_last_condense_ms.argtypes = [C.c_void_p]
    lib.mpcqp_last_predmat_ms.restype = C.c_double
    lib.mpcqp_last_predmat_ms.argtypes = [C.c_void_p]
    lib.mpcqp_kf_set.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.mpcqp_kf_correct.argtypes = [C.c_void_p] * 4
    lib.mpcqp_kf_predict.argtypes = [C.c_void_p] * 4
    lib.mpcqp_kf_correct_device.argtypes = [C.c_void_p] * 5
    lib.mpcqp_kf_predict_device.argtypes = [C.c_void_p] * 5
    lib.mpcqp_set_current_setpoint.argtypes = [C.c_void_p, C.c_void_p]
    lib.mpcqp_prepare.argtypes = [C.c_void_p]
    lib.mpcqp_kernel_kind.argtypes = [C.c_void_p]
    lib.mpcqp_lds_bytes.argtypes = [C.c_void_p]
    lib.mpcqp_row_groups.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    lib.mpcqp_prebuild.argtypes = [C.POINTER(Dims), C.c_uint32]
    lib.mpcqp_last_build_error.restype = C.c_char_p
    lib.mpcqp_multi_create.argtypes = [C.POINTER(Dims), _ip, C.c_int32, C.POINTER(C.c_void_p)]
    lib.mpcqp_multi_destroy.argtypes = [C.c_void_p]
    lib.mpcqp_multi_ndev.argtypes = [C.c_void_p]
    lib.mpcqp_multi_handle.restype = C.c_void_p
    lib.mpcqp_multi_handle.argtypes = [C.c_void_p, C.c_int32]
    lib.mpcqp_multi_shard.argtypes = [C.c_void_p, C.c_int32, _ip, _ip]
    lib.mpcqp_multi_set_model.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    lib.mpcqp_multi_set_weights.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    lib.mpcqp_multi_set_bounds.argtypes = [C.c_void_p, C.POINTER(Bounds)]
    lib.mpcqp_multi_prepare.argtypes = [C.c_void_p]
    lib.mpcqp_multi_step.argtypes = [C.c_void_p] + [C.c_void_p] * 11
    lib.mpcqp_multi_gather_device.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 7
    lib.mpcqp_multi_scatter_device.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 3 + [C.c_int32] + [C.c_void_p] * 4
    _lib = lib
    return lib


def _chk(lib, rc):
    if rc != 0:
        msg = lib.mpcqp_strerror(rc).decode()
        if rc == -6:
            msg += ": " + lib.mpcqp_last_hip_error().decode()
        raise MpcqpError(f"mpcqp error {rc}: {msg}")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def colmajor(M):
    """(B, rows, cols) logical -> the ABI's column-major-per-problem buffer (= Julia (rows,cols,B))."""
    return _f64(np.asarray(M, float).transpose(0, 2, 1))


class Handle:
    """Thin binding of the C-ABI; array arguments are already in ABI layout."""

    def __init__(self, B, nxhat, nu, ny, nd, Hp, Hc, nb=None, neps=1, device=0, flags=0,
                 max_iter=0, gap_tol=0.0, res_tol=0.0, dual_reg=0.0, lib=None):
        self.lib = lib or load_library()
        self._nb = None
        d = Dims(batch=B, nxhat=nxhat, nu=nu, ny=ny, nd=nd, Hp=Hp, Hc=Hc, neps=neps, device=device,
                 flags=flags, max_iter=max_iter, gap_tol=gap_tol, res_tol=res_tol, dual_reg=dual_reg)
        if nb is not None:
            self._nb = (C.c_int32 * len(nb))(*[int(v) for v in nb])
            d.nb = C.cast(self._nb, _ip)
        self.h = C.c_void_p()
        _chk(self.lib, self.lib.mpcqp_create(C.byref(d), C.byref(self.h)))
        s = Sizes()
        _chk(self.lib, self.lib.mpcqp_get_sizes(self.h, C.byref(s)))
        self.B, self.nxhat, self.nu, self.ny, self.nd, self.Hp, self.Hc = B, nxhat, nu, ny, nd, Hp, Hc
        self.nZ, self.nDU, self.nU, self.nY, self.nD = s.nZ, s.nDU, s.nU, s.nY, s.nD
        self.flags = flags
        self._keep = []

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mpcqp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_model(self, Ahat, Bu, Cm, Bd=None, Dd=None, dop=None):
        args = [None if a is None else _f64(a) for a in (Ahat, Bu, Cm, Bd, Dd, dop)]
        _chk(self.lib, self.lib.mpcqp_set_model(self.h, *[_ptr(a) for a in args]))

    def set_weights(self, Mdiag, Ndiag, Ldiag, Cwt=None):
        args = [None if a is None else _f64(a) for a in (Mdiag, Ndiag, Ldiag, Cwt)]
        _chk(self.lib, self.lib.mpcqp_set_weights(self.h, *[_ptr(a) for a in args]))

    def set_output_weight_blocks(self, Mblk):
        """Mblk (B, Hp, ny, ny), symmetric blocks, or None (back to the diagonal weight)."""
        a = None if Mblk is None else _f64(np.asarray(Mblk, float).transpose(0, 1, 3, 2))
        _chk(self.lib, self.lib.mpcqp_set_output_weight_blocks(self.h, _ptr(a)))

    def set_dense_weights(self, M_Hp=None, N_Hc=None, L_Hp=None):
        """Dense symmetric weights (B, n, n) or None; they replace the diagonals of set_weights."""
        arrs = [None if a is None else colmajor(a) for a in (M_Hp, N_Hc, L_Hp)]
        _chk(self.lib, self.lib.mpcqp_set_dense_weights(self.h, *[_ptr(a) for a in arrs]))

    def set_flags(self, flags):
        _chk(self.lib, self.lib.mpcqp_set_flags(self.h, int(flags)))
        self.flags = int(flags)

    def set_transcription(self, transcription):
        """MPCQP_SINGLE_SHOOTING / MPCQP_MULTIPLE_SHOOTING (the stage-structured kernel, csrc/ms_bodies.h)."""
        _chk(self.lib, self.lib.mpcqp_set_transcription(self.h, int(transcription)))

    def transcription_supported(self):
        """0 when the handle's transcription can run; else the reason mask of include/mpcqp.h."""
        return self.lib.mpcqp_transcription_supported(self.h)

    def set_iteration_limit(self, max_iter):
        _chk(self.lib, self.lib.mpcqp_set_iteration_limit(self.h, int(max_iter)))

    def set_current_setpoint(self, ry_now):
        a = None if ry_now is None else _f64(ry_now)
        _chk(self.lib, self.lib.mpcqp_set_current_setpoint(self.h, _ptr(a)))

    def prepare(self):
        """Build/load the specialised step kernel of the current shape and constraint pattern
        (mpcqp_prepare); returns KERNEL_GENERIC / KERNEL_AOT / KERNEL_ONDEMAND.  Steps never compile."""
        k = self.lib.mpcqp_prepare(self.h)
        if k < 0:
            _chk(self.lib, k)
        if k == KERNEL_GENERIC:
            msg = self.lib.mpcqp_last_build_error().decode()
            if msg:
                warnings.warn(f"mpcqp: specialised kernel not available, using the runtime-dimension "
                              f"kernel ({msg})", RuntimeWarning)
        return k

    def lds_bytes(self):
        return self.lib.mpcqp_lds_bytes(self.h)

    def kernel_kind(self):
        return self.lib.mpcqp_kernel_kind(self.h)

    def row_groups(self):
        g = C.c_uint32()
        _chk(self.lib, self.lib.mpcqp_row_groups(self.h, C.byref(g)))
        return int(g.value)

    def set_custom_constraints(self, nw, Wy=None, Wu=None, Wd=None, Wr=None, w_op=None):
        args = [None if a is None else _f64(a) for a in (Wy, Wu, Wd, Wr, w_op)]
        _chk(self.lib, self.lib.mpcqp_set_custom_constraints(self.h, int(nw), *[_ptr(a) for a in args]))

    def set_custom_bounds(self, Wmin=None, Wmax=None, C_wmin=None, C_wmax=None):
        args = [None if a is None else _f64(a) for a in (Wmin, Wmax, C_wmin, C_wmax)]
        _chk(self.lib, self.lib.mpcqp_set_custom_bounds(self.h, *[_ptr(a) for a in args]))

    def set_bounds(self, **kw):
        b = Bounds()
        keep = []
        for k in BOUND_FIELDS:
            v = kw.get(k)
            if v is not None:
                a = _f64(v)
                keep.append(a)
                setattr(b, k, a.ctypes.data_as(_dp))
        _chk(self.lib, self.lib.mpcqp_set_bounds(self.h, C.byref(b)))

    def step(self, xhat0, lastu0, Ry, Z, Ru=None, d0=None, Dhat0=None, want_Yhat=False):
        """Host-pointer step.  Z (B,nZ) is updated in place.  Returns u0, status, iters[, Yhat0]."""
        B = self.B
        x, lu, ry = _f64(xhat0), _f64(lastu0), _f64(Ry)
        ru = None if Ru is None else _f64(Ru)
        dd0 = None if d0 is None else _f64(d0)
        dh = None if Dhat0 is None else _f64(Dhat0)
        nry = self.ny if (self.flags & FLAG_RY_CONSTANT) else self.nY
        if x.size != B * self.nxhat or lu.size != B * self.nu or ry.size != B * nry or Z.size != B * self.nZ:
            raise ValueError("DimensionMismatch in step arguments")
        assert Z.dtype == np.float64 and Z.flags.c_contiguous
        u0 = np.empty((B, self.nu))
        status = np.empty(B, np.int32)
        iters = np.empty(B, np.int32)
        yh = np.empty((B, self.nY)) if want_Yhat else None
        _chk(self.lib, self.lib.mpcqp_step(self.h, _ptr(x), _ptr(lu), _ptr(ry), _ptr(ru), _ptr(dd0),
                                           _ptr(dh), _ptr(Z), _ptr(u0), _ptr(status), _ptr(iters),
                                           _ptr(yh)))
        return (u0, status, iters, yh) if want_Yhat else (u0, status, iters)

    def step_device(self, xhat0, lastu0, Ry, Z, u0, status, iters=0, Ru=0, d0=0, Dhat0=0, Yhat0=0,
                    stream=0):
        """All arguments are integer device addresses (0 = NULL); asynchronous on `stream`."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        _chk(self.lib, self.lib.mpcqp_step_device(self.h, v(xhat0), v(lastu0), v(Ry), v(Ru), v(d0),
                                                  v(Dhat0), v(Z), v(u0), v(status), v(iters),
                                                  v(Yhat0), v(stream)))

    def loop_device(self, xhat0, y0m, lastu0, Ry, Z, u0, status, iters=0, Ru=0, d0=0, Dhat0=0, Yhat0=0, stream=0):
        """preparestate! + moveinput! + updatestate! of one period in one launch (mpcqp_loop_device);
        integer device addresses, xhat0 updated in place."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        _chk(self.lib, self.lib.mpcqp_loop_device(self.h, v(xhat0), v(y0m), v(lastu0), v(Ry), v(Ru), v(d0), v(Dhat0),
                                                  v(Z), v(u0), v(status), v(iters), v(Yhat0), v(stream)))

    def recondense_device(self, stream=0):
        _chk(self.lib, self.lib.mpcqp_recondense_device(self.h, C.c_void_p(int(stream)) if stream else None))

    def get(self, which):
        shape = {GET_HESSIAN: (self.B, self.nZ, self.nZ), GET_STEPRESP: (self.B, self.Hp, self.nu, self.ny),
                 GET_KMAT: (self.B, self.nxhat, self.nY), GET_BVEC: (self.B, self.nY),
                 GET_QTILDE: (self.B, self.nZ), GET_FVEC: (self.B, self.nY), GET_AUDIT: (self.B, 4),
                 GET_XHAT_MS: (self.B, self.Hp, self.nxhat), GET_MS_DEFECT: (self.B,)}[which]
        out = np.empty(shape)
        _chk(self.lib, self.lib.mpcqp_get(self.h, which, _ptr(out)))
        return out

    def audit(self):
        """What the convergence test of the last step saw: dict of (B,) arrays -- complementarity gap `mu`, relative
        dual / primal residuals `rd`, `rp`, and `polished` (the returned point passed the KKT check of the polish)."""
        a = self.get(GET_AUDIT)
        return {"mu": a[:, 0], "rd": a[:, 1], "rp": a[:, 2], "polished": a[:, 3] > 0.5}

    def last_step_ms(self):
        return self.lib.mpcqp_last_step_ms(self.h)

    # -- SteadyKalmanFilter steps -----------------------------------------------
    def kf_set(self, Khat, i_ym):
        K = _f64(Khat)
        iy = np.ascontiguousarray(i_ym, dtype=np.int32)
        self.nym = int(iy.size)
        _chk(self.lib, self.lib.mpcqp_kf_set(self.h, _ptr(K), _ptr(iy), self.nym))

    def kf_correct(self, xhat0, y0m, d0=None):
        """x̂0 += K̂ (y0m - Ĉm x̂0 - D̂dm d0), in place on the (B,nx̂) host array."""
        assert xhat0.dtype == np.float64 and xhat0.flags.c_contiguous
        y, dd = _f64(y0m), (None if d0 is None else _f64(d0))
        _chk(self.lib, self.lib.mpcqp_kf_correct(self.h, _ptr(xhat0), _ptr(y), _ptr(dd)))

    def kf_predict(self, xhat0, u0, d0=None):
        """x̂0 <- Â x̂0 + B̂u u0 + B̂d d0 + (f̂op - x̂op), in place on the (B,nx̂) host array."""
        assert xhat0.dtype == np.float64 and xhat0.flags.c_contiguous
        u, dd = _f64(u0), (None if d0 is None else _f64(d0))
        _chk(self.lib, self.lib.mpcqp_kf_predict(self.h, _ptr(xhat0), _ptr(u), _ptr(dd)))

    def kf_correct_device(self, xhat0, y0m, d0=0, stream=0):
        v = lambda p: C.c_void_p(int(p)) if p else None
        _chk(self.lib, self.lib.mpcqp_kf_correct_device(self.h, v(xhat0), v(y0m), v(d0), v(stream)))

    def kf_predict_device(self, xhat0, u0, d0=0, stream=0):
        v = lambda p: C.c_void_p(int(p)) if p else None
        _chk(self.lib, self.lib.mpcqp_kf_predict_device(self.h, v(xhat0), v(u0), v(d0), v(stream)))

    def last_condense_ms(self):
        return self.lib.mpcqp_last_condense_ms(self.h)

    def last_predmat_ms(self):
        return self.lib.mpcqp_last_predmat_ms(self.h)


class MultiHandle:
    """Binding of the mpcqp_multi_* entry points: one batch over several GPUs of a node (an ordinal
    may repeat), whole-batch HOST arrays in, whole-batch HOST arrays out."""

    def __init__(self, B, nxhat, nu, ny, nd, Hp, Hc, devices, nb=None, neps=1, flags=0, max_iter=0,
                 gap_tol=0.0, res_tol=0.0, dual_reg=0.0, lib=None):
        self.lib = lib or load_library()
        d = Dims(batch=B, nxhat=nxhat, nu=nu, ny=ny, nd=nd, Hp=Hp, Hc=Hc, neps=neps, device=0, flags=flags,
                 max_iter=max_iter, gap_tol=gap_tol, res_tol=res_tol, dual_reg=dual_reg)
        self._nb = None
        if nb is not None:
            self._nb = (C.c_int32 * len(nb))(*[int(v) for v in nb])
            d.nb = C.cast(self._nb, _ip)
        dev = (C.c_int32 * len(devices))(*[int(v) for v in devices])
        self.h = C.c_void_p()
        _chk(self.lib, self.lib.mpcqp_multi_create(C.byref(d), dev, len(devices), C.byref(self.h)))
        self.B, self.nxhat, self.nu, self.ny, self.nd, self.Hp, self.Hc = B, nxhat, nu, ny, nd, Hp, Hc
        self.nDU, self.nZ, self.nU, self.nY, self.nD = nu * Hc, nu * Hc + neps, nu * Hp, ny * Hp, nd * Hp
        self.flags = flags
        self._devices = [int(v) for v in devices]
        self.ndev = self.lib.mpcqp_multi_ndev(self.h)

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mpcqp_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard(self, g):
        o, n = C.c_int32(), C.c_int32()
        _chk(self.lib, self.lib.mpcqp_multi_shard(self.h, g, C.byref(o), C.byref(n)))
        return int(o.value), int(n.value)

    def set_model(self, Ahat, Bu, Cm, Bd=None, Dd=None, dop=None):
        args = [None if a is None else _f64(a) for a in (Ahat, Bu, Cm, Bd, Dd, dop)]
        _chk(self.lib, self.lib.mpcqp_multi_set_model(self.h, *[_ptr(a) for a in args]))

    def set_weights(self, Mdiag, Ndiag, Ldiag, Cwt=None):
        args = [None if a is None else _f64(a) for a in (Mdiag, Ndiag, Ldiag, Cwt)]
        _chk(self.lib, self.lib.mpcqp_multi_set_weights(self.h, *[_ptr(a) for a in args]))

    def set_bounds(self, **kw):
        b = Bounds()
        keep = []
        for k in BOUND_FIELDS:
            v = kw.get(k)
            if v is not None:
                a = _f64(v)
                keep.append(a)
                setattr(b, k, a.ctypes.data_as(_dp))
        _chk(self.lib, self.lib.mpcqp_multi_set_bounds(self.h, C.byref(b)))

    def prepare(self):
        k = self.lib.mpcqp_multi_prepare(self.h)
        if k < 0:
            _chk(self.lib, k)
        return k

    def step(self, xhat0, lastu0, Ry, Z, Ru=None, d0=None, Dhat0=None, want_Yhat=False):
        B = self.B
        x, lu, ry = _f64(xhat0), _f64(lastu0), _f64(Ry)
        ru = None if Ru is None else _f64(Ru)
        dd0 = None if d0 is None else _f64(d0)
        dh = None if Dhat0 is None else _f64(Dhat0)
        assert Z.dtype == np.float64 and Z.flags.c_contiguous and Z.size == B * self.nZ
        u0 = np.empty((B, self.nu)); status = np.empty(B, np.int32); iters = np.empty(B, np.int32)
        yh = np.empty((B, self.nY)) if want_Yhat else None
        _chk(self.lib, self.lib.mpcqp_multi_step(self.h, _ptr(x), _ptr(lu), _ptr(ry), _ptr(ru), _ptr(dd0), _ptr(dh),
                                                 _ptr(Z), _ptr(u0), _ptr(status), _ptr(iters), _ptr(yh)))
        return (u0, status, iters, yh) if want_Yhat else (u0, status, iters)


    # ---- device-resident path: per-device buffers, nothing crosses PCIe --------------------------------------
    def device_of(self, g):
        """CUDA/HIP ordinal the shard g lives on (as given to the constructor)."""
        return self._devices[g]

    def step_device_shard(self, g, xhat0, lastu0, Ry, Z, u0, status, iters=0, stream=0):
        """mpcqp_step_device on the handle of shard g: integer device addresses of that shard's buffers."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        hg = C.c_void_p(self.lib.mpcqp_multi_handle(self.h, g))
        _chk(self.lib, self.lib.mpcqp_step_device(hg, v(xhat0), v(lastu0), v(Ry), None, None, None, v(Z), v(u0), v(status),
                                                  v(iters), None, v(stream)))

    def scatter_device(self, root, xhat0_root, lastu0_root, Ry_root, ry_rows, xhat0_shards, lastu0_shards, Ry_shards, stream=0):
        """mpcqp_multi_scatter_device: whole-batch inputs on device `root` -> the shards' buffers (integer addresses)."""
        arr = lambda ps: (C.c_void_p * len(ps))(*[C.c_void_p(int(p)) for p in ps])
        v = lambda p: C.c_void_p(int(p)) if p else None
        _chk(self.lib, self.lib.mpcqp_multi_scatter_device(self.h, root, v(xhat0_root), v(lastu0_root), v(Ry_root), int(ry_rows),
                                                          arr(xhat0_shards), arr(lastu0_shards), arr(Ry_shards), v(stream)))

    def gather_device(self, root, Z_shards, u0_shards, status_shards, Z_root, u0_root, status_root, stream=0):
        """mpcqp_multi_gather_device: the shards' Z̃, u0, status -> whole-batch buffers on device `root`."""
        arr = lambda ps: (C.c_void_p * len(ps))(*[C.c_void_p(int(p)) for p in ps])
        v = lambda p: C.c_void_p(int(p)) if p else None
        _chk(self.lib, self.lib.mpcqp_multi_gather_device(self.h, root, arr(Z_shards), arr(u0_shards), arr(status_shards),
                                                         v(Z_root), v(u0_root), v(status_root), v(stream)))


def steady_kalman_gain(Ahat, Chat, Qhat, Rhat, i_ym=None):
    """K̂ of `SteadyKalmanFilter` for a batch: filter-form gain P Ĉm'(Ĉm P Ĉm' + R̂)^-1 with P the
    predictor DARE solution (`init_skf`, src/estimator/kalman.jl:204-236; construction-time,
    host-side like the reference, which calls ControlSystemsBase.kalman)."""
    from scipy.linalg import solve_discrete_are
    Ahat, Chat = np.asarray(Ahat, float), np.asarray(Chat, float)
    B, ny, nxh = Chat.shape
    i_ym = np.arange(ny) if i_ym is None else np.asarray(i_ym, int)
    Q = np.broadcast_to(np.asarray(Qhat, float), (B, nxh, nxh))
    R = np.broadcast_to(np.asarray(Rhat, float), (B, len(i_ym), len(i_ym)))
    K = np.empty((B, nxh, len(i_ym)))
    for b in range(B):
        Cm = Chat[b][i_ym]
        P = solve_discrete_are(Ahat[b].T, Cm.T, Q[b], R[b])
        K[b] = P @ Cm.T @ np.linalg.inv(Cm @ P @ Cm.T + R[b])
    return K


def move_blocking(Hp, Hc):
    """`move_blocking` (src/controller/construct.jl:629-660): integer or vector Hc -> nb."""
    if np.isscalar(Hc):
        Hc = int(Hc)
        nb = [1] * Hc
        if Hc > 0:
            nb[-1] = Hp - Hc + 1
        return nb
    nb = [int(v) for v in Hc]
    if not all(v > 0 for v in nb):
        raise ValueError("Move blocking vector must be strictly positive integers.")
    if sum(nb) < Hp:
        nb = nb + [Hp - sum(nb)]
    elif sum(nb) > Hp:
        keep = int(np.argmax(np.cumsum(nb) >= Hp)) + 1
        nb = nb[:keep]
        if sum(nb) > Hp:
            nb[-1] = Hp - sum(nb[:-1])
    return nb


class BatchLinMPC:
    """B independent `LinMPC` controllers on augmented models (Â, B̂u, Ĉ, B̂d, D̂d), one GPU.

    Array arguments carry a leading batch axis: Ahat (B,nx̂,nx̂), Bhu (B,nx̂,nu), Chat (B,ny,nx̂),
    Bhd (B,nx̂,nd), Dhd (B,ny,nd).  Weights are per channel (ny / nu values, shared or (B,·)),
    repeated over the horizons like `Diagonal(repeat(Mwt, Hp))`; Cwt scalar or (B,).
    """

    def __init__(self, Ahat, Bhu, Chat, Bhd=None, Dhd=None, *, Hp, Hc=2, Mwt=None, Nwt=None,
                 Lwt=None, M_Hp=None, N_Hc=None, L_Hp=None, Cwt=1e5, Wy=None, Wu=None, Wd=None, Wr=None, uop=None, yop=None, dop=None, xhop=None, fhop=None, device=0,
                 cold_start=False, keep_qp=False, warm_dual=False, max_iter=0, gap_tol=0.0, res_tol=0.0, dual_reg=0.0,
                 transcription="SingleShooting", lib=None):
        # `transcription` (LinMPC keyword, linmpc.jl:288-316).  MultipleShooting (Z = [ΔU; X̂0] with the model as
        # equality constraints, transcription.jl:217-240, 373-414, 913-928) runs on the stage-structured kernel of
        # csrc/ms_bodies.h (Riccati recursion inside the interior-point iteration; H̃ and E are never formed): what the
        # reference recommends for unstable plants / long horizons, construct.jl:855-866.  Handles that kernel does not
        # take (block / dense weight matrices, custom linear constraints, stage data beyond the LDS) keep the condensed
        # kernels -- same optimal ΔU -- with a warning; getinfo returns Z̃ in the transcription's layout either way.
        if transcription not in ("SingleShooting", "MultipleShooting"):
            raise NotImplementedError(f"transcription {transcription!r}: only SingleShooting / MultipleShooting (LinModel)")
        self.transcription = transcription
        Ahat, Bhu, Chat = (np.asarray(a, float) for a in (Ahat, Bhu, Chat))
        if Ahat.ndim != 3 or Bhu.ndim != 3 or Chat.ndim != 3:
            raise ValueError("model matrices need a leading batch axis")
        B, nxh, nu = Bhu.shape
        ny = Chat.shape[1]
        nd = 0 if Bhd is None else np.asarray(Bhd).shape[2]
        if Ahat.shape != (B, nxh, nxh) or Chat.shape != (B, ny, nxh):
            raise ValueError("DimensionMismatch between Ahat, Bhu, Chat")
        Hp = int(Hp)
        if Hp < 1:
            raise ValueError("Prediction horizon Hp should be ≥ 1")
        nb = move_blocking(Hp, Hc)
        Hc = len(nb)
        if Hc < 1:
            raise ValueError("Control horizon Hc should be ≥ 1")
        if Hc > Hp:
            raise ValueError("Control horizon Hc should be ≤ prediction horizon Hp")
        cw = np.broadcast_to(np.asarray(Cwt, float), (B,)).copy()
        if np.any(cw < 0):
            raise ValueError("Cwt weight should be ≥ 0")
        if np.isinf(cw).any() and not np.isinf(cw).all():
            raise ValueError("Cwt must be finite for all controllers of a batch, or Inf for all")
        self.neps = 0 if np.isinf(cw).all() else 1
        self.B, self.nxh, self.nu, self.ny, self.nd, self.Hp, self.Hc, self.nb = B, nxh, nu, ny, nd, Hp, Hc, nb
        flags = (FLAG_RY_CONSTANT * 0 | (FLAG_COLD_START if cold_start else 0) | (FLAG_KEEP_QP if keep_qp else 0)
                 | (FLAG_WARM_DUAL if warm_dual else 0))
        self.hd = Handle(B, nxh, nu, ny, nd, Hp, Hc, nb=nb, neps=self.neps, device=device, flags=flags,
                         max_iter=max_iter, gap_tol=gap_tol, res_tol=res_tol, dual_reg=dual_reg, lib=lib)
        self.nZ, self.nDU, self.nU, self.nY = self.hd.nZ, self.hd.nDU, self.hd.nU, self.hd.nY
        vec = lambda v, n: np.zeros((B, n)) if v is None else np.broadcast_to(np.asarray(v, float), (B, n)).copy()
        self.uop, self.yop, self.dop = vec(uop, nu), vec(yop, ny), vec(dop, nd)
        self.xhop, self.fhop = vec(xhop, nxh), vec(fhop, nxh)
        self.Uop, self.Yop, self.Dop = np.tile(self.uop, Hp), np.tile(self.yop, Hp), np.tile(self.dop, Hp)
        self.Cwt = cw
        self.setmodel(Ahat, Bhu, Chat, Bhd, Dhd)
        self.setweights(Mwt, Nwt, Lwt, M_Hp=M_Hp, N_Hc=N_Hc, L_Hp=L_Hp)
        # custom linear constraints (validate_custom_lincon, src/controller/construct.jl:666-694)
        given = [np.asarray(W, float) for W in (Wy, Wu, Wd, Wr) if W is not None]
        self.nw = 0
        if given:
            first = given[0]
            self.nw = (first.shape[0] if first.ndim == 2 else first.shape[1]) if first.ndim >= 2 else 1
        nw = self.nw
        def wmat(Wm, n, name):
            if Wm is None:
                return np.zeros((B, nw, n))
            a = np.asarray(Wm, float)
            if a.ndim == 1:
                a = a.reshape(1, -1)
            if a.ndim == 2:
                a = np.broadcast_to(a, (B,) + a.shape)
            if a.shape[2] != n:
                raise ValueError(f"{name} must have {n} columns")           # DimensionMismatch, :686-689
            if a.shape[1] != nw:
                raise ValueError("Wy, Wu, Wd and Wr must have the same number of rows")
            return a.copy()
        self.Wy, self.Wu, self.Wd, self.Wr = wmat(Wy, ny, "Wy"), wmat(Wu, nu, "Wu"), wmat(Wd, nd, "Wd"), wmat(Wr, ny, "Wr")
        self._wb = dict(Wmin=None, Wmax=None, C_wmin=None, C_wmax=None)
        if nw > 0:
            w_op = (np.einsum("bij,bj->bi", self.Wy, self.yop) + np.einsum("bij,bj->bi", self.Wu, self.uop)
                    + np.einsum("bij,bj->bi", self.Wr, self.yop))
            if nd > 0:
                w_op = w_op + np.einsum("bij,bj->bi", self.Wd, self.dop)
            self.hd.set_custom_constraints(nw, colmajor(self.Wy), colmajor(self.Wu),
                                           colmajor(self.Wd) if nd > 0 else None, colmajor(self.Wr), w_op)
        # default constraints: none (src/controller/construct.jl:887-913)
        self._b = {k: None for k in BOUND_FIELDS}
        self._prepared = False                     # specialised kernel built/loaded for the current pattern
        self.Z = np.zeros((B, self.nZ))            # mpc.Z̃ (previous optimum)
        self.lastu0 = np.zeros((B, nu))            # mpc.lastu0
        self.solved_once = False
        self.status = np.zeros(B, np.int32)
        self.iters = np.zeros(B, np.int32)

    # -- model / weights ---------------------------------------------------------------------
    def setmodel(self, Ahat, Bhu, Chat, Bhd=None, Dhd=None):
        """`setmodel!` re-condensation path (src/controller/execute.jl:684-790): K1 (+K2)."""
        dopv = self.fhop - self.xhop
        self._Ahat, self._Bhu = np.asarray(Ahat, float).copy(), np.asarray(Bhu, float).copy()
        self._Chat = np.asarray(Chat, float).copy()
        self._Bhd = None if self.nd == 0 else np.asarray(Bhd, float).copy()
        self._Dhd = None if self.nd == 0 else np.asarray(Dhd, float).copy()
        self.hd.set_model(colmajor(Ahat), colmajor(Bhu), colmajor(Chat),
                          None if self.nd == 0 else colmajor(Bhd),
                          None if self.nd == 0 else colmajor(Dhd),
                          dopv if np.any(dopv != 0) else None)

    def setweights(self, Mwt=None, Nwt=None, Lwt=None, M_Hp=None, N_Hc=None, L_Hp=None):
        """Defaults of src/general.jl:3-6 (Mwt=1, Nwt=0.1, Lwt=0).  `M_Hp` (nY,nY), `N_Hc` (nΔU,nΔU), `L_Hp`
        (nU,nU) -- each also (B,n,n) -- are the reference's full weight matrices (src/controller/linmpc.jl:205-214,
        construct.jl:45-93); they override the per-channel vectors.  A block-diagonal M_Hp = blkdiag(M_1..M_Hp)
        (e.g. a terminal cost) keeps the specialised step kernels; an M_Hp or L_Hp that couples different prediction
        steps runs on the runtime-dimension kernel; a dense N_Hc only changes H̃."""
        B, Hp, Hc, ny = self.B, self.Hp, self.Hc, self.ny
        w = lambda v, n, dflt: (np.full((B, n), dflt) if v is None
                                else np.broadcast_to(np.asarray(v, float), (B, n)).copy())
        M, N, L = w(Mwt, self.ny, 1.0), w(Nwt, self.nu, 0.1), w(Lwt, self.nu, 0.0)
        for name, a in (("Mwt", M), ("Nwt", N), ("Lwt", L)):
            if np.any(a < 0):
                raise ValueError(f"{name} values should be nonnegative")
        self.Mwt, self.Nwt, self.Lwt = M, N, L
        self.hd.set_weights(np.tile(M, Hp), np.tile(N, Hc), np.tile(L, Hp),
                            self.Cwt if self.neps else None)

        def full(Mx, n, name):
            if Mx is None:
                return None
            Mf = np.asarray(Mx, float)
            if Mf.shape == (n, n):
                Mf = np.broadcast_to(Mf, (B, n, n))
            if Mf.shape != (B, n, n):
                raise ValueError(f"{name} size should be ({n}, {n})")
            if not np.allclose(Mf, Mf.transpose(0, 2, 1), rtol=0, atol=1e-12 * max(1.0, np.abs(Mf).max())):
                raise ValueError(f"{name} should be Hermitian")
            return np.ascontiguousarray(Mf)

        self.Mblk, self.Mfull = None, None
        Mf, self.Ndense, self.Ldense = full(M_Hp, self.nY, "M_Hp"), full(N_Hc, self.nDU, "N_Hc"), full(L_Hp, self.nU, "L_Hp")
        if Mf is not None:
            blk = np.stack([Mf[:, t * ny:(t + 1) * ny, t * ny:(t + 1) * ny] for t in range(Hp)], axis=1)
            off = Mf.copy()
            for t in range(Hp):
                off[:, t * ny:(t + 1) * ny, t * ny:(t + 1) * ny] = 0.0
            if np.any(off != 0.0):
                self.Mfull = Mf                      # couples different prediction steps: dense path
            else:
                self.Mblk = blk
        if self.Mblk is not None:
            self.hd.set_output_weight_blocks(self.Mblk)
        elif getattr(self, "_had_blocks", False):
            self.hd.set_output_weight_blocks(None)
        self._had_blocks = self.Mblk is not None
        dense = (self.Mfull, self.Ndense, self.Ldense)
        if any(a is not None for a in dense) or getattr(self, "_had_dense", False):
            self.hd.set_dense_weights(*dense)
        self._had_dense = any(a is not None for a in dense)
        self._prepared = False

    # -- constraints --------------------------------------------------------------------------
    def setconstraint(self, *, umin=None, umax=None, Δumin=None, Δumax=None, ymin=None, ymax=None,
                      x̂min=None, x̂max=None, Umin=None, Umax=None, ΔUmin=None, ΔUmax=None,
                      Ymin=None, Ymax=None, c_umin=None, c_umax=None, c_Δumin=None, c_Δumax=None,
                      c_ymin=None, c_ymax=None, c_x̂min=None, c_x̂max=None,
                      Deltaumin=None, Deltaumax=None, xhatmin=None, xhatmax=None,
                      DeltaUmin=None, DeltaUmax=None, c_Deltaumin=None, c_Deltaumax=None,
                      c_xhatmin=None, c_xhatmax=None, wmin=None, wmax=None, Wmin=None, Wmax=None,
                      c_wmin=None, c_wmax=None, C_wmin=None, C_wmax=None,
                      C_umin=None, C_umax=None, C_Δumin=None, C_Δumax=None, C_ymin=None, C_ymax=None,
                      C_Deltaumin=None, C_Deltaumax=None):
        """`setconstraint!` (src/controller/construct.jl:324-559), with the ASCII
        aliases.  Bounds are engineering values (operating points are subtracted here, :356-435);
        per-channel vectors (n,) or (B,n) are repeated over the horizon, capitalised keywords take
        the whole horizon."""
        Δumin = Deltaumin if Δumin is None else Δumin
        Δumax = Deltaumax if Δumax is None else Δumax
        x̂min = xhatmin if x̂min is None else x̂min
        x̂max = xhatmax if x̂max is None else x̂max
        ΔUmin = DeltaUmin if ΔUmin is None else ΔUmin
        ΔUmax = DeltaUmax if ΔUmax is None else ΔUmax
        c_Δumin = c_Deltaumin if c_Δumin is None else c_Δumin
        c_Δumax = c_Deltaumax if c_Δumax is None else c_Δumax
        c_x̂min = c_xhatmin if c_x̂min is None else c_x̂min
        c_x̂max = c_xhatmax if c_x̂max is None else c_x̂max
        B, Hp, Hc, nu, ny, nxh = self.B, self.Hp, self.Hc, self.nu, self.ny, self.nxh

        def rep(v, n, reps, name):
            a = np.asarray(v, float)
            if a.shape not in ((n,), (B, n)):
                raise ValueError(f"{name} size must be ({n},)")
            return np.tile(np.broadcast_to(a, (B, n)), reps)

        def full(v, n, name):
            a = np.asarray(v, float)
            if a.shape not in ((n,), (B, n)):
                raise ValueError(f"{name} size must be ({n},)")
            return np.broadcast_to(a, (B, n)).copy()

        new = dict(self._b)
        if Umin is not None:
            new["U0min"] = full(Umin, nu * Hp, "Umin") - self.Uop
        elif umin is not None:
            new["U0min"] = rep(umin, nu, Hp, "umin") - self.Uop
        if Umax is not None:
            new["U0max"] = full(Umax, nu * Hp, "Umax") - self.Uop
        elif umax is not None:
            new["U0max"] = rep(umax, nu, Hp, "umax") - self.Uop
        if ΔUmin is not None:
            new["DUmin"] = full(ΔUmin, nu * Hc, "ΔUmin")
        elif Δumin is not None:
            new["DUmin"] = rep(Δumin, nu, Hc, "Δumin")
        if ΔUmax is not None:
            new["DUmax"] = full(ΔUmax, nu * Hc, "ΔUmax")
        elif Δumax is not None:
            new["DUmax"] = rep(Δumax, nu, Hc, "Δumax")
        if Ymin is not None:
            new["Y0min"] = full(Ymin, ny * Hp, "Ymin") - self.Yop
        elif ymin is not None:
            new["Y0min"] = rep(ymin, ny, Hp, "ymin") - self.Yop
        if Ymax is not None:
            new["Y0max"] = full(Ymax, ny * Hp, "Ymax") - self.Yop
        elif ymax is not None:
            new["Y0max"] = rep(ymax, ny, Hp, "ymax") - self.Yop
        if x̂min is not None:
            new["x0min"] = full(x̂min, nxh, "x̂min") - self.xhop
        if x̂max is not None:
            new["x0max"] = full(x̂max, nxh, "x̂max") - self.xhop
        # softness: per channel (repeated over the horizon) or horizon-long `C_umin` ... `C_ymax` (construct.jl:446-509).  A
        # C_umin / C_umax that varies inside a move-blocking interval sends the handle to the stage-structured kernel
        # (mpcqp_set_bounds), which keeps one input row per step.
        C_Δumin = C_Deltaumin if C_Δumin is None else C_Δumin
        C_Δumax = C_Deltaumax if C_Δumax is None else C_Δumax
        ecr = dict(C_umin=(c_umin, nu, Hp, C_umin), C_umax=(c_umax, nu, Hp, C_umax), C_dumin=(c_Δumin, nu, Hc, C_Δumin),
                   C_dumax=(c_Δumax, nu, Hc, C_Δumax), C_ymin=(c_ymin, ny, Hp, C_ymin), C_ymax=(c_ymax, ny, Hp, C_ymax),
                   c_x0min=(c_x̂min, nxh, 1, None), c_x0max=(c_x̂max, nxh, 1, None))
        if any(v[0] is not None or v[3] is not None for v in ecr.values()):
            if self.neps != 1:
                raise ValueError("Slack variable weight Cwt must be finite to set softness parameters")
            if self.solved_once:
                raise RuntimeError("Cannot set softness parameters after calling moveinput!")
        for k, (v, n, reps, whole) in ecr.items():
            if whole is not None or v is not None:
                a = full(whole, n * reps, k) if whole is not None else rep(v, n, reps, k)
                if np.any(a < 0):
                    raise ValueError(f"{k} weights should be non-negative")
                new[k] = a
                self._prepared = False         # (the kernel that takes the handle may change, see above)
        if self.solved_once:
            # src/controller/construct.jl:541-551: the ±Inf pattern is frozen after the first solve
            for k in BOUND_FIELDS[:8]:
                old_inf = np.isinf(self._b[k]) if self._b[k] is not None else True
                new_inf = np.isinf(new[k]) if new[k] is not None else True
                if np.any(old_inf != new_inf):
                    raise RuntimeError("Cannot modify ±Inf constraints after calling moveinput!")
        # custom linear constraints: wmin/wmax (nw,) repeated over the Hp+1 steps, or Wmin/Wmax
        # (nw (Hp+1),) (construct.jl:411-426, 487-494)
        neww = dict(self._wb)
        nW = self.nw * (Hp + 1)
        for key, per, whole, nm in (("Wmin", wmin, Wmin, "wmin"), ("Wmax", wmax, Wmax, "wmax"),
                                    ("C_wmin", c_wmin, C_wmin, "c_wmin"), ("C_wmax", c_wmax, C_wmax, "c_wmax")):
            if whole is not None:
                neww[key] = full(whole, nW, nm.capitalize())
            elif per is not None:
                neww[key] = rep(per, self.nw, Hp + 1, nm)
        for key in ("C_wmin", "C_wmax"):
            if neww[key] is not self._wb[key]:
                if self.neps != 1:
                    raise ValueError("Slack variable weight Cwt must be finite to set softness parameters")
                if self.solved_once:
                    raise RuntimeError("Cannot set softness parameters after calling moveinput!")
                if np.any(neww[key] < 0):
                    raise ValueError(f"{key} weights should be non-negative")
        if self.solved_once:
            for key in ("Wmin", "Wmax"):
                old_inf = np.isinf(self._wb[key]) if self._wb[key] is not None else True
                new_inf = np.isinf(neww[key]) if neww[key] is not None else True
                if np.any(old_inf != new_inf):
                    raise RuntimeError("Cannot modify ±Inf constraints after calling moveinput!")
        self._b = new
        self.hd.set_bounds(**{k: v for k, v in new.items() if v is not None})
        self._prepared = False
        if self.nw > 0 and any(neww[k] is not self._wb[k] for k in neww):
            fin = lambda a: None if a is None or np.all(np.isinf(a)) else a
            self.hd.set_custom_bounds(fin(neww["Wmin"]), fin(neww["Wmax"]), neww["C_wmin"], neww["C_wmax"])
        self._wb = neww
        return self

    # -- estimator steps on both sides of moveinput! (SteadyKalmanFilter) ---------------------
    def setestimator(self, Khat, i_ym=None, xhat0=None):
        """Attach a SteadyKalmanFilter: Khat (B,nx̂,nym) steady-state gain (see
        `steady_kalman_gain`), i_ym measured-output indices (default all).  The estimate x̂0
        (deviation, (B,nx̂)) is then carried by this object like `mpc.estim.x̂0`."""
        self.i_ym = np.arange(self.ny) if i_ym is None else np.asarray(i_ym, int)
        Khat = np.asarray(Khat, float)
        if Khat.shape != (self.B, self.nxh, len(self.i_ym)):
            raise ValueError("Khat size must be (B, nx̂, nym)")
        self.hd.kf_set(colmajor(Khat), self.i_ym)
        self.xhat0 = np.zeros((self.B, self.nxh)) if xhat0 is None else _f64(np.broadcast_to(xhat0, (self.B, self.nxh))).copy()
        return self

    def preparestate(self, ym, d=None):
        """`preparestate!` (src/estimator/execute.jl:334-345 -> correct_estimate_obsv!,
        kalman.jl:284-295): x̂0 += K̂ (y0m - Ĉm x̂0 - D̂dm d0).  Returns x̂ = x̂0 + x̂op."""
        y0m = self._bc(ym, len(self.i_ym), "ym") - self.yop[:, self.i_ym]
        d0 = None if self.nd == 0 else self._bc(d, self.nd, "d") - self.dop
        self.hd.kf_correct(self.xhat0, y0m, d0)
        return self.xhat0 + self.xhop

    def updatestate(self, u, ym=None, d=None):
        """`updatestate!` (src/estimator/execute.jl:374-386 -> predict_estimate_obsv!,
        kalman.jl:298-309): x̂0 <- Â x̂0 + B̂u u0 + B̂d d0 + f̂op - x̂op."""
        u0 = self._bc(u, self.nu, "u") - self.uop
        d0 = None if self.nd == 0 else self._bc(d, self.nd, "d") - self.dop
        self.hd.kf_predict(self.xhat0, u0, d0)
        return self.xhat0 + self.xhop

    # -- per-step ---------------------------------------------------------------------------
    def initstate(self, u):
        """controller part of `initstate!` (src/controller/execute.jl:9-13)."""
        self.Z[:] = 0.0
        self.lastu0 = np.broadcast_to(np.asarray(u, float), (self.B, self.nu)) - self.uop

    def moveinput(self, xhat0, ry=None, d=None, *, lastu=None, Dhat=None, Rhaty=None, Rhatu=None,
                  want_info=False):
        """`moveinput!` for the whole batch (src/controller/execute.jl:59-80).

        xhat0 (B,nx̂) is `estim.x̂0` (deviation); ry (B,ny)/(ny,), d (B,nd), `lastu`, `Rhaty`
        (B,ny*Hp), `Rhatu` (B,nu*Hp), `Dhat` (B,nd*Hp) are engineering values like the reference.
        Returns u (B,nu)."""
        B, Hp = self.B, self.Hp
        if xhat0 is None:
            xhat0 = self.xhat0                      # the attached estimator's state (setestimator)
        xhat0 = np.asarray(xhat0, float)
        if xhat0.shape != (B, self.nxh):
            raise ValueError(f"xhat0 size must be ({B},{self.nxh})")
        bc = lambda v, n, name: self._bc(v, n, name)
        ry = self.yop if ry is None else bc(ry, self.ny, "ry")
        held = Rhaty is None                      # R̂y = repeat(ry, Hp): send ry once (MPCQP_FLAG_RY_CONSTANT)
        Rhaty = None if held else bc(Rhaty, self.nY, "R̂y")
        want = (self.hd.flags | FLAG_RY_CONSTANT) if held else (self.hd.flags & ~FLAG_RY_CONSTANT)
        if want != self.hd.flags:
            self.hd.set_flags(want)
        Rhatu = None if Rhatu is None else bc(Rhatu, self.nU, "R̂u")
        lastu0 = self.lastu0 if lastu is None else bc(lastu, self.nu, "lastu") - self.uop
        d0 = Dh0 = None
        if self.nd > 0:
            d = bc(d, self.nd, "d")
            Dhat = np.tile(d, Hp) if Dhat is None else bc(Dhat, self.nd * Hp, "D̂")
            d0, Dh0 = d - self.dop, Dhat - self.Dop
        elif d is not None and np.size(d) != 0:
            raise ValueError("d size must be (0,)")
        if self.nw > 0 and not held:               # r̂e(k) = ry(k) for the Wr term (execute.jl:351)
            self.hd.set_current_setpoint(ry - self.yop)
        if not self._prepared:                     # like JuMP's model build: before the loop, not in mpcqp_step
            self._ms_kernel = False
            if self.transcription == "MultipleShooting":
                self.hd.set_transcription(MULTIPLE_SHOOTING)
                why = self.hd.transcription_supported()
                if why:
                    self.hd.set_transcription(SINGLE_SHOOTING)
                    warnings.warn("mpcqp: MultipleShooting kernel not available for this controller (reason mask "
                                  f"{why}: 1 block/dense weights, 2 custom constraints, 4 LDS, 8 flags): the SingleShooting "
                                  "kernels solve the same problem", RuntimeWarning)
                else:
                    self._ms_kernel = True
            self.kernel = KERNEL_MS if self._ms_kernel else self.hd.prepare()
            if self.kernel == KERNEL_MS and not self._ms_kernel and self.nZ <= 256 and self.hd.lds_bytes() > 160 * 1024:
                # (the reroute used to be silent -- the condensed problem does not fit the LDS of a CU in the
                #  carve-up of the runtime-dimension kernel; same QP, solved in stage form, an order of magnitude slower)
                warnings.warn(f"mpcqp: the condensed problem (nZ̃ = {self.nZ}, {self.hd.lds_bytes()} B of LDS) does not fit one "
                              "compute unit: the SingleShooting controller's steps run on the stage-structured "
                              "(MultipleShooting) kernel", RuntimeWarning)
            self._ms_kernel = self.kernel == KERNEL_MS       # (also: nZ̃ > 256, which only the stage-structured kernel takes)
            self._prepared = True
        out = self.hd.step(xhat0, lastu0, (ry - self.yop) if held else (Rhaty - self.Yop), self.Z,
                           Ru=None if Rhatu is None else Rhatu - self.Uop, d0=d0, Dhat0=Dh0,
                           want_Yhat=want_info)
        u0, self.status, self.iters = out[0], out[1], out[2]
        self._Yhat0 = out[3] if want_info else None
        self._lastu0_prev = lastu0
        self._winfo = (xhat0, np.tile(ry, Hp) if (held and self.nw > 0) else Rhaty, d0, Dh0)
        self._ry_now = ry.copy()
        self._ginfo = (xhat0.copy(), np.tile(ry, Hp) if held else Rhaty, self.Uop.copy() if Rhatu is None else Rhatu)
        self.solved_once = True
        nerr = int(np.sum(self.status == STATUS_ERROR))
        nwarn = int(np.sum(self.status == STATUS_ITERATION_LIMIT))
        if nerr:      # @error branch, src/controller/execute.jl:484-489
            warnings.warn(f"MPC terminated without solution: returning last solution shifted "
                          f"({nerr} of {B} controllers)", RuntimeWarning)
        if nwarn:     # @warn branch, :491-496
            warnings.warn(f"MPC termination status not OPTIMAL: keeping solution anyway "
                          f"({nwarn} of {B} controllers)", RuntimeWarning)
        self.lastu0 = u0.copy()                       # getinput!: lastu0 <- u - uop
        return u0 + self.uop

    __call__ = moveinput

    def _bc(self, v, n, name):
        a = np.asarray(v, float)
        if a.shape == (n,):
            a = np.broadcast_to(a, (self.B, n))
        if a.shape != (self.B, n):
            raise ValueError(f"{name} size must be ({n},)")       # DimensionMismatch
        return a

    def getinfo(self):
        """`getinfo` (src/controller/execute.jl:145-198) for every controller of the batch: ΔU, ϵ, J,
        U, u, lastu, d, D̂, x̂, ŷ, Ŷ, x̂end, Ŷs, R̂y, R̂u, W (+ the reference's ASCII aliases).  Ŷ comes
        from the kernel (`moveinput(..., want_info=True)`, predict! transcription.jl:1136-1145); the
        quantities derived from it (J, ŷ-independent ones aside) need that flag too."""
        nu, Hc, Hp, ny, nd = self.nu, self.Hc, self.Hp, self.ny, self.nd
        DU = self.Z[:, :self.nDU]
        blk = np.repeat(np.arange(Hc), self.nb)
        cum = np.cumsum(DU.reshape(self.B, Hc, nu), axis=1)
        U0 = cum[:, blk, :].reshape(self.B, -1) + np.tile(self._lastu0_prev, self.Hp)
        xhat0, Rhaty, Rhatu = self._ginfo
        _, _, d0, Dh0 = self._winfo
        eps = self.Z[:, -1].copy() if self.neps else np.zeros(self.B)
        info = {"ΔU": DU.copy(), "ϵ": eps, "u": self.lastu0 + self.uop, "U": U0 + self.Uop,
                "lastu": self._lastu0_prev + self.uop, "x̂": xhat0 + self.xhop, "R̂y": Rhaty.copy(), "R̂u": Rhatu.copy(),
                "d": (d0 + self.dop) if nd > 0 else np.zeros((self.B, 0)),
                "D̂": (Dh0 + self.Dop) if nd > 0 else np.zeros((self.B, 0)),
                "Ŷs": np.zeros((self.B, self.nY))}          # stochastic predictions: InternalModel only (predictstoch!)
        # ŷ(k) = Ĉ x̂0 + D̂d d0 + yop (evaloutput, execute.jl:297-314)
        yk = np.einsum("bij,bj->bi", self._Chat, xhat0) + self.yop
        if nd > 0:
            yk = yk + np.einsum("bij,bj->bi", self._Dhd, d0)
        info["ŷ"] = yk
        # x̂end = x̂0(k+Hp): the augmented model driven by the optimal inputs (predict!, transcription.jl:1136-1145)
        x = xhat0.copy()
        dop_x = self.fhop - self.xhop
        U0s = U0.reshape(self.B, Hp, nu)
        X0 = np.empty((self.B, Hp, self.nxh))
        for t in range(Hp):
            x = np.einsum("bij,bj->bi", self._Ahat, x) + np.einsum("bij,bj->bi", self._Bhu, U0s[:, t]) + dop_x
            if nd > 0:
                dt = d0 if t == 0 else Dh0[:, (t - 1) * nd:t * nd]
                x = x + np.einsum("bij,bj->bi", self._Bhd, dt)
            X0[:, t] = x
        if getattr(self, "_ms_kernel", False):     # X̂0 is part of the MultipleShooting decision vector: the kernel's own
            X0 = self.hd.get(GET_XHAT_MS)
            x = X0[:, -1]
        info["x̂end"] = x + self.xhop
        # decision vector in the transcription's layout: [ΔU; ϵ] or [ΔU; X̂0(k+1..k+Hp); ϵ] (get_nZ_mpc, transcription.jl:2-7)
        parts = [DU] + ([X0.reshape(self.B, -1)] if self.transcription == "MultipleShooting" else []) + ([eps[:, None]] if self.neps else [])
        info["Z̃"] = np.concatenate(parts, axis=1)
        info["X̂0"] = X0.reshape(self.B, -1)
        if self._Yhat0 is not None:
            info["Ŷ"] = self._Yhat0 + self.Yop
            # J = (Ŷ-R̂y)'M(Ŷ-R̂y) + ΔU'N ΔU + (U-R̂u)'L(U-R̂u) + C ϵ²   (obj_nonlinprog!, general.jl:107 with r)
            ey, eu = info["Ŷ"] - Rhaty, info["U"] - Rhatu
            if getattr(self, "Mfull", None) is not None:
                Jy = np.einsum("bi,bij,bj->b", ey, self.Mfull, ey)
            elif self.Mblk is not None:
                eyb = ey.reshape(self.B, Hp, ny)
                Jy = np.einsum("bti,btij,btj->b", eyb, self.Mblk, eyb)
            else:
                Jy = np.sum(np.tile(self.Mwt, Hp) * ey * ey, axis=1)
            Jdu = (np.einsum("bi,bij,bj->b", DU, self.Ndense, DU) if getattr(self, "Ndense", None) is not None
                   else np.sum(np.tile(self.Nwt, Hc) * DU * DU, axis=1))
            Ju = (np.einsum("bi,bij,bj->b", eu, self.Ldense, eu) if getattr(self, "Ldense", None) is not None
                  else np.sum(np.tile(self.Lwt, Hp) * eu * eu, axis=1))
            info["J"] = Jy + Jdu + Ju + (self.Cwt * eps * eps if self.neps else 0.0)
            if self.nw > 0:       # W = Wy ŷe + Wu ue + Wd d̂e + Wr r̂e   (execute.jl:221, relaxW)
                xhat0, Rhaty, d0, Dh0 = self._winfo
                Hp, ny, nd = self.Hp, self.ny, self.nd
                yk = np.einsum("bij,bj->bi", self._Chat, xhat0) + self.yop
                if nd > 0:
                    yk = yk + np.einsum("bij,bj->bi", self._Dhd, d0)
                ye = np.concatenate([yk, info["Ŷ"]], axis=1).reshape(self.B, Hp + 1, ny)
                U = info["U"].reshape(self.B, Hp, nu)
                ue = np.concatenate([U, U[:, -1:, :]], axis=1)
                re = np.concatenate([self._ry_now, Rhaty], axis=1).reshape(self.B, Hp + 1, ny)
                Wv = (np.einsum("bij,btj->bti", self.Wy, ye) + np.einsum("bij,btj->bti", self.Wu, ue)
                      + np.einsum("bij,btj->bti", self.Wr, re))
                if nd > 0:
                    de = np.concatenate([d0 + self.dop, Dh0 + self.Dop], axis=1).reshape(self.B, Hp + 1, nd)
                    Wv = Wv + np.einsum("bij,btj->bti", self.Wd, de)
                info["W"] = Wv.reshape(self.B, -1)
        for a, k in (("DeltaU", "ΔU"), ("epsilon", "ϵ"), ("Dhat", "D̂"), ("xhat", "x̂"), ("yhat", "ŷ"), ("Yhat", "Ŷ"),
                     ("xhatend", "x̂end"), ("Yhats", "Ŷs"), ("Rhaty", "R̂y"), ("Rhatu", "R̂u"), ("Ztilde", "Z̃"), ("Xhat0", "X̂0")):
            if k in info:
                info[a] = info[k]
        return info
