"""MI355X-native batched LinMPC step (condense + QP solve) -- the `moveinput!` hot path of
JuliaControl/ModelPredictiveControl.jl behind a C-ABI (include/mpcqp.h).

The directory name carries a dot, so it is imported through the `mpcqp` alias package at the
repository root (`import mpcqp`), which points its `__path__` here.
"""
from .api import (BatchLinMPC, Handle, MultiHandle, MpcqpError, load_library, move_blocking, colmajor,  # noqa: F401
                  steady_kalman_gain,
                  STATUS_OPTIMAL, STATUS_ITERATION_LIMIT, STATUS_ERROR, EXPORTS,
                  FLAG_RY_CONSTANT, FLAG_COLD_START, FLAG_KEEP_QP, FLAG_WARM_DUAL,
                  GET_HESSIAN, GET_STEPRESP, GET_KMAT, GET_BVEC, GET_QTILDE, GET_FVEC)
from . import synth, sharding, mhe  # noqa: F401
from .mhe import BatchMHE, MheHandle  # noqa: F401
