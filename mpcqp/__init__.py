"""Import alias for the package directory `modelpredictivecontrol.jl_amd/` (a dotted directory
name cannot be spelled in an `import` statement)."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "modelpredictivecontrol.jl_amd"))
from .api import *  # noqa: F401,F403,E402
from .api import (BatchLinMPC, Handle, MultiHandle, MpcqpError, load_library, move_blocking, colmajor,  # noqa: F401,E402
                  steady_kalman_gain,
                  EXPORTS, DEFAULT_LIB)
from . import synth, api, sharding, mhe  # noqa: F401,E402
from .mhe import BatchMHE, MheHandle  # noqa: F401,E402
