"""Developer tool: time the MultipleShooting kernel of several library builds: python scripts/ms_time.py CFG B lib1.so lib2.so ..."""
import sys, os, warnings
sys.path.insert(0, '.')
import numpy as np, mpcqp
from mpcqp import synth, api
from tests.parity_util import make_controller
cfg = synth.get_config(sys.argv[1]); B = int(sys.argv[2])
bt = synth.make_batch(cfg, B, seed=0)
ref = None
for path in sys.argv[3:]:
    mpcqp.api._lib = None
    lib = mpcqp.api.load_library(os.path.abspath(path))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mpc = make_controller(cfg, bt, transcription="MultipleShooting", lib=lib, cold_start=True)
        ms = []
        for rep in range(3):
            mpc.lastu0 = bt["lastu0"].copy(); mpc.Z[:] = 0
            mpc.moveinput(bt["xhat0"], bt["ry"]); ms.append(mpc.hd.last_step_ms())
    if ref is None: ref = mpc.Z.copy()
    print(f"{os.path.basename(path)}: kernel kind {mpc.kernel} lds {mpc.hd.lds_bytes()} ms {['%.2f' % m for m in ms]} -> {B / min(ms) * 1e3:.4g} solves/s, status {np.bincount(mpc.status, minlength=3)}, iters {mpc.iters.mean():.2f}, max |dZ| vs first {np.abs(mpc.Z - ref).max():.2e}", flush=True)
