"""Developer tool: kernel time of a small problem against the iteration limit (intercept = set-up + launch, slope = one iteration)."""
import sys, warnings
sys.path.insert(0, '.')
import numpy as np, mpcqp
from mpcqp import synth
from tests.parity_util import make_controller
cfg = synth.get_config(sys.argv[1] if len(sys.argv) > 1 else "C2")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
bt = synth.make_batch(cfg, B, seed=0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mpc = make_controller(cfg, bt, cold_start=True)
    for lim in (1, 2, 4, 8, 16, 80):
        mpc.hd.set_iteration_limit(lim)
        ms = []
        for rep in range(6):
            mpc.lastu0 = bt["lastu0"].copy()
            mpc.moveinput(bt["xhat0"], bt["ry"]); ms.append(mpc.hd.last_step_ms())
        print(f"limit {lim:3d}: kernel {min(ms) * 1e3:8.1f} us  (kind {mpc.kernel}, iters {mpc.iters.mean():.2f})", flush=True)
