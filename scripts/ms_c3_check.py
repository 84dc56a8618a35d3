"""MultipleShooting kernel on BASELINE-shaped batches (GPU): statuses, iterations, agreement with the condensed kernels and
the time per step.   python scripts/ms_c3_check.py [config] [B]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, warnings
import mpcqp
from mpcqp import synth, api
from tests.parity_util import make_controller, rel_err
cfg = synth.get_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
DR = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0        # dual regularisation of the MultipleShooting run (0: default)
import os
bt = synth.make_batch(cfg, B, seed=int(os.environ.get("MS_SEED", 5)))
out = {}
for tr in ("SingleShooting", "MultipleShooting"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mpc = make_controller(cfg, bt, transcription=tr, **({"dual_reg": DR} if tr == "MultipleShooting" and DR else {}))
        mpc.lastu0 = bt["lastu0"].copy()
        mpc.moveinput(bt["xhat0"], bt["ry"])
        mpc.Z[:] = 0
        mpc.lastu0 = bt["lastu0"].copy()
        t = time.time(); mpc.moveinput(bt["xhat0"], bt["ry"]); dt = time.time() - t
    out[tr] = mpc
    print(tr, "kernel", mpc.kernel, "status counts", np.bincount(mpc.status, minlength=3), "iters mean %.2f max %d" % (mpc.iters.mean(), mpc.iters.max()),
          "step %.2f ms (kernel %.2f ms)" % (dt * 1e3, mpc.hd.last_step_ms()), flush=True)
a, b_ = out["SingleShooting"], out["MultipleShooting"]
nDU = cfg.nu * cfg.Hc
e = rel_err(b_.Z, a.Z, nDU)
bad = np.flatnonzero(b_.status != 0)
print("max rel dU difference MS vs condensed %.3e (99.9%% %.3e); non-optimal MS members %s" % (e.max(), np.quantile(e, 0.999), bad[:10]))
if len(bad):
    au = b_.hd.get(api.GET_AUDIT)
    for i in bad[:5]:
        print("  member", i, "status", b_.status[i], "iters", b_.iters[i], "audit mu/rd/rp", au[i, :3], "err vs condensed %.2e" % e[i], "defect", b_.hd.get(api.GET_MS_DEFECT)[i])
