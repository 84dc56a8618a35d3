"""Round 6 diagnostic of the stage-structured kernel as a fallback of large condensed problems:
per-instance status / error of shape 12,4,4,46,46 (nZ~ = 185) against the C port, and of the MultipleShooting
transcription on C3 shapes against the condensed kernel.  Prints what a tightened test would have to hold."""
import sys, os, warnings
sys.path.insert(0, '.')
import numpy as np
import mpcqp
from mpcqp import synth, api
from oracle import cport
from tests.parity_util import make_controller, rel_err

B = int(os.environ.get("DIAG_B", 256))
cfg = synth.get_config("12,4,4,46,46")
bt = synth.make_batch(cfg, B, seed=11)
hd = mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, neps=1, flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START)
hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt))
full = lambda v, n: None if not np.isfinite(v) else np.full((B, n), float(v))
hd.set_bounds(U0min=full(cfg.umin, hd.nU), U0max=full(cfg.umax, hd.nU), Y0max=full(cfg.ymax, hd.nY))
print("kind", hd.prepare(), "lds", hd.lds_bytes())
Z = np.zeros((B, hd.nZ))
_, st, it = hd.step(bt["xhat0"], bt["lastu0"], bt["ry"], Z)
print("ms", hd.last_step_ms())
Zc, _, stc, itc = cport.from_synth(cfg, bt).step(bt["xhat0"], bt["lastu0"], bt["ry"])
err = rel_err(Z, Zc, hd.nDU)
print("nZ185: status counts", np.unique(st, return_counts=True), "cport", np.unique(stc, return_counts=True))
print("iters", it.mean(), it.max(), "cport", itc.mean(), itc.max())
o = np.argsort(-err)[:12]
print("worst", [(int(i), float(err[i]), int(st[i]), int(it[i]), int(itc[i])) for i in o])
au = hd.audit()
print("audit of worst", {k: v[o[:6]].tolist() for k, v in au.items()})

cfg = synth.C3
B2 = int(os.environ.get("DIAG_B2", 2048))
bt = synth.make_batch(cfg, B2, seed=5)
out = {}
for tr in ("SingleShooting", "MultipleShooting"):
    mpc = make_controller(cfg, bt, transcription=tr)
    mpc.lastu0 = bt["lastu0"].copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        mpc.moveinput(bt["xhat0"], bt["ry"])
    a_ = mpc.hd.audit(); out[tr] = (mpc.Z.copy(), mpc.status.copy(), mpc.iters.copy(), np.stack([a_["mu"], a_["rd"], a_["rp"], a_["polished"].astype(float)], 1))
e = rel_err(out["MultipleShooting"][0], out["SingleShooting"][0], cfg.nu * cfg.Hc)
stm = out["MultipleShooting"][1]
print("C3 MS status", np.unique(stm, return_counts=True), "iters mean/max", out["MultipleShooting"][2].mean(), out["MultipleShooting"][2].max())
o = np.argsort(-e)[:10]
print("C3 MS worst", [(int(i), float(e[i]), int(stm[i]), int(out["MultipleShooting"][2][i])) for i in o])
bad = np.where(stm != 0)[0]
print("non-optimal:", [(int(i), float(e[i]), int(out["MultipleShooting"][2][i]), out["MultipleShooting"][3][i].tolist()) for i in bad[:10]])
