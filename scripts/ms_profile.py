"""Cycles per phase of the MultipleShooting step (scripts/ms_profile.sh builds the instrumented library).
   python scripts/ms_profile.py [config] [B]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import mpcqp
from mpcqp import synth, api
from tests.parity_util import make_controller
lib = api.load_library(os.path.join('modelpredictivecontrol.jl_amd', 'lib', 'ab', 'libmpcqp_msprof.so'))
cfg = synth.get_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
bt = synth.make_batch(cfg, B, seed=5)
mpc = make_controller(cfg, bt, lib=lib, transcription="MultipleShooting")
mpc.lastu0 = bt["lastu0"].copy()
mpc.moveinput(bt["xhat0"], bt["ry"])
P = mpc.hd.get(api.GET_XHAT_MS).reshape(B, -1)[:, :8]
it = mpc.iters.mean()
names = ["residuals", "stage data", "factor", "psi sweep", "newton x2", "update", "polish", "run total"]
tot = P[:, 7].mean()
print(f"{cfg.name}: B {B} kernel {mpc.hd.last_step_ms():.1f} ms, iterations {it:.2f}; cycles per wavefront {tot:.3e} ({tot / max(it, 1):.3e} per iteration)")
for i, n in enumerate(names):
    if n != "-":
        print(f"  {n:12s} {P[:, i].mean():12.3e}  {100 * P[:, i].mean() / tot:5.1f} %   per iteration {P[:, i].mean() / max(it, 1):10.3e}")
