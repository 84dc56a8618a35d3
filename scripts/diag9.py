import sys, numpy as np, warnings, os
sys.path.insert(0,'.')
warnings.filterwarnings("ignore")
import mpcqp
from mpcqp import synth
cfg = synth.Config("dbg", nx=3, nu=int(sys.argv[1]), ny=int(sys.argv[2]), Hp=int(sys.argv[3]), Hc=int(sys.argv[4]), umin=-np.inf, umax=np.inf, ymin=-1.2, ymax=1.0)
bt = synth.make_batch(cfg, 4, seed=1)
from tests.parity_util import run_batch
for jit in ("1", "0"):
    os.environ["MPCQP_JIT"] = jit
    mpcqp.api._lib = None
    got = run_batch(cfg, bt, keep_qp=True)
    hd = got["mpc"].hd
    print("JIT", jit, "status", got["status"], "iters", got["iters"], "Z0[:4]", got["Z"][0,:4], "eps", got["Z"][:, -1])
