"""Which instances of a big synthetic batch does the step kernel NOT call optimal, and how far are they from the optimum?
   python scripts/nonoptimal_instances.py [config] [B] [seed]      (GPU; the oracle runs on the flagged instances only)"""
import sys, warnings
sys.path.insert(0, '.')
import numpy as np
import mpcqp
from mpcqp import synth, api
from oracle import qp, qp_hp
from tests.parity_util import run_batch, make_oracle, rel_err
cfg = synth.get_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
bt = synth.make_batch(cfg, B, seed=seed)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    got = run_batch(cfg, bt)
st, it = got["status"], got["iters"]
au = got["mpc"].hd.audit()
print(cfg.name, "B", B, "seed", seed, "status counts", np.bincount(st, minlength=3), "iters mean %.2f max %d" % (it.mean(), it.max()))
nDU = cfg.nu * cfg.Hc
for i in np.flatnonzero(st != 0)[:12]:
    m = make_oracle(cfg, bt, i)
    m.initpred(bt["xhat0"][i], bt["lastu0"][i], bt["ry"][i]); m.linconstraint()
    z, sto, info = qp.solve_qp(*m.qp_data(), m.warmstart(), return_info=True)
    e = rel_err(got["Z"][i:i + 1], z[None, :], nDU).max()
    line = f"  instance {i}: status {st[i]} iters {it[i]} mu {au['mu'][i]:.2e} rd {au['rd'][i]:.2e} rp {au['rp'][i]:.2e}; oracle {info['certificate']}: kernel vs oracle {e:.2e}"
    if info["certificate"] != "active-set" or "--hp" in sys.argv:
        zh, ih = qp_hp.solve_reference_qp(*m.qp_data(), z0=m.warmstart(), digits=50)
        line += f"; 50-digit optimum: kernel {rel_err(got['Z'][i:i + 1], zh[None, :], nDU).max():.2e}, oracle {rel_err(z[None, :], zh[None, :], nDU).max():.2e}"
    print(line, flush=True)
