"""Developer tool: the C3 instances (B = 65536, seed 0) where the kernel and the oracle's C port differ most, against the
independent oracle (certificate, rigorous error bound).  python scripts/hard_instances.py [K]"""
import sys, numpy as np
sys.path.insert(0, '.')
import mpcqp
from mpcqp import synth
from tests.parity_util import run_batch, make_oracle, rel_err
from oracle import cport, qp
cfg = synth.C3; B = 65536; K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bt = synth.make_batch(cfg, B, seed=0)
got = run_batch(cfg, bt)
Z = got["Z"]; nDU = cfg.nu * cfg.Hc
Zc, u0c, stc, itc = cport.from_synth(cfg, bt).step(bt["xhat0"], bt["lastu0"], bt["ry"])
dif = rel_err(Z, Zc, nDU)
hard = np.argsort(-dif)[:K]
print("instances with kernel-vs-C-port difference > 1e-5:", int((dif > 1e-5).sum()), " > 2e-6:", int((dif > 2e-6).sum()))
for i in hard:
    m = make_oracle(cfg, bt, i)
    m.initpred(bt["xhat0"][i], bt["lastu0"][i], bt["ry"][i]); m.linconstraint()
    z, st, info = qp.solve_qp(*m.qp_data(), m.warmstart(), return_info=True)
    sc = max(1.0, np.abs(z[:nDU]).max())
    print(f"i={i:6d} kernel-vs-cport {dif[i]:.2e}  oracle cert={info['certificate']:10s} bound={info.get('err_bound', 0):.2e} "
          f"kernel-vs-oracle {np.abs(Z[i,:nDU]-z[:nDU]).max()/sc:.2e} cport-vs-oracle {np.abs(Zc[i,:nDU]-z[:nDU]).max()/sc:.2e} eps={z[-1]:.3f} iters={got['iters'][i]}")
