"""Time K1 (k_predmat, the setmodel! re-condensation) on a bench workload: [AB_CFG=C3] python scripts/time_k1.py libA.so libB.so ...
Prints the K1 / K1 + K2 times of mpcqp_recondense_device and a checksum of the tables (through the first step's optimum)."""
import sys, os, numpy as np
sys.path.insert(0, '.')
import mpcqp
from mpcqp import synth
cfg = synth.get_config(os.environ.get('AB_CFG', 'C3')); B = int(os.environ.get('AB_B', 65536))
bt = synth.make_batch(cfg, B, seed=0)
for path in sys.argv[1:]:
    mpcqp.api._lib = None
    lib = mpcqp.api.load_library(os.path.abspath(path))
    hd = mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, neps=0 if np.isinf(cfg.Cwt) else 1, flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START, lib=lib)
    hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
    hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt) if np.isfinite(cfg.Cwt) else None)
    full = lambda v, n: None if not np.isfinite(v) else np.full((B, n), float(v))
    hd.set_bounds(U0min=full(cfg.umin, hd.nU), U0max=full(cfg.umax, hd.nU), DUmin=full(cfg.dumin, hd.nDU), DUmax=full(cfg.dumax, hd.nDU),
                  Y0min=full(cfg.ymin, hd.nY), Y0max=full(cfg.ymax, hd.nY))
    hd.prepare()
    k1, k12 = [], []
    for rep in range(5):
        hd.recondense_device()
        hd.sync() if hasattr(hd, "sync") else None
        k1.append(hd.last_predmat_ms()); k12.append(hd.last_condense_ms())
    Z = np.zeros((B, hd.nZ))
    u0, st, it = hd.step(bt["xhat0"], bt["lastu0"], bt["ry"], Z)
    print(f"{os.path.basename(path)}: K1 ms {['%.3f' % m for m in k1]}  K1+K2 ms {['%.3f' % m for m in k12]}  optimal {np.mean(st == 0):.6f} "
          f"iters {it.mean():.3f} checksum {Z.sum():.12e}", flush=True)
    hd.close()
