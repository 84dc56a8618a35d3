"""A/B several builds of the library on a bench workload: [AB_CFG=C2] python scripts/ab_lib.py libA.so libB.so ..."""
import sys, os, time, numpy as np
sys.path.insert(0, '.')
import ctypes as C
import mpcqp
from mpcqp import synth
cfg = synth.get_config(os.environ.get('AB_CFG', 'C3')); B = int(os.environ.get('AB_B', 65536))
bt = synth.make_batch(cfg, B, seed=0)
Zref = None
for path in sys.argv[1:]:
    mpcqp.api._lib = None
    lib = mpcqp.api.load_library(os.path.abspath(path))
    hd = mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, neps=0 if np.isinf(cfg.Cwt) else 1, flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START, lib=lib)
    hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
    hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt) if np.isfinite(cfg.Cwt) else None)
    full = lambda v, n: None if not np.isfinite(v) else np.full((B, n), float(v))
    hd.set_bounds(U0min=full(cfg.umin, hd.nU), U0max=full(cfg.umax, hd.nU), DUmin=full(cfg.dumin, hd.nDU), DUmax=full(cfg.dumax, hd.nDU),
                  Y0min=full(cfg.ymin, hd.nY), Y0max=full(cfg.ymax, hd.nY),
                  **({'C_ymax': np.ones((B, hd.nY)), 'C_ymin': np.ones((B, hd.nY))} if os.environ.get('AB_SOFT') else {}))   # AB_SOFT=1: explicit softness arrays
    hd.prepare()
    Z = np.zeros((B, hd.nZ)); ms = []
    for rep in range(4):
        u0, st, it = hd.step(bt["xhat0"], bt["lastu0"], bt["ry"], Z)
        ms.append(hd.last_step_ms())
    if Zref is None: Zref = Z.copy()
    dz = np.max(np.abs(Z - Zref)[:, :-1], axis=1) / np.maximum(1.0, np.max(np.abs(Zref[:, :-1]), axis=1))
    print(f"{os.path.basename(path)}: kernel ms {['%.2f' % m for m in ms]}  optimal {np.mean(st == 0):.6f} iters {it.mean():.3f}  checksum {Z.sum():.12e}  "
          f"max rel dU diff vs first {dz.max():.2e} (99.9%: {np.quantile(dz, 0.999):.1e})", flush=True)
    hd.close()
