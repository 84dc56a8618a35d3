"""Developer tool: the C3 bench workload with extra handle flags (e.g. 16 = MPCQP_FLAG_NO_POLISH): kernel time, iterations,
agreement with the default run.  python scripts/ab_flags.py FLAGS [FLAGS ...]"""
import sys, os, numpy as np
sys.path.insert(0, '.')
import mpcqp
from mpcqp import synth
cfg = synth.C3; B = 65536
bt = synth.make_batch(cfg, B, seed=0)
Zref = None
for fl in [0] + [int(a) for a in sys.argv[1:]]:
    hd = mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, neps=1, flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START | fl)
    hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
    hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt))
    hd.set_bounds(U0min=np.full((B, hd.nU), cfg.umin), U0max=np.full((B, hd.nU), cfg.umax), Y0max=np.full((B, hd.nY), cfg.ymax))
    Z = np.zeros((B, hd.nZ)); ms = []
    for rep in range(4):
        u0, st, it = hd.step(bt["xhat0"], bt["lastu0"], bt["ry"], Z)
        ms.append(hd.last_step_ms())
    if Zref is None: Zref = Z.copy()
    dz = np.max(np.abs(Z - Zref)[:, :-1], axis=1) / np.maximum(1.0, np.max(np.abs(Zref[:, :-1]), axis=1))
    print(f"flags +{fl}: kernel ms {['%.2f' % m for m in ms]} optimal {np.mean(st == 0):.6f} iters {it.mean():.3f} max {it.max()} "
          f"max rel dU diff vs default {dz.max():.2e} (99.9%: {np.quantile(dz, 0.999):.1e}, median {np.median(dz):.1e})", flush=True)
    hd.close()
