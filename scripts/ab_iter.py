"""Developer tool: compare two builds of the library after 1, 2, 3, ... interior-point iterations (iteration cap)
on a few C3 controllers: python scripts/ab_iter.py libA.so libB.so [B]"""
import sys, os, numpy as np
sys.path.insert(0, '.')
import mpcqp
from mpcqp import synth
cfg = synth.C3; B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
bt = synth.make_batch(cfg, B, seed=0)
for mi in (1, 2, 3, 5, 8):
    Zs = []
    for path in sys.argv[1:3]:
        mpcqp.api._lib = None
        lib = mpcqp.api.load_library(os.path.abspath(path))
        hd = mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, neps=1, max_iter=mi,
                          flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START | 16, lib=lib)
        hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
        hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt))
        hd.set_bounds(U0min=np.full((B, hd.nU), cfg.umin), U0max=np.full((B, hd.nU), cfg.umax), Y0max=np.full((B, hd.nY), cfg.ymax))
        Z = np.zeros((B, hd.nZ))
        u0, st, it = hd.step(bt["xhat0"], bt["lastu0"], bt["ry"], Z)
        Zs.append(Z.copy()); hd.close()
    d = np.abs(Zs[0] - Zs[1])
    print(f"max_iter {mi}: max |dZ| {d.max():.3e} at {np.unravel_index(d.argmax(), d.shape)}; per-variable max {np.round(d.max(axis=0), 12)[:8]} ...", flush=True)
