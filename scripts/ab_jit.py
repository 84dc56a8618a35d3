"""Developer tool: time the specialisations built by scripts/ab_jit.sh.  python scripts/ab_jit.py CFG B name1 name2 ...
(one subprocess per variant: the cache directory is read once per process)"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np, mpcqp
    from mpcqp import synth
    cfg = synth.get_config(sys.argv[2]); B = int(sys.argv[3])
    bt = synth.make_batch(cfg, B, seed=0)
    hd = mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, neps=1, flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START)
    hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
    hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt))
    hd.set_bounds(U0min=np.full((B, hd.nU), cfg.umin), U0max=np.full((B, hd.nU), cfg.umax), Y0max=np.full((B, hd.nY), cfg.ymax))
    kind = hd.prepare()
    Z = np.zeros((B, hd.nZ)); ms = []
    for rep in range(5):
        u0, st, it = hd.step(bt["xhat0"], bt["lastu0"], bt["ry"], Z)
        ms.append(hd.last_step_ms())
    print(f"{sys.argv[4]:>12}: kind {kind} kernel ms {['%.2f' % m for m in ms]} best {B / min(ms) * 1e3:.4g} solves/s optimal {np.mean(st == 0):.6f} "
          f"iters {it.mean():.3f} checksum {Z.sum():.12e}", flush=True)
else:
    cfg, B = sys.argv[1], sys.argv[2]
    for name in sys.argv[3:]:
        env = dict(os.environ, MPCQP_CACHE_DIR=os.path.join(ROOT, "modelpredictivecontrol.jl_amd", "lib", "ab", "jit", name))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", cfg, B, name], env=env)
