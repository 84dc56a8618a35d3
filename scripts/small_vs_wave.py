"""Developer tool: a small problem (nZ~ <= 16) on the four-controllers-per-wavefront kernel against the same problem on the
one-controller-per-wavefront kernel (MPCQP_SMALL=0 in a child process) at several batch sizes.
   python scripts/small_vs_wave.py [CFG] [B ...]"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, '.')
    import warnings
    import numpy as np, mpcqp
    from mpcqp import synth
    from tests.parity_util import make_controller
    cfg = synth.get_config(sys.argv[2])
    for B in [int(v) for v in sys.argv[3:]]:
        bt = synth.make_batch(cfg, B, seed=0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mpc = make_controller(cfg, bt, cold_start=True)
            ms = []
            for rep in range(6):
                mpc.lastu0 = bt["lastu0"].copy()
                mpc.moveinput(bt["xhat0"], bt["ry"]); ms.append(mpc.hd.last_step_ms())
        print(f"  B {B:6d} kind {mpc.kernel} ms {min(ms):.4f} -> {B / min(ms) * 1e3:.4g} solves/s, iters {mpc.iters.mean():.2f}, optimal {np.mean(mpc.status == 0):.4f}", flush=True)
    sys.exit(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
Bs = sys.argv[2:] or ["1024", "4096", "16384", "65536"]
for small in ("1", "0"):
    print(f"MPCQP_SMALL={small}", flush=True)
    subprocess.run([sys.executable, __file__, "--child", cfg] + Bs, env=dict(os.environ, MPCQP_SMALL=small))
