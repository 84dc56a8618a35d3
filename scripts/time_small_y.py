"""C2-size controllers with a soft output band at B = 65536: kernel time of the step (third of three cold-started calls),
iterations, kernel kind.  MPCQP_SMALL_Y=0 python scripts/time_small_y.py  runs the one-controller-per-wavefront kernel."""
import sys, os, numpy as np
sys.path.insert(0, '.')
import mpcqp
from mpcqp import synth
B = int(os.environ.get("AB_B", 65536))
cfg = synth.Config('C2y', nx=4, nu=2, ny=2, Hp=20, Hc=5, Cwt=1e5)
bt = synth.make_batch(cfg, B, seed=0)
mpc = mpcqp.BatchLinMPC(bt['Ahat'], bt['Bhu'], bt['Chat'], Hp=20, Hc=5, Cwt=1e5, Mwt=np.ones(2), Nwt=np.full(2, 0.1), Lwt=np.zeros(2))
mpc.setconstraint(umin=[-1, -1], umax=[1, 1], Δumin=[-0.5, -0.5], Δumax=[0.5, 0.5], ymin=[-0.15, -0.2], ymax=[0.15, 0.2])
ms = []
for rep in range(3):
    mpc.lastu0 = bt['lastu0'].copy(); mpc.Z[:] = 0.0
    mpc.moveinput(bt['xhat0'], bt['ry'])
    ms.append(mpc.hd.last_step_ms())
print(f"kind {mpc.hd.kernel_kind()} ms {['%.2f' % m for m in ms]} optimal {np.mean(mpc.status == 0):.6f} iters {mpc.iters.mean():.3f} max {mpc.iters.max()} "
      f"eps>1e-6: {np.mean(mpc.Z[:, -1] > 1e-6):.3f} checksum {mpc.Z.sum():.10e}")
