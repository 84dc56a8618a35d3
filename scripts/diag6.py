import numpy as np, sys
sys.path.insert(0,'.')
import mpcqp
from mpcqp import synth
from tests.parity_util import run_batch, rel_err
from oracle import cport
cfg=synth.C3; B=65536
bt=synth.make_batch(cfg,B,seed=0)
got=run_batch(cfg,bt)
Zc,u0c,stc,itc=cport.from_synth(cfg,bt).step(bt["xhat0"],bt["lastu0"],bt["ry"])
dif=rel_err(got["Z"],Zc,40)
hard=np.argsort(-dif)[:12]
print("HARD", hard.tolist(), dif[hard].tolist(), got["iters"][hard].tolist(), itc[hard].tolist())
np.savez("gpurun_out/diag6.npz", Z=got["Z"][hard], Zc=Zc[hard], hard=hard)
