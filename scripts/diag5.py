import numpy as np, sys
sys.path.insert(0,'.')
import mpcqp
from mpcqp import synth
from tests.parity_util import run_batch
cfg=synth.C3; B=65536
bt=synth.make_batch(cfg,B,seed=0)
got=run_batch(cfg,bt)
it=got["iters"]; pol=it>=1000; it=it%1000
print("polished fraction", pol.mean(), "iters polished", it[pol].mean(), "iters unpolished", it[~pol].mean(), "n unpolished", (~pol).sum())
print("hist unpolished iters", np.bincount(it[~pol])[:60])
np.savez("gpurun_out/diag5.npz", Z=got["Z"], it=got["iters"])
