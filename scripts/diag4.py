import numpy as np, sys
sys.path.insert(0,'.')
import mpcqp
from mpcqp import synth
from tests.parity_util import run_batch
cfg=synth.C3; B=262144
bt=synth.make_batch(cfg,B,seed=3)
got=run_batch(cfg,bt)
Z=got["Z"]; eps=Z[:,-1]
viol=(got["Yhat"]-cfg.ymax-eps[:,None]).max(axis=1)
idx=np.argsort(-viol)[:6]
print("IDX", idx.tolist(), "iters", got["iters"][idx].tolist())
big=np.argsort(-got["iters"])[:12]
print("MAXIT", big.tolist(), got["iters"][big].tolist())
np.save("gpurun_out/diag4_Z.npy", Z[np.concatenate([idx,big])])
