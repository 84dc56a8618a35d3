"""Timing of the MHE solve kernel on resident data: B estimators, full window."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import mpcqp
from mpcqp import synth, mhe as pm
import mhe_util
if os.environ.get('MPCQP_LIB'):
    mpcqp.api.load_library(os.environ['MPCQP_LIB']); mpcqp.api._lib = mpcqp.api.load_library(os.environ['MPCQP_LIB'])

cfg = synth.get_mhe_config(sys.argv[1] if len(sys.argv) > 1 else "C5")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nper = cfg.He + 6
bt = synth.make_mhe_batch(cfg, B, seed=1)
Y, U, D = synth.make_mhe_data(cfg, bt, nper, seed=0)
bm = mhe_util.make_product(cfg, bt, keep_windows=False)
h = bm.handle
dev = torch.device("cuda:0")
Yd, Ud = torch.tensor(Y, device=dev), torch.tensor(U, device=dev)
Dd = torch.tensor(D, device=dev) if cfg.nd else None
for k in range(nper):
    h.prepare_device(Yd[k].data_ptr(), Dd[k].data_ptr() if cfg.nd else 0)
    h.update_device(Ud[k].data_ptr(), Yd[k].data_ptr(), Dd[k].data_ptr() if cfg.nd else 0)
    h.sync()
    st = h.get(pm.GET_STATUS); it = h.get(pm.GET_ITERS)
    print(f"k={k:2d} Nk={h.Nk:2d} ms={h.last_ms():8.3f} bad={int((st!=0).sum())} iters mean {it.mean():.2f} max {it.max()}", flush=True)
