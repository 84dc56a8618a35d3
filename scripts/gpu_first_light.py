import numpy as np, sys, time
sys.path.insert(0,'.')
import mpcqp
from mpcqp import synth
from tests.parity_util import *
for name,B in (("C2",64),("C3",64)):
    cfg=synth.CONFIGS[name]; bt=synth.make_batch(cfg,B,seed=0)
    got=run_batch(cfg,bt); ref=oracle_batch(cfg,bt)
    e=rel_err(got["Z"],ref["Z"],cfg.nu*cfg.Hc)
    print(name,"status",np.bincount(got["status"],minlength=3),"iters mean %.1f max %d"%(got["iters"].mean(),got["iters"].max()),"err max %.2e med %.2e"%(e.max(),np.median(e)),"certified",ref["certified"].mean(), flush=True)
# timing
for name,B in (("C2",1024),("C3",8192),("C3",65536)):
    cfg=synth.CONFIGS[name]; bt=synth.make_batch(cfg,B,seed=0)
    t0=time.time(); mpc=make_controller(cfg,bt); t1=time.time()
    mpc.lastu0=bt["lastu0"].copy()
    for rep in range(3):
        mpc.Z[:]=0; mpc.lastu0=bt["lastu0"].copy()
        t2=time.time(); u=mpc.moveinput(bt["xhat0"],bt["ry"]); t3=time.time()
        print(name,B,"setup %.2fs step wall %.1f ms kernel %.2f ms -> %.3g QP/s"%(t1-t0,(t3-t2)*1e3,mpc.hd.last_step_ms(),B/(mpc.hd.last_step_ms()*1e-3)),"condense ms",mpc.hd.last_condense_ms(),"status",np.bincount(mpc.status,minlength=3),"iters mean %.2f"%mpc.iters.mean(), flush=True)
